"""The LQP / JACC formulations on the REDUCED (centroidal) system -- SURVEY 8 row f3, src/dwbc.cpp:3946-4302, 4455-4760:
ConfigureLQP_R / CalcControlTorqueLQP_R, ConfigureLQP_R_NC / CalcControlTorqueLQP_R_NC, CalcSingleTaskTorqueWithJACC_QP_R and
CalcSingleTaskTorqueWithJACC_QP_R_NC.  PARITY UNPINNED in the reference (timing harnesses only); the oracle restates the
problems (oracle/hqp_np.py), the device code is checked against it through the host emulation and on the GPU.

Task set: pelvis 6-D (contact-chain level) and a 6-D task on the upper body (non-contact level), the shape the reference's
harness uses (tests/sp_test/jacc_compare.cpp:456-487)."""
import numpy as np
import pytest

from oracle import hqp_np as H
from oracle.dwbc_reduced_np import ReducedCycle
from tests import cases
from tests.test_reduced_path import make

TASKS_R = [[(0, 0, (0, 0, 0))], [(0, 15, (0, 0, 0))]]
RS, VCD, NCD = 24, 18, 21


def _states(B, seed):
    q, fl, fs9 = cases.synth_batch(B, seed=seed, yaw=True)
    rng = np.random.default_rng(seed)
    fs = np.concatenate([fs9[:, :6], 0.3 * rng.uniform(-1, 1, size=(B, 3)), fs9[:, 6:9]], axis=1)  # level 1: 6-D (linear | angular)
    q[0] = cases.Q_CASE[1]
    return q, fl, fs


def _oracle(q, fs):
    r = make(ReducedCycle, TASKS_R)
    r.update_kinematics(q)
    r.set_contact([1, 1])
    r.reduced_dynamics()
    assert r.reduced_contact_constraint() == 1
    Js = [r.task_jacobian(i) for i in range(2)]
    f = [fs[:6], fs[6:]]
    out = dict(r=r, Js=Js)
    hq = H.configure_lqp_r(r, Js, f, [False, True])
    out["lqp_ok"] = hq.solveSequential()
    out["lqp"] = hq
    y = hq.hqp_hs_[-1].y_ans_
    out["lqp_tau"] = H.lqp_r_torque(r, y)
    hnc = H.configure_lqp_r_nc(r, y[:RS], Js[1], f[1], r.p[15])
    out["nc_ok"] = H.solve_lqp_r_nc(hnc)
    out["nc"] = hnc
    ok, acc, tau, fc, s, _ = H.jacc_qp_r(r, 0, Js[:1], f[:1], [])
    out["jacc"] = dict(ok=ok, acc=acc, tau=tau, f=fc, s=s)
    a, tq, g, fq, _ = H.jacc_qp_r_nc(r, acc, Js[1], f[1], r.p[15])
    out["jacc_nc"] = dict(acc=a, tau=tq, gacc=g, s=fq)
    return out


def test_oracle_reduced_formulations_are_consistent():
    """structure of the restated problems on the reference's CASE 1 state: sizes, the identities the reduction rests on, and that
    every answer keeps the rows it was asked to keep"""
    q, fl, fs = _states(1, 5)
    o = _oracle(q[0], fs[0])
    r, hq = o["r"], o["lqp"]
    assert o["lqp_ok"] == 1 and o["nc_ok"] == 1 and o["jacc"]["ok"] == 1
    assert [(h.ineq_const_size_, h.eq_const_size_) for h in hq.hqp_hs_] == [(36, 6), (56, 12), (0, 6)]
    assert [h.null_space_size_ for h in hq.hqp_hs_] == [30, 18, 12]
    y = hq.hqp_hs_[-1].y_ans_
    # reduced dynamics rows hold; J_task J_R_INV_T^T J_R^T = J_task on the contact-chain columns (dwbc.cpp:2976-2980)
    assert np.abs(r.A_R[:6] @ y[:RS] + r.J_CR.T[:6] @ y[RS:] + r.G_R[:6]).max() < 1e-8
    assert np.abs(r.J_CR @ y[:RS]).max() < 5e-2  # the contact level is a least-squares level WITH a cost (dwbc.cpp:4598): traded, not exact
    tau = o["lqp_tau"]
    lim = np.full(18, 200.0)
    lim[14] = 600.0
    assert (np.abs(tau) < lim).all() and tau[14] > 200.0  # the centroidal vertical force carries the body group: needs the 600
    # non-contact half: centroidal acceleration of the reduced answer is reproduced, accelerations inside +-5
    a = o["nc"].hqp_hs_[-1].y_ans_
    g = y[RS - 6 : RS]  # (a least-squares level with a cost again: the centroidal acceleration is approached, not met)
    assert np.linalg.norm(r.J_I_nc @ o["nc"].hqp_hs_[0].y_ans_ - g) < np.linalg.norm(g)
    assert np.abs(a).max() <= H.LQP_NC_ACC_LIM + 1e-6
    # JACC_R: dynamics, contact, task rows
    j = o["jacc"]
    dyn = r.A_R @ j["acc"] + r.J_CR.T @ j["f"] + r.G_R
    assert np.abs(dyn[:6]).max() < 1e-7 and np.abs(dyn[6:] - j["tau"]).max() < 1e-9
    assert np.abs(j["acc"][6:]).max() <= H.JACC_ACC_LIM + 1e-6 and np.abs(j["tau"][:12]).max() <= H.JACC_TAU_LIM + 1e-6
    # JACC_R_NC: a weighted least-squares problem (the upper-body Jacobian only sees the three waist joints, so the 6-D task
    # cannot be met): the answer is stationary for 1/2 |gacc|^2 + 5/2 |s|^2 + eps/2 |a|^2
    n = o["jacc_nc"]
    grad = r.J_I_nc.T @ n["gacc"] + H.JACC_NC_W_TASK * o["Js"][1][:, VCD:].T @ n["s"] + H.HQP_EPS * n["acc"]
    assert np.abs(grad).max() < 1e-9


def _rel(a, ref):
    return (np.abs(a - ref) / (1.0 + np.abs(ref))).max()


def test_emulated_reduced_lqp_and_jacc_match_oracle():
    from tests.emu.emu import Emu, EmuHQP

    B = 4
    q, fl, fs = _states(B, 41)
    e = Emu(cases.URDF, cases.CONTACTS_2, TASKS_R, None)
    r = e.run(q, fl, fs, dump=True, reduced=True)  # (the cycle's own QP status is not asserted: only its model quantities are used)
    dmp = r["dump"]
    rrec = EmuHQP.reduced_record(e, B, VCD, 12, [0], dmp)
    # ---- LQP_R
    eh = EmuHQP(B, RS + 12, [36, 56, 0], [6, 12, 6], [0, 1, 1], share_cost=True)
    eh.configure_lqp_r(e, [0, 1], RS, [0], rrec, fs)
    eh.solve()
    tau = eh.lqp_torque_r(e, [0, 1], RS, rrec)
    # ---- LQP_R_NC from the device answer
    y_dev = eh.block(2, 5, (RS + 12,)).copy()
    enc = EmuHQP(B, NCD, [42, 42], [6, 6], [1, 1], share_cost=True)
    enc.configure_lqp_nc(e, VCD, 1, dmp, fs, y_dev)
    enc.solve_levels(1, True)
    enc.solve_levels(2, False)
    # ---- JACC_R level 0, JACC_R_NC
    ej = EmuHQP(B, RS + 12, [20 + 36 + 24, 0], [18, 6], [0, 1])
    ej.set_exact(0)
    jout, jst = ej.jacc_solve_r(e, [0, 1], RS, [0], 0, rrec, fs, [])
    ejn = EmuHQP(B, NCD, [0], [12], [0], solve_first=True)
    nout, nst = ejn.jacc_solve_nc(e, VCD, 1, dmp, fs, jout)
    assert jst.all() and nst.all()
    for b in range(B):
        o = _oracle(q[b], fs[b])
        rr = o["r"]
        # the reduced record is the oracle's reduced system
        assert np.abs(EmuHQP.rrec_field(rrec, RS, "A", (RS, RS))[b] - rr.A_R).max() < 1e-8 * np.abs(rr.A_R).max()
        assert np.abs(EmuHQP.rrec_field(rrec, RS, "A_inv", (RS, RS))[b] - rr.A_R_inv).max() < 1e-9
        assert np.abs(EmuHQP.rrec_field(rrec, RS, "J_C", (12, RS))[b] - rr.J_CR).max() < 1e-10
        assert np.abs(EmuHQP.rrec_field(rrec, RS, "G", (RS,))[b] - rr.G_R).max() < 1e-8
        assert np.abs(EmuHQP.rrec_field(rrec, RS, "J_task", (6, RS))[b] - o["Js"][0] @ rr.J_R_INV_T.T).max() < 1e-9
        assert abs(EmuHQP.rrec_field(rrec, RS, "com", (1,))[b, 0] - np.linalg.norm(rr.A)) < 1e-8
        hq = o["lqp"]
        assert o["lqp_ok"] == 1
        for lv, h in enumerate(hq.hqp_hs_):
            assert eh.status(lv)[b] == 1 and eh.null_size(lv)[b] == h.null_space_size_
            if h.ineq_const_size_:
                assert np.abs(eh.block(lv, 0, (h.ineq_const_size_, RS + 12))[b] - h.A_).max() < 1e-8
                assert np.abs(eh.block(lv, 1, (h.ineq_const_size_,))[b] - h.a_).max() < 1e-7
            assert np.abs(eh.block(lv, 2, (h.eq_const_size_, RS + 12))[b] - h.B_).max() < 1e-8
            assert _rel(eh.block(lv, 5, (RS + 12,))[b], h.y_ans_) < 2e-6, (b, lv)
        assert _rel(tau[b], o["lqp_tau"]) < 1e-4
        hn = o["nc"]
        for lv, h in enumerate(hn.hqp_hs_):
            assert enc.status(lv)[b] == 1
            assert np.abs(enc.block(lv, 0, (42, NCD))[b] - h.A_).max() < 1e-9 and np.abs(enc.block(lv, 2, (6, NCD))[b] - h.B_).max() < 1e-9
            assert np.abs(enc.block(lv, 3, (6,))[b] - h.b_).max() < 1e-5  # b holds the device's own reduced answer
            assert _rel(enc.block(lv, 5, (NCD,))[b], h.y_ans_) < 1e-4, (b, lv)
        j = o["jacc"]
        assert _rel(jout[b, :RS], j["acc"]) < 1e-5 and _rel(jout[b, RS : RS + 18], j["tau"]) < 1e-3
        assert _rel(jout[b, RS + 18 : RS + 30], j["f"]) < 1e-3 and _rel(jout[b, RS + 30 : RS + 36], j["s"]) < 1e-5
        n = o["jacc_nc"]
        assert _rel(nout[b, :NCD], n["acc"]) < 1e-4 and _rel(nout[b, NCD : 2 * NCD], n["tau"]) < 1e-4
        assert np.abs(nout[b, 2 * NCD : 2 * NCD + 6] - n["gacc"]).max() < 1e-4 and np.abs(nout[b, 2 * NCD + 6 :] - n["s"]).max() < 1e-4


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_gpu_reduced_lqp_and_jacc_match_oracle_and_keep_their_rows():
    """The four reduced formulations on the device after a reduced cycle: against the restatement on a subset, through the
    rows each formulation must keep on the whole batch; the reduced-model getters; the refusals."""
    import libdwbc_amd as D
    from libdwbc_amd import hqp as Hq

    B, NS = 128, 4
    q, fl, fs = _states(B, 43)
    wbc = D.Batch(D.Model.from_urdf(cases.URDF), B, device=0)
    for c in cases.CONTACTS_2:
        wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    wbc.add_task(0, D.TASK_LINK_6D, 0)
    wbc.add_task(1, D.TASK_LINK_6D, 15)
    wbc.enable_dump(True)
    wbc.set_state(q); wbc.set_contact(fl); wbc.set_fstar_all(fs)
    wbc.solve()
    hr = D.HQP.for_lqp_r(wbc, RS, 12)
    with pytest.raises(D.DwbcError, match="reduced cycle first"):
        hr.configure_lqp_r(wbc)
    wbc.solve(reduced=True)
    with pytest.raises(D.DwbcError, match="full-model cycle first"):
        D.HQP.for_lqp(wbc, 12).configure_lqp(wbc)
    A_R, A_Ri, G_R, JI, JIi = wbc.get("A_R"), wbc.get("A_R_inv"), wbc.get("G_R"), wbc.get("J_I_nc"), wbc.get("J_I_nc_inv_T")
    JC = wbc.get("J_C")
    # ---- LQP_R + LQP_R_NC
    hr.configure_lqp_r(wbc)
    hr.solveSequential()
    assert hr.num_levels() == 3
    tau = hr.lqp_torque(wbc)
    y = hr.y_ans(2)
    hn = D.HQP.for_nc(wbc, NCD)
    with pytest.raises(D.DwbcError, match="non-contact link"):
        hn.configure_lqp_r_nc(wbc, hr, 0)
    hn.configure_lqp_r_nc(wbc, hr, 1)
    hn.solvefirst()
    hn.solveSequential()
    # ---- JACC_R + JACC_R_NC
    hj = D.HQP.for_lqp_r(wbc, RS, 12)
    hjn = D.HQP.for_nc(wbc, NCD)
    with pytest.raises(D.DwbcError, match="solve_jacc_r first"):
        hjn.solve_jacc_r_nc(wbc, 1, 0)
    hj.solve_jacc_r(wbc, 0)
    jr = D.HQP.jacc_result(wbc, 0, system_dof=RS)
    hjn.solve_jacc_r_nc(wbc, 1, 0)
    nr = hjn.jacc_nc_result(wbc)
    for lv in range(3):
        assert hr.get(lv, Hq.STATUS).all()
    assert hn.get(0, Hq.STATUS).all() and hn.get(1, Hq.STATUS).all() and jr["status"].all() and nr["status"].all()
    # ---- rows every instance keeps
    lim = np.full(18, 200.0)
    lim[14] = 600.0
    assert (np.abs(tau) < lim + 1e-6).all()
    dyn = np.einsum("bij,bj->bi", A_R[:, :6], y[:, :RS]) + np.einsum("bji,bj->bi", JC[:, :, :6], y[:, RS:]) + G_R[:, :6]
    assert np.abs(dyn).max() < 1e-7  # J_CR^T[:6] = J_C^T[:6]: base columns
    assert np.abs(y[:, 6:RS]).max() <= H.LQP_ACC_LIM + 1e-5
    assert np.abs(hn.y_ans(1)).max() <= H.LQP_NC_ACC_LIM + 1e-5
    acc, tq, f = jr["acc_qp"], jr["torque_qp"], jr["contact_qp"]
    JCR = np.zeros((B, 12, RS))
    JCR[:, :, :VCD] = JC[:, :, :VCD]
    dj = np.einsum("bij,bj->bi", A_R, acc) + np.einsum("bji,bj->bi", JCR, f) + G_R
    dj[:, 6:] -= tq
    assert np.abs(dj).max() < 1e-6 and np.abs(np.einsum("bij,bj->bi", JCR, acc)).max() < 1e-6
    assert np.abs(acc[:, 6:]).max() <= H.JACC_ACC_LIM + 1e-5 and np.abs(tq[:, :12]).max() <= H.JACC_TAU_LIM + 1e-4
    # ---- against the restatement
    for b in range(NS):
        o = _oracle(q[b], fs[b])
        r = o["r"]
        assert np.abs(A_R[b] - r.A_R).max() < 1e-8 * np.abs(r.A_R).max() and np.abs(A_Ri[b] - r.A_R_inv).max() < 1e-9
        assert np.abs(G_R[b] - r.G_R).max() < 1e-8 and np.abs(JI[b, :, :NCD] - r.J_I_nc).max() < 1e-10 and np.abs(JIi[b, :, :NCD] - r.J_I_nc_inv_T).max() < 1e-8
        for lv, h in enumerate(o["lqp"].hqp_hs_):
            assert _rel(hr.y_ans(lv)[b], h.y_ans_) < 2e-6, (b, lv)
        assert _rel(tau[b], o["lqp_tau"]) < 1e-4
        for lv, h in enumerate(o["nc"].hqp_hs_):
            assert _rel(hn.y_ans(lv)[b], h.y_ans_) < 1e-4, (b, lv)
        j, n = o["jacc"], o["jacc_nc"]
        assert _rel(acc[b], j["acc"]) < 1e-5 and _rel(tq[b], j["tau"]) < 1e-3 and _rel(f[b], j["f"]) < 1e-3 and _rel(jr["f_star_qp"][b], j["s"]) < 1e-5
        assert _rel(nr["acc_qp"][b], n["acc"]) < 1e-4 and _rel(nr["torque_qp"][b], n["tau"]) < 1e-4
        assert np.abs(nr["gacc_qp"][b] - n["gacc"]).max() < 1e-4 and np.abs(nr["f_star_qp"][b] - n["s"]).max() < 1e-4

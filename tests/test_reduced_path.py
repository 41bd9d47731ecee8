"""Reduced (centroidal) dynamics path -- SURVEY.md §8 row a15, BASELINE config 5.

PARITY UNPINNED in the reference (no fixture, no assertion: SURVEY §8c); what is checked here:
  * the numpy restatement oracle/dwbc_reduced_np.py satisfies the identities the reference relies on
    (A_R_inv = J_R A^-1 J_R^T, J_R_INV_T J_R^T = I) and reproduces the reference's own implicit check
    "reduced ~ full" (tests/sp_test/redu_dyn_test.cpp:304-317): gravity and task torques of the reduced model equal
    the full model's, on the reference's CASE 1/2 states (whose full-model torques are pinned by the goldens);
  * not-gpu: the kernel source (libdwbc_amd/csrc/dwbc_reduced.h) compiled for the host (tests/emu) against that
    restatement; gpu: the HIP kernel through the C-ABI (DWBC_SOLVE_REDUCED) against it.
"""
import numpy as np
import pytest

from oracle.dwbc_np import Cycle
from oracle.dwbc_reduced_np import ReducedCycle
from tests import cases

TOL_TAU = 1e-6  # BASELINE north star: tau within 1e-6 (fp64)


def make(cls, tasks=cases.TASKS_2LEVEL):
    c = cls(cases.tocabi_model())
    for cc in cases.CONTACTS_2:
        c.add_contact(cc["link"], cc["point"], cc["lx"], cc["ly"], cc["mu"], cc["muz"])
    for lv, links in enumerate(tasks):
        for mode, link, pt in links:
            c.add_task(lv, mode, link, pt)
    return c


def split_fstar(fs, tasks):
    out, off = [], 0
    for links in tasks:
        t = sum(6 if m <= 2 else 3 for m, _, _ in links)
        out.append(np.asarray(fs[off : off + t], float))
        off += t
    return out


def oracle_batch(q, fl, fs, tasks=cases.TASKS_2LEVEL):
    B = q.shape[0]
    tau = np.zeros((B, 3, 33))
    wr = np.zeros((B, 12))
    st = np.zeros(B, np.int32)
    for b in range(B):
        c = make(ReducedCycle, tasks)
        try:
            tt = c.run_reduced(q[b], list(fl[b]), split_fstar(fs[b], tasks))
        except ValueError:  # outside the reduced path's scope (contact chains not on the leading dofs)
            continue
        st[b] = c.status
        tau[b, 0] = c.tau_grav
        if c.status or hasattr(c, "tau_task"):
            tau[b, 1] = getattr(c, "tau_task", 0.0)
        tau[b, 2] = c.tau_contact
        w = c.contact_force(tau[b].sum(axis=0))
        wr[b, : w.size] = w
    return tau, wr, st


@pytest.mark.parametrize("case", [1, 2])
def test_reduced_oracle_identities_and_full_model_agreement(case):
    q = np.array(cases.Q_CASE[case])
    fs = [np.array(f) for f in cases.FSTAR_CASE[case]]
    full = make(Cycle)
    full.run(q, [1, 1], fs)
    red = make(ReducedCycle)
    red.run_reduced(q, [1, 1], fs)
    assert full.status == 1 and red.status == 1
    assert (red.r_sys_dof, red.r_model_dof, red.nc_dof, red.co_dof) == (24, 18, 21, 12)
    assert np.abs(red.A_R_inv - red.J_R @ red.A_inv @ red.J_R.T).max() < 1e-12  # dwbc.cpp:2932-2956
    assert np.abs(red.J_R_INV_T @ red.J_R.T - np.eye(24)).max() < 1e-10
    assert np.abs(red.A_R_inv_N_CR - red.J_R @ (red.A_inv @ red.N_C) @ red.J_R.T).max() < 1e-12
    # mass / COM of the non-contact body group from the masked composite pass (dwbc.cpp:2828-2888)
    m = cases.tocabi_model()
    assert abs(red.mass_nc - m["mass"][13:].sum()) < 1e-9
    # the reference's own check: reduced ~ full (redu_dyn_test.cpp:304-317).  Gravity and task torques agree to round-off
    # (the reduction is exact); the contact torque differs by design: CalcContactRedistributeR minimises the tangential
    # contact wrench (H = H_temp^T H_temp, dwbc.cpp:4846-4848), CalcContactRedistribute minimises |c| (dwbc.cpp:1458).
    assert np.abs(full.tau_grav - red.tau_grav).max() < 1e-9
    assert np.abs(full.tau_task - red.tau_task).max() < 1e-8
    # pinned through the full model: torque_grav_ golden
    assert np.abs(red.tau_grav - cases.golden(case, "torque_grav_")[:, 0]).max() < 1e-8
    # the redistributed wrench respects the friction / CoP cones
    F = red.contact_force(red.tau_grav + red.tau_task + red.tau_contact)
    assert (red.cone_matrix() @ F > -1e-6).all()


@pytest.mark.parametrize("cfg", ["ds", "ds_yaw", "ss_L"])
def test_emulated_reduced_kernel_vs_oracle(cfg):
    from tests.emu.emu import Emu

    B = 12
    kw = dict(seed=4321)
    if cfg == "ds_yaw":
        kw["yaw"] = True
    elif cfg == "ss_L":
        kw["contact_mode"] = "L"
    q, fl, fs = cases.synth_batch(B, **kw)
    q[0] = cases.Q_CASE[1]
    fs[0] = list(cases.FSTAR_CASE[1][0]) + list(cases.FSTAR_CASE[1][1])
    e = Emu(cases.URDF, cases.CONTACTS_2, cases.TASKS_2LEVEL, None)
    r = e.run(q, fl, fs, reduced=True)
    tau, wr, st = oracle_batch(q, fl, fs)
    assert (r["status"] == st).all()
    ok = st == 1
    assert ok.mean() > 0.5
    assert np.abs(r["tau"][ok] - tau[ok]).max() < TOL_TAU
    assert np.abs(r["wrench"][ok] - wr[ok]).max() < 1e-5
    # gravity torque does not depend on the QPs: it must match on every instance
    assert np.abs(r["tau"][:, 0] - tau[:, 0]).max() < 1e-8


HIERARCHIES = {
    # reference tests/sp_test/redu_dyn_test.cpp:98-103 without its COM level: pelvis, upper body (first non-contact level),
    # right hand (a later non-contact level: the null_force_ branch, dwbc.cpp:3311-3325)
    "pelvis_rot-upper-rhand": ([[(6, 0, (0, 0, 0))], [(6, 15, (0, 0, 0))], [(0, 33, (0, 0, 0))]], 12),
    "pelvis6d-upper-rhand-lhandpos": ([[(0, 0, (0, 0, 0))], [(6, 15, (0, 0, 0))], [(0, 33, (0, 0, 0))], [(3, 23, (0, 0, 0))]], 18),
    # the reference harness itself (redu_dyn_test.cpp:98-103): COM position (a "cmm" level, task.cpp:106-114), pelvis rotation,
    # upper-body rotation, right hand 6D
    "com_pos-pelvis_rot-upper-rhand": ([[(3, 34, (0, 0, 0))], [(6, 0, (0, 0, 0))], [(6, 15, (0, 0, 0))], [(0, 33, (0, 0, 0))]], 15),
    # a contact-chain level after the non-contact one (Null_task_R_ is carried over it, dwbc.cpp:3247-3250)
    "pelvis_rot-upper-hiproll_pos": ([[(6, 0, (0, 0, 0))], [(6, 15, (0, 0, 0))], [(3, 1, (0, 0, 0))]], 9),
}


@pytest.mark.parametrize("name", sorted(HIERARCHIES))
def test_emulated_reduced_kernel_task_hierarchies(name):
    from tests.emu.emu import Emu

    tasks, nf = HIERARCHIES[name]
    B = 6
    q, fl, _ = cases.synth_batch(B, seed=11)
    fs = 2.5 * np.random.default_rng(3).uniform(-1, 1, size=(B, nf))
    e = Emu(cases.URDF, cases.CONTACTS_2, tasks, None)
    r = e.run(q, fl, fs, reduced=True)
    tau, wr, st = oracle_batch(q, fl, fs, tasks)
    assert (r["status"] == st).all() and st.mean() > 0.6
    ok = st == 1
    assert np.abs(tau[:, 1]).max() > 1.0  # the hierarchy does something
    assert np.abs(r["tau"][ok] - tau[ok]).max() < TOL_TAU
    assert np.abs(r["wrench"][ok] - wr[ok]).max() < 1e-5


def test_emulated_reduced_kernel_rejects_out_of_scope():
    from tests.emu.emu import Emu

    q, fl, fs = cases.synth_batch(2, seed=1, contact_mode="R")  # right-foot chain = dofs 12..17: not the leading dofs
    e = Emu(cases.URDF, cases.CONTACTS_2, cases.TASKS_2LEVEL, None)
    r = e.run(q, fl, fs, reduced=True)
    assert (r["status"] == 0).all() and np.abs(r["tau"]).max() == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["ds", "ds_yaw", "mixed_L"])
def test_gpu_reduced_kernel_vs_oracle(cfg):
    import libdwbc_amd as D

    B = 24
    kw = dict(seed=99)
    if cfg == "ds_yaw":
        kw["yaw"] = True
    q, fl, fs = cases.synth_batch(B, **kw)
    if cfg == "mixed_L":
        fl[::3, 1] = 0  # every third instance in left single support
    model = D.Model.from_urdf(cases.URDF)
    wbc = D.Batch(model, B, device=0)
    for c in cases.CONTACTS_2:
        wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    wbc.add_task(0, D.TASK_LINK_6D, 0)
    wbc.add_task(1, D.TASK_LINK_ROTATION, 15)
    wbc.set_state(q)
    wbc.set_contact(fl)
    wbc.set_fstar_all(fs)
    wbc.solve(reduced=True)
    tau_g, wr_g, st_g = wbc.get("tau"), wbc.get("wrench"), wbc.get("status")
    tau, wr, st = oracle_batch(q, fl, fs)
    assert (st_g == st).all()
    ok = st == 1
    assert ok.mean() > 0.5
    assert np.abs(tau_g[ok] - tau[ok]).max() < TOL_TAU
    assert np.abs(wr_g[ok] - wr[ok]).max() < 1e-5
    # with a torque limit set the reduced path is refused loudly (reference App. C-10)
    wbc.set_torque_limit(np.array(cases.TAU_LIM))
    with pytest.raises(RuntimeError):
        wbc.solve(reduced=True)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(HIERARCHIES))
def test_gpu_reduced_task_hierarchies(name):
    import libdwbc_amd as D

    tasks, nf = HIERARCHIES[name]
    B = 16
    q, fl, _ = cases.synth_batch(B, seed=12)
    fs = 2.0 * np.random.default_rng(5).uniform(-1, 1, size=(B, nf))
    wbc = D.Batch(D.Model.from_urdf(cases.URDF), B, device=0)
    for c in cases.CONTACTS_2:
        wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    for lv, links in enumerate(tasks):
        for mode, link, pt in links:
            wbc.add_task(lv, mode, link, pt)
    wbc.set_state(q)
    wbc.set_contact(fl)
    wbc.set_fstar_all(fs)
    wbc.solve(reduced=True)
    tau, wr, st = oracle_batch(q, fl, fs, tasks)
    assert (wbc.get("status") == st).all() and st.mean() > 0.5
    ok = st == 1
    assert np.abs(wbc.get("tau")[ok] - tau[ok]).max() < TOL_TAU


@pytest.mark.gpu
def test_gpu_reduced_full_batch_properties():
    """BASELINE config 5 shape on one GPU (8192 of the 65536 instances one rank would own): size-independent properties
    -- gravity and task torque of the reduced model equal the full model's on the same batch (the reduction is exact and
    neither depends on the redistribution objective), wrenches of solved instances lie in the cones."""
    import libdwbc_amd as D

    B = 8192
    q, fl, fs = cases.synth_batch(B, seed=555)
    model = D.Model.from_urdf(cases.URDF)
    out = {}
    for reduced in (False, True):
        wbc = D.Batch(model, B, device=0)
        for c in cases.CONTACTS_2:
            wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
        wbc.add_task(0, D.TASK_LINK_6D, 0)
        wbc.add_task(1, D.TASK_LINK_ROTATION, 15)
        wbc.set_state(q)
        wbc.set_contact(fl)
        wbc.set_fstar_all(fs)
        wbc.solve(reduced=reduced)
        out[reduced] = (wbc.get("tau"), wbc.get("wrench"), wbc.get("status"))
    (tf, wf, sf), (tr, wrr, sr) = out[False], out[True]
    assert sf.mean() > 0.95 and sr.mean() > 0.7
    assert np.abs(tf[:, 0] - tr[:, 0]).max() < 1e-7
    both = (sf == 1) & (sr == 1)
    # the task QPs of the two models are the same problems up to round-off; the few instances whose QP sits on the
    # acceptance threshold of the lexicographic point (DESIGN.md "QP canon", step 2 vs 3) may take the other branch,
    # which moves delta by the Tikhonov weight (~1e-4, the reference's own qpOASES slack is 8.5e-4)
    err = np.abs(tf[both, 1] - tr[both, 1]).max(axis=1)
    assert np.quantile(err, 0.99) < 1e-6 and err.max() < 5e-3
    # cone check on the reduced wrenches: f_z < 0 under load, |f_x|,|f_y| <= mu |f_z| in the contact frame is frame
    # dependent; the world-frame normal force sign is not (flat feet in the synthetic batch)
    assert (wrr[sr == 1][:, 2] < 0).all() and (wrr[sr == 1][:, 8] < 0).all()

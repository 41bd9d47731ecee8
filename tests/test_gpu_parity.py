"""-m gpu: the HIP path (through the C-ABI, libdwbc_hip.so) against the oracle and the reference goldens.

Tolerances (BASELINE.json north_star: tau within 1e-6 of the reference, fp64):
  * vs oracle (same canonical QP solution): 1e-6 max-abs on tau and on the contact wrench
  * vs reference goldens: tau_grav/tau_task 1e-6; tau_contact 1e-8 (case 1) / 1e-3 (case 2, qpOASES' own
    regularisation slack)
"""
import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu

TOL = 1e-6


def _make(B, contacts=cases.CONTACTS_2, tasks=cases.TASKS_2LEVEL, tau_lim=cases.TAU_LIM):
    import libdwbc_amd as D

    model = D.Model.from_urdf(cases.URDF)
    wbc = D.Batch(model, B, device=0)
    for c in contacts:
        wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    for lv, links in enumerate(tasks):
        for mode, link, pt in links:
            wbc.add_task(lv, mode, link, pt)
    wbc.set_torque_limit(None if tau_lim is None else np.array(tau_lim))
    return wbc


def _oracle(B, q, flags, fstar, contacts=cases.CONTACTS_2, tasks=cases.TASKS_2LEVEL, tau_lim=cases.TAU_LIM):
    from oracle import orc

    M = orc.make_model(cases.tocabi_model())
    S = orc.make_setup(contacts, tasks, tau_lim)
    return orc.cycle_batch(M, S, q, flags, fstar, 0)


def _run(wbc, q, flags, fstar):
    wbc.set_state(q)
    wbc.set_contact(flags)
    wbc.set_fstar_all(fstar)
    wbc.solve()
    return wbc.get("tau"), wbc.get("wrench"), wbc.get("status")


@pytest.mark.parametrize("case", [1, 2])
def test_golden_cases_through_c_abi(case):
    """reference tests/dwbc_test.cpp CASE 1 / CASE 2 (four registered contacts, two enabled)."""
    wbc = _make(1, contacts=cases.CONTACTS_4)
    wbc.enable_dump(True)
    q = np.array([cases.Q_CASE[case]], dtype=np.float64)
    flags = np.array([[1, 1, 0, 0]], dtype=np.uint8)
    fstar = np.array([list(cases.FSTAR_CASE[case][0]) + list(cases.FSTAR_CASE[case][1])])
    tau, wr, st = _run(wbc, q, flags, fstar)
    g = lambda n: cases.golden(case, n)
    e = lambda a, b: float(np.abs(a - b).max())
    assert st[0] == 1
    assert e(wbc.get("A")[0], g("Acontact_mat")) < 1e-9
    assert e(wbc.get("A_inv")[0], g("A_inv_")) < 1e-8
    assert e(wbc.get("J_C")[0], g("J_C")) < 1e-12
    assert e(wbc.get("Lambda_c")[0].reshape(12, 12), g("Lambda_contact")) < 1e-8
    assert e(wbc.get("J_C_INV_T")[0], g("J_C_INV_T")) < 1e-9
    assert e(wbc.get("A_inv_N_C")[0][6:, 6:], g("W")) < 1e-8
    assert e(wbc.get("W_inv")[0], g("W_inv")) < 1e-7
    assert e(wbc.get("NwJw")[0], g("NwJw")) < 1e-9
    assert e(tau[0, 0], g("torque_grav_")[:, 0]) < TOL
    assert e(tau[0, 1], g("torque_task_")[:, 0]) < TOL
    assert e(tau[0, 2], g("torque_contact_")[:, 0]) < (1e-8 if case == 1 else 1e-3)


def test_case3_yaw_invariance_on_device():
    """reference tests/dwbc_test.cpp:262-361."""
    from oracle.dwbc_np import quat_to_R

    qu = cases.yaw_quat(np.pi / 2)
    q = np.array([cases.Q_CASE[2]], dtype=np.float64)
    q[0, 3:6] = qu[:3]
    q[0, 39] = qu[3]
    Rz = quat_to_R(*qu)
    f1, f2 = np.array(cases.FSTAR_CASE[2][0]), np.array(cases.FSTAR_CASE[2][1])
    fstar = np.concatenate([Rz @ f1[:3], Rz @ f1[3:], Rz @ f2])[None, :]
    wbc = _make(1, contacts=cases.CONTACTS_4)
    tau, wr, st = _run(wbc, q, np.array([[1, 1, 0, 0]], dtype=np.uint8), fstar)
    g = lambda n: cases.golden(2, n)[:, 0]
    assert st[0] == 1
    assert np.abs(tau[0, 0] - g("torque_grav_")).max() < TOL
    assert np.abs(tau[0, 1] - g("torque_task_")).max() < TOL
    assert np.abs(tau[0, 2] - g("torque_contact_")).max() < 1e-3


@pytest.mark.parametrize("yaw", [False, True])
def test_double_support_batch_vs_oracle(yaw):
    """BASELINE configs[1] at its full size (batch = 1024, double support, 2-level HQP, tau limit): every instance against the
    oracle (the C restatement does 1024 cycles in a fraction of a second)."""
    B = 1024
    q, flags, fstar = cases.synth_batch(B, seed=20251226 + 2, yaw=yaw)
    wbc = _make(B)
    tau, wr, st = _run(wbc, q, flags, fstar)
    tau_r, wr_r, st_r, _ = _oracle(B, q, flags, fstar)
    assert (st == st_r).all()
    ok = st_r == 1
    assert ok.mean() > 0.9 and (ok.all() or yaw)  # tilted bases (roll/pitch +-0.1) make a few cone QPs infeasible
    assert np.abs(tau[ok] - tau_r[ok]).max() < TOL
    assert np.abs(wr[ok] - wr_r[ok]).max() < 1e-5  # wrench ~ 1e3 N: 1e-8 relative
    # the QPs really bite on this distribution: at least one active constraint somewhere
    assert wbc.get("diag")[:, 9:12].sum() > 0


@pytest.mark.parametrize("side", ["L", "R"])
def test_single_support_three_levels_vs_oracle(side):
    """BASELINE config 3 shape: single support + swing foot task as third level (k = 0: strictly convex QPs)."""
    B = 256
    tasks = cases.TASKS_3LEVEL_SWING_R if side == "L" else cases.TASKS_3LEVEL_SWING_L
    q, flags, fstar = cases.synth_batch(B, seed=20251226 + 3, contact_mode=side, levels=3)
    wbc = _make(B, tasks=tasks)
    tau, wr, st = _run(wbc, q, flags, fstar)
    tau_r, wr_r, st_r, _ = _oracle(B, q, flags, fstar, tasks=tasks)
    assert (st == st_r).all()
    ok = st_r == 1
    assert ok.mean() > 0.5
    assert np.abs(tau[ok] - tau_r[ok]).max() < TOL
    assert np.abs(tau[ok, 2]).max() == 0.0  # torque_contact_ = 0 when k = 0 (reference src/dwbc.cpp:1562-1567)


def test_mixed_contact_modes_vs_oracle():
    """BASELINE config 4 shape: per-instance contact flags (LR / L / R) inside one launch."""
    B = 384
    q, flags, fstar = cases.synth_batch(B, seed=20251226 + 4, contact_mode="mixed")
    assert len({tuple(f) for f in flags}) == 3
    wbc = _make(B)
    tau, wr, st = _run(wbc, q, flags, fstar)
    tau_r, wr_r, st_r, _ = _oracle(B, q, flags, fstar)
    assert (st == st_r).all()
    ok = st_r == 1
    assert np.abs(tau[ok] - tau_r[ok]).max() < TOL


def test_no_torque_limit_main_cpp_config():
    """BASELINE config 1a (reference example/main.cpp:62-108): nominal stance, f*0=(0.1,2,0.1,0,0,0), f*1=0, no tau limit."""
    q = np.array([cases.Q_CASE[1]], dtype=np.float64)
    fstar = np.array([[0.1, 2.0, 0.1, 0, 0, 0, 0, 0, 0]], dtype=np.float64)
    flags = np.array([[1, 1]], dtype=np.uint8)
    wbc = _make(1, tau_lim=None)
    tau, wr, st = _run(wbc, q, flags, fstar)
    tau_r, wr_r, st_r, _ = _oracle(1, q, flags, fstar, tau_lim=None)
    assert st[0] == 1 and st_r[0] == 1
    assert np.abs(tau - tau_r).max() < TOL


def test_full_size_properties_without_oracle():
    """At BASELINE config-2 size (B = 1024): size-independent properties.
    (a) yaw invariance: rotating the base by a yaw and f* with it leaves every torque unchanged (CASE 3 as a property);
    (b) the returned contact wrench satisfies the friction / CoP cones it was constrained to;
    (c) |tau_total| <= tau limit."""
    from oracle.dwbc_np import quat_to_R

    B = 1024
    q, flags, fstar = cases.synth_batch(B, seed=99)
    wbc = _make(B)
    tau0, wr0, st0 = _run(wbc, q, flags, fstar)
    assert st0.all()
    rng = np.random.default_rng(5)
    q2, f2 = q.copy(), fstar.copy()
    for b in range(B):
        qu = cases.yaw_quat(rng.uniform(-np.pi, np.pi))
        R = quat_to_R(*qu)
        q2[b, 3:6], q2[b, 39] = qu[:3], qu[3]
        q2[b, 0:3] = R @ q[b, 0:3]
        f2[b, 0:3], f2[b, 3:6], f2[b, 6:9] = R @ fstar[b, 0:3], R @ fstar[b, 3:6], R @ fstar[b, 6:9]
    tau1, wr1, st1 = _run(wbc, q2, flags, f2)
    assert st1.all()
    assert np.abs(tau1 - tau0).max() < TOL
    total = tau0.sum(axis=1)
    assert (np.abs(total) <= 300.0 + 1e-6).all()
    # cones in the contact frame: nominal feet are flat and yaw-aligned with the pelvis for this distribution up to
    # 0.01 rad joint noise, so check in the world frame with a matching slack on the un-rotated states
    mu, lx, ly = 0.2, 0.15, 0.075
    for a in range(2):
        f, mo = wr0[:, 6 * a : 6 * a + 3], wr0[:, 6 * a + 3 : 6 * a + 6]
        fz = -f[:, 2]
        assert (fz > 0).all()
        slack = 0.05 * fz + 1e-6
        assert (np.abs(f[:, 0]) <= mu * fz + slack).all() and (np.abs(f[:, 1]) <= mu * fz + slack).all()
        assert (np.abs(mo[:, 1]) <= lx * fz + slack).all() and (np.abs(mo[:, 0]) <= ly * fz + slack).all()


def test_bound_torch_tensors_zero_copy():
    """PyTorch owns device memory and the stream; results land in the bound tensors without a host round trip."""
    import torch

    B = 64
    q, flags, fstar = cases.synth_batch(B, seed=11)
    dev = torch.device("cuda:0")
    tq = torch.from_numpy(q).to(dev)
    tf = torch.from_numpy(flags).to(dev)
    ts = torch.from_numpy(fstar).to(dev)
    ttau = torch.zeros((B, 3, 33), dtype=torch.float64, device=dev)
    twr = torch.zeros((B, 12), dtype=torch.float64, device=dev)
    tst = torch.zeros((B,), dtype=torch.int32, device=dev)
    wbc = _make(B)
    for name, t in (("in_q", tq), ("in_contact", tf), ("in_fstar", ts), ("tau", ttau), ("wrench", twr), ("status", tst)):
        wbc.bind_tensor(name, t)
    wbc.set_stream(torch.cuda.current_stream().cuda_stream)
    wbc.solve()
    torch.cuda.synchronize()
    tau_r, wr_r, st_r, _ = _oracle(B, q, flags, fstar)
    assert (tst.cpu().numpy() == st_r).all()
    assert np.abs(ttau.cpu().numpy() - tau_r).max() < TOL


def test_centroidal_outputs_on_device():
    """CMM_, com_pos, COM inertia, jac_com_ (reference src/dwbc.cpp:318-352) against the numpy restatement."""
    from oracle import dwbc_np

    B = 4
    q, fl, fs = cases.synth_batch(B, seed=91, yaw=True)
    wbc = _make(B)
    wbc.enable_dump(True)
    _run(wbc, q, fl, fs)
    cmm, com, jc, ic = wbc.get("CMM"), wbc.get("com"), wbc.get("J_com"), wbc.get("com_inertia")
    for i in range(B):
        cy = dwbc_np.Cycle(cases.tocabi_model())
        cy.update_kinematics(q[i])
        assert np.abs(cmm[i] - cy.CMM).max() < 1e-10
        assert np.abs(com[i] - cy.com).max() < 1e-12
        assert np.abs(jc[i] - cy.J_com).max() < 1e-10
        assert np.abs(ic[i] - ic[i].T).max() < 1e-10 and np.all(np.linalg.eigvalsh(ic[i]) > 0)


@pytest.mark.parametrize("cfg", ["config3_ss_3level_8192", "config4_mixed_65536"])
def test_full_size_configs_3_and_4(cfg):
    """BASELINE configs[2] / configs[3] at their full batch sizes.  Size-independent properties:
    (a) a random sample of 192 instances of the big launch agrees with the oracle to 1e-6 and with the same states
        solved in a small batch (instances are independent; the small batch runs the wide-register kernel build, the
        big one the register-capped build);
    (b) every instance returns a status, |tau_total| <= tau limit where the cycle succeeded, tau_contact = 0 when k = 0."""
    if cfg.startswith("config3"):
        B, tasks, kw = 8192, cases.TASKS_3LEVEL_SWING_R, dict(seed=20251226 + 3, contact_mode="L", levels=3)
    else:
        B, tasks, kw = 65536, cases.TASKS_2LEVEL, dict(seed=20251226 + 4, contact_mode="mixed")
    q, flags, fstar = cases.synth_batch(B, **kw)
    wbc = _make(B, tasks=tasks)
    tau, wr, st = _run(wbc, q, flags, fstar)
    assert set(np.unique(st)) <= {0, 1} and st.mean() > 0.9
    ok = st == 1
    assert (np.abs(tau[ok].sum(axis=1)) <= 300.0 + 1e-6).all()
    ss = flags.sum(axis=1) == 1
    assert np.abs(tau[ss & ok, 2]).max() == 0.0
    idx = np.random.default_rng(11).choice(B, size=192, replace=False)
    small = _make(192, tasks=tasks)
    tau_s, wr_s, st_s = _run(small, q[idx], flags[idx], fstar[idx])
    assert (st_s == st[idx]).all()
    assert np.abs(tau_s - tau[idx]).max() < 1e-8
    tau_r, wr_r, st_r, _ = _oracle(192, q[idx], flags[idx], fstar[idx], tasks=tasks)
    assert (st_r == st[idx]).all()
    okr = st_r == 1
    assert np.abs(tau[idx][okr] - tau_r[okr]).max() < TOL


@pytest.mark.gpu
def test_gpu_com_task_hierarchy_vs_oracle():
    """COM position + pelvis rotation + upper-body rotation: the COM link's Jacobian is jac_com_ (reference
    src/dwbc.cpp:352-353); link id = link_num_ = 34, also reachable as model.link_id("COM")"""
    import libdwbc_amd as D
    from oracle import orc

    B = 64
    tasks = [[(3, 34, (0, 0, 0))], [(6, 0, (0, 0, 0))], [(6, 15, (0, 0, 0))]]
    q, fl, _ = cases.synth_batch(B, seed=77, yaw=True)
    fs = 1.5 * np.random.default_rng(7).uniform(-1, 1, size=(B, 9))
    model = D.Model.from_urdf(cases.URDF)
    assert model.link_id("COM") == 34
    wbc = D.Batch(model, B, device=0)
    for c in cases.CONTACTS_2:
        wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    for lv, links in enumerate(tasks):
        for mode, link, pt in links:
            wbc.add_task(lv, mode, link, pt)
    wbc.set_torque_limit(np.array(cases.TAU_LIM))
    wbc.set_state(q)
    wbc.set_contact(fl)
    wbc.set_fstar_all(fs)
    wbc.solve()
    M = orc.make_model(cases.tocabi_model())
    S = orc.make_setup(cases.CONTACTS_2, tasks, cases.TAU_LIM)
    tau, wr, st, _ = orc.cycle_batch(M, S, q, fl, fs, 0)
    assert (wbc.get("status") == st).all() and st.mean() > 0.9
    ok = st == 1
    assert np.abs(wbc.get("tau")[ok] - tau[ok]).max() < 1e-6


@pytest.mark.gpu
def test_gpu_custom_task_level_vs_oracle():
    """AddTaskSpace(h, TASK_CUSTOM, dof) + SetTaskSpace(h, f*, J) (reference include/dwbc.h:318,333) through the C-ABI"""
    import libdwbc_amd as D
    from tests.test_kernel_emulation import _custom_case

    B = 16
    q, fl, fs, J, tau, st = _custom_case(B, 43)
    wbc = D.Batch(D.Model.from_urdf(cases.URDF), B, device=0)
    for c in cases.CONTACTS_2:
        wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    wbc.add_task(0, D.TASK_LINK_6D, 0)
    wbc.add_custom_task(1, 3)
    wbc.set_torque_limit(np.array(cases.TAU_LIM))
    wbc.set_state(q)
    wbc.set_contact(fl)
    wbc.set_fstar(0, fs[:, :6])
    wbc.set_custom_task(1, fs[:, 6:], J)
    wbc.solve()
    assert (wbc.get("status") == st).all() and st.all()
    assert np.abs(wbc.get("tau") - tau).max() < 1e-6
    with pytest.raises(RuntimeError):
        wbc.solve(reduced=True)


@pytest.mark.gpu
def test_gpu_copy_kinematics_data():
    """RobotData::CopyKinematicsData (reference src/dwbc.cpp:1711-1762, the cross-thread hand-off of tests/test_thread.cpp):
    the target batch reproduces the source's torques from the copied state / contacts / task spaces"""
    import libdwbc_amd as D

    B = 32
    q, fl, fs = cases.synth_batch(B, seed=88, contact_mode="mixed")
    model = D.Model.from_urdf(cases.URDF)
    src = D.Batch(model, B, device=0)
    for c in cases.CONTACTS_2:
        src.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    src.add_task(0, D.TASK_LINK_6D, 0)
    src.add_task(1, D.TASK_LINK_ROTATION, 15)
    src.set_torque_limit(np.array(cases.TAU_LIM))
    src.set_state(q)
    src.set_contact(fl)
    src.set_fstar_all(fs)
    src.solve()
    dst = D.Batch(model, B, device=0)
    src.copy_kinematics_to(dst)
    dst.solve()
    assert (dst.get("status") == src.get("status")).all()
    assert np.abs(dst.get("tau") - src.get("tau")).max() == 0.0
    assert np.abs(dst.get("wrench") - src.get("wrench")).max() == 0.0


@pytest.mark.gpu
def test_gpu_hqp_false_vs_oracle():
    """hqp = false: plain hierarchy + closed-form redistribution (reference src/dwbc.cpp:856-873, 1570-1619)"""
    import libdwbc_amd as D
    from tests.test_kernel_emulation import _no_hqp_oracle

    B = 32
    q, fl, fs = cases.synth_batch(B, seed=72, yaw=True)
    wbc = D.Batch(D.Model.from_urdf(cases.URDF), B, device=0)
    for c in cases.CONTACTS_2:
        wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    wbc.add_task(0, D.TASK_LINK_6D, 0)
    wbc.add_task(1, D.TASK_LINK_ROTATION, 15)
    wbc.set_state(q)
    wbc.set_contact(fl)
    wbc.set_fstar_all(fs)
    wbc.solve(hqp=False)
    tau, st = _no_hqp_oracle(q, fl, fs)
    assert (wbc.get("status") == st).all()
    assert np.abs(wbc.get("tau") - tau).max() < 1e-6
    wbc.solve(hqp=True)  # and back
    assert wbc.get("status").mean() > 0.9


@pytest.mark.gpu
def test_gpu_generic_tree_instantiation_matches_tocabi_one(monkeypatch):
    """A model whose tree is not the constant TOCABI one runs the TopoGeneric instantiation (dense A^-1 sweep).  The test hook
    DWBC_DENSE_SWEEP makes a TOCABI batch take that route: same torques to rounding, full and reduced dynamics, and the
    launcher reports the other kernel."""
    B = 256
    q, flags, fstar = cases.synth_batch(B, seed=4242, yaw=True)
    w1 = _make(B)
    tau1, wr1, st1 = _run(w1, q, flags, fstar)
    assert "TopoTocabi" in w1.kernel_name()
    monkeypatch.setenv("DWBC_DENSE_SWEEP", "1")
    w2 = _make(B)
    monkeypatch.delenv("DWBC_DENSE_SWEEP")
    tau2, wr2, st2 = _run(w2, q, flags, fstar)
    assert "TopoGeneric" in w2.kernel_name()
    assert (st1 == st2).all() and (st1 == 1).mean() > 0.5
    ok = st1 == 1
    assert np.abs(tau1[ok] - tau2[ok]).max() < 1e-7
    assert np.abs(wr1[ok] - wr2[ok]).max() < 1e-6
    tau_r, wr_r, st_r, _ = _oracle(B, q, flags, fstar)
    ok = ok & (st_r == 1)
    assert np.abs(tau2[ok] - tau_r[ok]).max() < TOL
    for w in (w1, w2):  # reduced dynamics (no torque limit on that path)
        w.set_torque_limit(None)
        w.solve(reduced=True)
    assert "TopoGeneric" in w2.kernel_name() and "reduced" in w2.kernel_name()
    s1, s2 = w1.get("status"), w2.get("status")
    ok = (s1 == 1) & (s2 == 1)
    assert ok.mean() > 0.5
    d = np.abs(w1.get("tau")[ok] - w2.get("tau")[ok]).max(axis=(1, 2))
    assert np.quantile(d, 0.99) < 1e-6


def test_more_than_two_active_contacts_fail_loudly():
    """The device path stacks at most two simultaneous 6D contacts (the reference: any number, src/dwbc.cpp:445-453).  An
    instance with three flags is never solved with a subset: the host entry point refuses the flags, and flags that reach the
    kernel through a bound device buffer give status 0 with zero torques and wrench for that instance only."""
    import torch

    import libdwbc_amd as D

    B = 8
    wbc = _make(B, contacts=cases.CONTACTS_4)
    q, fl2, fs = cases.synth_batch(B, seed=31)
    flags = np.zeros((B, 4), np.uint8)
    flags[:, :2] = 1
    bad = flags.copy()
    bad[3] = [1, 1, 1, 0]
    wbc.set_state(q)
    wbc.set_fstar_all(fs)
    with pytest.raises(D.batch.DwbcError, match="more than 2"):
        wbc.set_contact(bad)
    # device-resident flags cannot be inspected by the host: the kernel fails the instance
    tf = torch.from_numpy(bad).cuda()
    wbc.bind_tensor("in_contact", tf)
    wbc.solve()
    tau, wr, st = wbc.get("tau"), wbc.get("wrench"), wbc.get("status")
    assert st[3] == 0 and np.abs(tau[3]).max() == 0.0 and np.abs(wr[3]).max() == 0.0
    tau_o, wr_o, st_o, _ = _oracle(B, q, flags[:, :2].copy(), fs)
    keep = np.arange(B) != 3
    assert (st[keep] == st_o[keep]).all() and st[keep].all()
    assert np.abs(tau[keep] - tau_o[keep]).max() < TOL


def test_zmp_and_contact_frames_on_device():
    """getZMP(getContactForce(tau_total)) and cc_[i].xc_pos / rotm / zmp_pos (reference src/dwbc.cpp:898-939,
    src/contact_constraint.cpp:53-54) from the dump record of the HIP kernel, against the numpy restatement."""
    from oracle.dwbc_np import Cycle

    B = 6
    q, fl, fs = cases.synth_batch(B, seed=61, yaw=True)
    wbc = _make(B)
    wbc.enable_dump(True)
    tau, wr, st = _run(wbc, q, fl, fs)
    z, cp, cr = wbc.get("zmp"), wbc.get("contact_pos"), wbc.get("contact_rot")
    assert st.all()
    for b in range(B):
        c = Cycle(cases.tocabi_model())
        for cc in cases.CONTACTS_2:
            c.add_contact(cc["link"], cc["point"], cc["lx"], cc["ly"], cc["mu"], cc["muz"])
        c.add_task(0, 0, 0)
        c.add_task(1, 6, 15)
        c.set_torque_limit(cases.TAU_LIM)
        t = c.run(q[b], [1, 1], [fs[b, :6], fs[b, 6:]])
        zmp, zs = c.get_zmp(c.contact_force(t))
        assert np.abs(z[b, 0] - zmp).max() < 1e-6 and np.abs(z[b, 1] - zs[0]).max() < 1e-6 and np.abs(z[b, 2] - zs[1]).max() < 1e-6
        for a, link in enumerate((6, 12)):
            assert np.abs(cp[b, a] - (c.p[link] + c.R[link] @ np.array(cases.FOOT_POINT))).max() < 1e-12
            assert np.abs(cr[b, a] - c.R[link]).max() < 1e-12
    # the ZMP lies inside the support polygon spanned by the two feet (a size-independent sanity property)
    assert (np.abs(z[:, 0, :2] - 0.5 * (cp[:, 0, :2] + cp[:, 1, :2])) < 0.5).all()


def test_gpu_bench_two_ranks_gloo():
    """`python bench.py --gpus 2` on a one-GPU box with the gloo rehearsal switch: the launcher starts two ranks that share the
    card, each runs bench.rank_main with the HIP engine, rank 0 prints one line with n_gpus = 2."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["DWBC_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["collective_backend"] == "gloo" and line["config"]["status_ok_fraction"] > 0.9
    assert line["value"] > 1e4 and line["roofline"]["kernel"].startswith("dwbc::dwbc_cycle_kernel_v2")


@pytest.mark.gpu
def test_gpu_bench_one_rank_rccl_gather():
    """the RCCL branch of bench.rank_main (process group with backend nccl on the rank's device, packed rows on
    the device, all_gather_into_tensor, barrier, max-over-ranks all_reduce) executed with world = 1 on this box's one GPU, so that the
    driver's 1 -> 8 run is not the first execution of that code.  The rows that come back through RCCL equal the rank's own."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "DWBC_BENCH_BACKEND")}
    env["DWBC_BENCH_FORCE_COLLECTIVE"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "3", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["config"]["collective_backend"] == "rccl"
    assert line["config"]["forced_one_rank_gather_matches_local"] is True
    assert line["config"]["status_ok_fraction"] > 0.9 and line["value"] > 1e4


def test_redundant_task_levels_do_not_abort_the_cascade():
    """a redundant lower level (TASK_CUSTOM level whose Jacobian repeats rows of level 0) through the HIP
    kernel and the restatement: both finish with status 1 and the same torques (see tests/test_kernel_emulation.py for why the
    t x t blocks stay nonsingular; a rank-deficient task Jacobian itself makes Lambda_task undefined in the reference too)."""
    from tests.test_kernel_emulation import _redundant_custom_case

    import libdwbc_amd as D

    B = 8
    q, fl, fs, J, tau, st = _redundant_custom_case(B, 10)
    wbc = D.Batch(D.Model.from_urdf(cases.URDF), B, device=0)
    for c in cases.CONTACTS_2:
        wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    wbc.add_task(0, D.TASK_LINK_6D, 0)
    wbc.add_custom_task(1, 3)
    wbc.set_torque_limit(np.array(cases.TAU_LIM))
    wbc.set_state(q)
    wbc.set_contact(fl)
    wbc.set_fstar(0, fs[:, :6])
    wbc.set_custom_task(1, fs[:, 6:9], J)
    wbc.solve()
    assert (wbc.get("status") == st).all() and st.all()
    assert np.abs(wbc.get("tau") - tau).max() < TOL


def test_warm_start_sequence_on_device():
    """dwbc_batch_solve without DWBC_SOLVE_INIT (init = false): working sets carried over in HBM between launches.  A short
    sequence of correlated states: every warm solve equals the cold solve of the same state; active-set steps do not grow."""
    B = 256
    q, fl, fs = cases.synth_batch(B, seed=78, yaw=True)
    rng = np.random.default_rng(2)
    wbc, ref = _make(B), _make(B)
    steps_w = steps_c = 0
    for k in range(4):
        qk = q.copy()
        qk[:, 6:39] += 0.002 * k * rng.standard_normal((B, 33))
        fk = fs + 0.01 * k
        wbc.set_state(qk); wbc.set_contact(fl); wbc.set_fstar_all(fk)
        wbc.solve(init=(k == 0))
        tw, sw, dw = wbc.get("tau"), wbc.get("status"), wbc.get("diag")
        ref.set_state(qk); ref.set_contact(fl); ref.set_fstar_all(fk)
        ref.solve(init=True)
        tc, sc, dc = ref.get("tau"), ref.get("status"), ref.get("diag")
        assert (sw == sc).all() and sc.mean() > 0.9
        assert np.abs(tw - tc).max() < 1e-8
        if k > 0:
            steps_w += int(dw[:, 4:9].sum())
            steps_c += int(dc[:, 4:9].sum())
    assert 0 < steps_w <= steps_c
    # cold = a lean build (the two-wave kernel at this batch size, or the one-wave lean kernel), warm = the full build (carries the sets)
    assert ("v2p<" in ref.kernel_name() or "false" in ref.kernel_name()) and "true" in wbc.kernel_name()


def test_pipelined_set_state_does_not_tear_the_previous_upload():
    """the host mirrors are page-locked, so dwbc_batch_solve returns while the DMA engine is still reading
    them.  Rewriting a mirror right behind a solve -- through dwbc_batch_set_state / set_fstar or in place through
    dwbc_batch_host_ptr -- must wait for that upload: the first solve's results are those of ITS inputs, bit for bit."""
    B = 65536  # 21 MB of q: the upload is still in flight when the next call comes
    q, flags, fstar = cases.synth_batch(B, seed=20251226 + 9)
    ref = _make(B)
    tau_ref, wr_ref, st_ref = _run(ref, q, flags, fstar)
    junk_q = np.full_like(q, 7.0)
    junk_f = np.full_like(fstar, -3.0)
    for mode in ("set", "host_view"):
        wbc = _make(B)
        wbc.set_contact(flags)
        if mode == "set":
            wbc.set_state(q)
            wbc.set_fstar_all(fstar)
            wbc.solve()
            wbc.set_state(junk_q)       # no read-back, no sync in between
            wbc.set_fstar_all(junk_f)
        else:
            v = wbc.host_view("in_q")
            v[:] = q
            wbc.set_state(v)
            wbc.set_fstar_all(fstar)
            wbc.solve()
            v2 = wbc.host_view("in_q")  # the documented way to rewrite the mirror in place: waits for the pending upload
            v2[:] = junk_q
        tau, wr, st = wbc.get("tau"), wbc.get("wrench"), wbc.get("status")
        assert (st == st_ref).all(), mode
        assert np.array_equal(tau, tau_ref), mode
        assert np.array_equal(wr, wr_ref), mode


@pytest.mark.parametrize("cfg", ["ds", "ds_yaw", "mixed", "ss_L"])
def test_compact_throughput_kernel_vs_oracle(cfg, monkeypatch):
    """The lean register-capped kernel on the compact 20 KB LDS map (Lds3) is what every batch beyond four instances per CU
    runs -- configs[2..4] and the 8-GPU shards of 8192.  Forced here at a size the oracle finishes in seconds (DWBC_NO_WIDE: the
    launcher skips the one-wave-per-SIMD build), every instance against the oracle."""
    monkeypatch.setenv("DWBC_NO_WIDE", "1")
    B = 1536
    tasks, kw = cases.TASKS_2LEVEL, dict(seed=20251226 + 11)
    if cfg == "ds_yaw":
        kw["yaw"] = True
    elif cfg == "mixed":
        kw["contact_mode"] = "mixed"
    elif cfg == "ss_L":
        kw.update(contact_mode="L", levels=3)
        tasks = cases.TASKS_3LEVEL_SWING_R
    q, flags, fstar = cases.synth_batch(B, **kw)
    wbc = _make(B, tasks=tasks)
    tau, wr, st = _run(wbc, q, flags, fstar)
    name = wbc.kernel_name()
    assert "dwbc_cycle_kernel_v2<" in name and name.endswith("TopoTocabi, true>"), name
    nt, lds = wbc.launch_info()
    assert lds <= (20480 if len(tasks) == 2 else 26624), lds
    tau_r, wr_r, st_r, _ = _oracle(B, q, flags, fstar, tasks=tasks)
    assert (st == st_r).all()
    ok = st_r == 1
    assert ok.mean() > 0.9
    assert np.abs(tau[ok] - tau_r[ok]).max() < TOL
    assert np.abs(wr[ok] - wr_r[ok]).max() < 1e-5


def test_near_singular_knee_truncated_pseudo_inverse_on_device():
    """a7 / VERDICT r2 next 4(c): the pelvis level's Q W^+ Q^T with the left knee 1e-2 .. 1e-6 rad from straight -- below 3e-3 rad
    the reference's complete orthogonal decomposition (threshold 1e-6, src/wbd.cpp:5-30,212) truncates the block to rank 5 and the
    kernel follows it.  Envelope and reasons: tests/test_kernel_emulation.py (same states, same thresholds); both the wide and the
    compact kernel."""
    import os

    from tests.test_kernel_emulation import KNEES, _straight_knee_states

    q, fl, fs = _straight_knee_states(KNEES)
    B = len(KNEES)
    tau_r, wr_r, st_r, _ = _oracle(B, q, fl, fs)
    for no_wide in ("", "1"):
        if no_wide:
            os.environ["DWBC_NO_WIDE"] = "1"
        try:
            wbc = _make(B)
            tau, wr, st = _run(wbc, q, fl, fs)
        finally:
            os.environ.pop("DWBC_NO_WIDE", None)
        assert (st == st_r).all() and st_r.all()
        err = np.abs(tau - tau_r).max(axis=(1, 2))
        assert err[:3].max() < TOL, err
        assert err[3:].max() < 5e-5, err


@pytest.mark.parametrize("cfg", ["ds", "ds_yaw", "mixed", "one_level", "free"])
def test_paired_two_wave_kernel_vs_oracle(cfg):
    """Batches of at most one instance per SIMD run the lean cycle with TWO wavefronts per instance (dwbc_cycle2p.h: the side chains
    -- contact Jacobians, internal-wrench algebra, task Jacobians, Lambda_task -- on a helper wave beside the main chain, five
    workgroup barriers).  Same arithmetic per block as the one-wave kernel; every instance against the oracle, both role
    assignments (DWBC_PAIR_SWAP_BIT) and the one-wave kernel (DWBC_NO_PAIR) side by side."""
    import os

    B = 1024
    tasks, kw = cases.TASKS_2LEVEL, dict(seed=20251226 + 12)
    if cfg == "ds_yaw":
        kw["yaw"] = True
    elif cfg in ("mixed", "free"):
        kw["contact_mode"] = "mixed"
    q, flags, fstar = cases.synth_batch(B, **kw)
    if cfg == "free":
        flags[::3] = 0  # a third of the batch without any active contact
    if cfg == "one_level":
        tasks, fstar = [cases.TASKS_2LEVEL[0]], fstar[:, :6].copy()
    tau_r, wr_r, st_r, _ = _oracle(B, q, flags, fstar, tasks=tasks)
    ok = st_r == 1
    assert ok.mean() > 0.9
    for env in ({}, {"DWBC_PAIR_SWAP_BIT": "-1"}, {"DWBC_PAIR_SWAP_BIT": "0"}, {"DWBC_NO_PAIR": "1"}):
        os.environ.update(env)
        try:
            wbc = _make(B, tasks=tasks)
            tau, wr, st = _run(wbc, q, flags, fstar)
            name = wbc.kernel_name()
            nt, lds = wbc.launch_info()
        finally:
            for k_ in env:
                os.environ.pop(k_, None)
        if "DWBC_NO_PAIR" in env:
            assert "dwbc_cycle_kernel_v2w<" in name and nt == 64, name
        else:
            assert "dwbc_cycle_kernel_v2p<" in name and nt == 128 and lds <= 40960, (name, nt, lds)
        assert (st == st_r).all(), env
        assert np.abs(tau[ok] - tau_r[ok]).max() < TOL, env
        assert np.abs(wr[ok] - wr_r[ok]).max() < 1e-5, env

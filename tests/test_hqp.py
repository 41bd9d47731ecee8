"""the generic hierarchical-QP class (reference include/dwbc_hqp.h, src/dwbc_hqp.cpp) and the LQP
configurator (RobotData::ConfigureLQP / CalcControlTorqueLQP, src/dwbc.cpp:4304-4452).

PARITY UNPINNED in the reference: each level goes to OSQP (not vendored) and no test asserts a number of this path
(tests/sp_test/herzog_test.cpp, jacc_compare.cpp only print).  What is pinned here:
  * the numpy restatement (oracle/hqp_np.py) by first principles: every level's answer satisfies the KKT conditions of the
    problem the reference poses, and the hierarchy holds (a later level never changes an earlier level's equality residual or
    slack);
  * the kernels (host emulation, then HIP through the C-ABI) against that restatement to 1e-6;
  * on the golden CASE 1 / 2 states, the matrices the configurator builds from golden-pinned A_, A_inv_, J_C.
"""
import numpy as np
import pytest

from oracle import dwbc_np as Dn
from oracle import hqp_np as H
from tests import cases

TOL = 1e-6


def _random_hierarchy(rng, nv=14, levels=((6, 3), (5, 2), (0, 3), (4, 2)), cost_levels=(1, 3)):
    """a small generic hierarchy with inequalities that bite: returns per-level dicts"""
    out = []
    for i, (m, e) in enumerate(levels):
        d = dict(m=m, e=e)
        d["A"] = rng.standard_normal((m, nv))
        d["a"] = -0.3 * np.abs(rng.standard_normal(m)) + 0.4 * rng.standard_normal(m)
        d["B"] = rng.standard_normal((e, nv))
        d["b"] = rng.standard_normal(e)
        if i in cost_levels:
            Q = rng.standard_normal((nv, nv))
            d["H"] = 0.05 * (Q @ Q.T)
        out.append(d)
    return out


def _oracle_generic(levels, nv, solve_first=True):
    hq = H.HQP()
    hq.initialize(nv, 0, 0)
    for d in levels:
        hq.addHierarchy(d["m"], d["e"])
        h = hq.hqp_hs_[-1]
        h.updateConstraintMatrix(d["A"] if d["m"] else None, d["a"] if d["m"] else None, d["B"], d["b"])
        if "H" in d:
            h.updateCostMatrix(d["H"], np.zeros(nv))
        h.normalizeConstraintMatrix()
    hq.prepare()
    ok = hq.solvefirst() if solve_first else 1
    ok &= hq.solveSequential()
    return hq, ok


def _check_level_kkt(hq, i, tol=1e-6):
    """KKT of level i's problem (module docstring of oracle/hqp_np.py) at its answer"""
    hs = hq.hqp_hs_
    h = hs[i]
    n = h.variable_size_
    y_prev = hs[i - 1].y_ans_ if i > 0 else np.zeros(n)
    Z = hs[i - 1].Z_ if i > 0 else np.eye(n)
    u = np.linalg.lstsq(Z, h.y_ans_ - y_prev, rcond=None)[0]
    assert np.abs(Z @ u - (h.y_ans_ - y_prev)).max() < 1e-8  # the step stays in the null space of the earlier equalities
    Bz = h.B_ @ Z
    grad = Bz.T @ (Bz @ u + h.B_ @ y_prev + h.b_) + H.HQP_EPS * u
    if h.enable_cost_:
        grad = grad + Z.T @ h.H_ @ (y_prev + Z @ u)
    lam_rows, lam_vals = [], []
    if h.ineq_const_size_ > 0:
        s = h.A_ @ h.y_ans_ + h.a_  # own rows: v = max(0, s), multiplier = v
        assert np.abs(h.v_ans_ - np.maximum(s, 0.0)).max() < tol
        grad = grad + (h.A_ @ Z).T @ h.v_ans_
    for j in range(i):
        hj = hs[j]
        if hj.ineq_const_size_ == 0:
            continue
        sj = hj.A_ @ h.y_ans_ + hj.a_ - hj.v_ans_
        assert sj.max() < tol  # earlier levels' rows hold with their frozen slack
        act = sj > -1e-7
        lam_rows.append((hj.A_ @ Z)[act])
    if lam_rows and sum(r.shape[0] for r in lam_rows):
        C = np.vstack(lam_rows)
        lam = np.linalg.lstsq(C.T, -grad, rcond=None)[0]
        assert np.abs(C.T @ lam + grad).max() < tol * (1 + np.abs(grad).max())
        assert lam.min() > -1e-6 * (1 + np.abs(lam).max())
    else:
        assert np.abs(grad).max() < tol


def test_oracle_generic_hierarchy_satisfies_kkt_and_priority():
    rng = np.random.default_rng(5)
    for trial in range(6):
        levels = _random_hierarchy(rng)
        hq, ok = _oracle_generic(levels, 14)
        assert ok == 1
        assert [h.null_space_size_ for h in hq.hqp_hs_] == [11, 9, 6, 4]
        for i in range(len(levels)):
            _check_level_kkt(hq, i)
        # priority: later answers keep the earlier levels' equality residuals (they move inside the null spaces)
        for i in range(len(levels)):
            for j in range(i):
                hj = hq.hqp_hs_[j]
                assert np.abs((hj.B_ @ hq.hqp_hs_[i].y_ans_ + hj.b_) - (hj.B_ @ hj.y_ans_ + hj.b_)).max() < 1e-8
        assert any(len(h.working_set_) > 0 for h in hq.hqp_hs_)


def _lqp_oracle(q, fstar, tasks=cases.TASKS_2LEVEL):
    m = cases.tocabi_model()
    c = Dn.Cycle(m)
    for cc in cases.CONTACTS_2:
        c.add_contact(cc["link"], cc["point"], cc["lx"], cc["ly"], cc["mu"], cc["muz"])
    for lv, links in enumerate(tasks):
        for mode, link, pt in links:
            c.add_task(lv, mode, link, pt)
    c.update_kinematics(q)
    c.set_contact([1, 1])
    Bn = Dn.nonlinear_effects(m, q, np.zeros(39))
    Js = [c.task_jacobian(i) for i in range(len(tasks))]
    fs, off = [], 0
    for J in Js:
        fs.append(fstar[off : off + J.shape[0]])
        off += J.shape[0]
    hq = H.configure_lqp(c, Bn, Js, fs)
    ok = hq.solveSequential()
    return hq, ok, H.lqp_torque(c, Bn, hq.hqp_hs_[-1].y_ans_), c


@pytest.mark.parametrize("case", [1, 2])
def test_oracle_lqp_on_golden_states(case):
    """ConfigureLQP on the reference's CASE 1 / 2 states: the level matrices are built from golden-pinned quantities, the cascade
    satisfies KKT level by level, and the resulting torque respects the rows it was asked to respect."""
    q = np.array(cases.Q_CASE[case])
    fs = np.array(list(cases.FSTAR_CASE[case][0]) + list(cases.FSTAR_CASE[case][1]))
    hq, ok, tau, c = _lqp_oracle(q, fs)
    assert ok == 1
    assert np.abs(c.A_inv - cases.golden(case, "A_inv_")).max() < 1e-8 and np.abs(c.J_C - cases.golden(case, "J_C")).max() < 1e-10
    hs = hq.hqp_hs_
    assert [h.null_space_size_ for h in hs] == [45, 33, 27, 24]
    assert [(h.ineq_const_size_, h.eq_const_size_) for h in hs] == [(66, 6), (86, 12), (0, 6), (0, 3)]
    for i in range(1, 4):
        _check_level_kkt(hq, i, tol=2e-6)
    y = hs[-1].y_ans_
    # floating-base dynamics hold exactly through the whole cascade (level 0 equality), the torque stays inside +-200
    assert np.abs(hs[0].B_ @ y + hs[0].b_).max() < 1e-9
    assert np.abs(tau).max() < H.LQP_TAU_LIM
    # unilateral, loaded feet (f_z < 0 in the reference's sign convention) and accelerations inside +-5
    assert y[39 + 2] < 0 and y[39 + 8] < 0 and np.abs(y[6:39]).max() <= H.LQP_ACC_LIM + 1e-6
    # B_(q, 0) = G_ (what the kernel uses when no qdot was supplied)
    assert np.abs(Dn.nonlinear_effects(cases.tocabi_model(), q, np.zeros(39)) - c.G).max() < 1e-9


def test_emulated_hqp_generic_hierarchy_matches_oracle():
    from tests.emu.emu import EmuHQP

    rng = np.random.default_rng(11)
    B, nv = 5, 14
    probs = [_random_hierarchy(rng) for _ in range(B)]
    m = [d["m"] for d in probs[0]]
    e = [d["e"] for d in probs[0]]
    hc = [1 if "H" in d else 0 for d in probs[0]]
    eh = EmuHQP(B, nv, m, e, hc, share_cost=False, solve_first=True)
    refs = []
    for b, levels in enumerate(probs):
        hq, ok = _oracle_generic(levels, nv)
        assert ok == 1
        refs.append(hq)
        for lv, h in enumerate(hq.hqp_hs_):  # the emulation takes the normalised matrices the oracle built
            if m[lv]:
                eh.block(lv, 0, (m[lv], nv))[b] = h.A_
                eh.block(lv, 1, (m[lv],))[b] = h.a_
            eh.block(lv, 2, (e[lv], nv))[b] = h.B_
            eh.block(lv, 3, (e[lv],))[b] = h.b_
            if hc[lv]:
                eh.block(lv, 4, (nv, nv))[b] = h.H_
    eh.solve()
    for b, hq in enumerate(refs):
        for lv, h in enumerate(hq.hqp_hs_):
            assert eh.status(lv)[b] == 1 and eh.null_size(lv)[b] == h.null_space_size_
            assert np.abs(eh.block(lv, 5, (nv,))[b] - h.y_ans_).max() < TOL, (b, lv)
            if m[lv]:
                assert np.abs(eh.block(lv, 6, (m[lv],))[b] - h.v_ans_).max() < TOL
            assert np.abs(eh.block(lv, 7, (e[lv],))[b] - h.w_ans_).max() < TOL
    assert sum(int(eh.iters(lv).sum()) for lv in range(4)) > 0  # inequalities were active somewhere


def test_emulated_lqp_matches_oracle_on_golden_and_synthetic_states():
    from tests.emu.emu import Emu, EmuHQP

    B = 6
    q, fl, fs = cases.synth_batch(B, seed=23, yaw=True)
    q[0], q[1] = cases.Q_CASE[1], cases.Q_CASE[2]
    for i, case in enumerate((1, 2)):
        fs[i] = list(cases.FSTAR_CASE[case][0]) + list(cases.FSTAR_CASE[case][1])
    fs[2:] *= 3.0  # larger task accelerations: more rows of the limits become active
    e = Emu(cases.URDF, cases.CONTACTS_2, cases.TASKS_2LEVEL, cases.TAU_LIM)
    r = e.run(q, fl, fs, dump=True)
    eh = EmuHQP(B, 51, [66, 86, 0, 0], [6, 12, 6, 3], [0, 1, 1, 1], share_cost=True)
    eh.configure_lqp(e, [0, 1], r["dump"], fs)
    eh.solve()
    tau = eh.lqp_torque(e, 2, r["dump"])
    for b in range(B):
        hq, ok, tau_ref, _ = _lqp_oracle(q[b], fs[b])
        assert ok == 1
        for lv, h in enumerate(hq.hqp_hs_):
            assert eh.status(lv)[b] == 1 and eh.null_size(lv)[b] == h.null_space_size_
            if h.ineq_const_size_:
                assert np.abs(eh.block(lv, 0, (h.ineq_const_size_, 51))[b] - h.A_).max() < 1e-9
                assert np.abs(eh.block(lv, 1, (h.ineq_const_size_,))[b] - h.a_).max() < 1e-9
            assert np.abs(eh.block(lv, 2, (h.eq_const_size_, 51))[b] - h.B_).max() < 1e-9
            assert (np.abs(eh.block(lv, 5, (51,))[b] - h.y_ans_) / (1.0 + np.abs(h.y_ans_))).max() < TOL, (b, lv)
        assert np.abs(tau[b] - tau_ref).max() < 1e-5  # |A| ~ 1e2 amplifies the 1e-7 of y


# ------------------------------------------------------------------------------------------------------------------ GPU
def test_oracle_row_weights_enter_solvefirst_only():
    """HQP_Hierarch::updateInequalityCostWeight / updateEqualityCostWeight (reference src/dwbc_hqp.cpp:503-553): solvefirst poses level 0
    on (V A, V a, W B, W b) (:245-254); no later solve reads the weights.  The restatement: weighted solvefirst == unweighted solvefirst
    of the pre-scaled problem, and solveSequential is untouched by the weights."""
    rng = np.random.default_rng(5)
    nv = 14
    lv = _random_hierarchy(rng)
    m, e = lv[0]["m"], lv[0]["e"]
    V, W = np.diag(0.5 + rng.random(m)) + 0.1 * rng.standard_normal((m, m)), np.diag(0.5 + rng.random(e))

    def build(pre_scaled, weights):
        hq = H.HQP()
        hq.initialize(nv, 0, 0)
        for i, d in enumerate(lv):
            hq.addHierarchy(d["m"], d["e"])
            h = hq.hqp_hs_[-1]
            A, a, Bm, b = d["A"], d["a"], d["B"], d["b"]
            if pre_scaled and i == 0:
                A, a, Bm, b = V @ A, V @ a, W @ Bm, W @ b
            h.updateConstraintMatrix(A if d["m"] else None, a if d["m"] else None, Bm, b)
            if "H" in d:
                h.updateCostMatrix(d["H"], np.zeros(nv))
        if weights:
            hq.hqp_hs_[0].updateConstraintWeight(V, W)
        if weights == 2:
            hq.hqp_hs_[2].updateEqualityCostWeight(np.full(lv[2]["e"], 7.0))  # a later level's weight: stored, never read
        hq.prepare()
        return hq

    hw, hw2, hs_, hu = build(False, 1), build(False, 2), build(True, 0), build(False, 0)
    assert hw.solvefirst() == 1 and hw2.solvefirst() == 1 and hs_.solvefirst() == 1 and hu.solvefirst() == 1
    assert np.abs(hw.hqp_hs_[0].y_ans_ - hs_.hqp_hs_[0].y_ans_).max() < 1e-12
    assert np.abs(hw.hqp_hs_[0].y_ans_ - hu.hqp_hs_[0].y_ans_).max() > 1e-4  # the weights do change level 0's answer
    # solveSequential starts at level 1 from level 0's (weighted) answer and reads level 0's UNWEIGHTED rows (dwbc_hqp.cpp:291-403);
    # a weight set on a later level changes nothing
    assert hw.solveSequential() == 1 and hw2.solveSequential() == 1
    for a_, b_ in zip(hw.hqp_hs_, hw2.hqp_hs_):
        assert np.abs(a_.y_ans_ - b_.y_ans_).max() < 1e-14
    assert np.abs(hw.hqp_hs_[0].A_ - lv[0]["A"]).max() == 0.0  # the level keeps its unweighted matrices


@pytest.mark.gpu
def test_gpu_hqp_row_weights_match_oracle():
    """dwbc_hqp_update_constraint_weight: solvefirst with (V, W) on the device equals the restatement's; the solveSequential that
    follows reads the unweighted matrices again (the device blocks are put back after the weighted launch)."""
    import libdwbc_amd as D
    from libdwbc_amd import hqp as Hq

    rng = np.random.default_rng(21)
    B, nv = 16, 14
    probs = [_random_hierarchy(rng) for _ in range(B)]
    sizes = [(d["m"], d["e"]) for d in probs[0]]
    m0, e0 = sizes[0]
    Vs = np.array([np.diag(0.5 + rng.random(m0)) + 0.1 * rng.standard_normal((m0, m0)) for _ in range(B)])
    Ws = np.array([0.5 + rng.random(e0) for _ in range(B)])  # vectors: diagonals
    hq = D.HQP(B, nv, 0, 0)
    for m, e in sizes:
        hq.addHierarchy(m, e)
    for lv, (m, e) in enumerate(sizes):
        hq.updateConstraintMatrix(lv, np.array([p[lv]["A"] for p in probs]) if m else None, np.array([p[lv]["a"] for p in probs]) if m else None,
                                  np.array([p[lv]["B"] for p in probs]), np.array([p[lv]["b"] for p in probs]))
        if "H" in probs[0][lv]:
            hq.updateCostMatrix(lv, np.array([p[lv]["H"] for p in probs]))
    hq.updateConstraintWeight(0, Vs, Ws)
    hq.prepare()
    hq.solvefirst()
    y_first = hq.y_ans(0).copy()
    hq.solveSequential()
    for b in range(B):
        ref = H.HQP()
        ref.initialize(nv, 0, 0)
        for d in probs[b]:
            ref.addHierarchy(d["m"], d["e"])
            h = ref.hqp_hs_[-1]
            h.updateConstraintMatrix(d["A"] if d["m"] else None, d["a"] if d["m"] else None, d["B"], d["b"])
            if "H" in d:
                h.updateCostMatrix(d["H"], np.zeros(nv))
        ref.hqp_hs_[0].updateConstraintWeight(Vs[b], np.diag(Ws[b]))
        ref.prepare()
        assert ref.solvefirst() == 1
        assert np.abs(y_first[b] - ref.hqp_hs_[0].y_ans_).max() < TOL, b
        assert ref.solveSequential() == 1
        for lv, h in enumerate(ref.hqp_hs_):
            assert hq.get(lv, Hq.STATUS)[b] == 1
            assert np.abs(hq.y_ans(lv)[b] - h.y_ans_).max() < TOL, (b, lv)


@pytest.mark.gpu
def test_gpu_hqp_class_generic_hierarchy_matches_oracle():
    import libdwbc_amd as D

    rng = np.random.default_rng(12)
    B, nv = 48, 14
    probs = [_random_hierarchy(rng) for _ in range(B)]
    lv_sizes = [(d["m"], d["e"]) for d in probs[0]]
    hq = D.HQP(B, nv, 0, 0)
    for m, e in lv_sizes:
        hq.addHierarchy(m, e)
    for lv, (m, e) in enumerate(lv_sizes):
        hq.updateConstraintMatrix(lv, np.array([p[lv]["A"] for p in probs]) if m else None, np.array([p[lv]["a"] for p in probs]) if m else None,
                                  np.array([p[lv]["B"] for p in probs]), np.array([p[lv]["b"] for p in probs]))
        if "H" in probs[0][lv]:
            hq.updateCostMatrix(lv, np.array([p[lv]["H"] for p in probs]))
        hq.normalizeConstraintMatrix(lv)
    hq.prepare()
    hq.solvefirst()
    hq.solveSequential()
    from libdwbc_amd import hqp as Hq

    nact = 0
    for b in range(B):
        ref, ok = _oracle_generic(probs[b], nv)
        assert ok == 1
        for lv, h in enumerate(ref.hqp_hs_):
            assert hq.get(lv, Hq.STATUS)[b] == 1 and hq.get(lv, Hq.NULL_SIZE)[b] == h.null_space_size_
            assert np.abs(hq.y_ans(lv)[b] - h.y_ans_).max() < TOL, (b, lv)
            assert np.abs(hq.w_ans(lv)[b] - h.w_ans_).max() < TOL
            if h.ineq_const_size_:
                assert np.abs(hq.v_ans(lv)[b] - h.v_ans_).max() < TOL
            nact += len(h.working_set_)
    assert nact > B  # the inequalities bite


@pytest.mark.gpu
def test_gpu_lqp_batch_matches_oracle_and_keeps_its_constraints():
    """RobotData::ConfigureLQP + CalcControlTorqueLQP at B = 1024 (BASELINE configs[1] shape) on the device: a seeded subset
    against the restatement, the whole batch through the properties of the formulation."""
    import libdwbc_amd as D
    from libdwbc_amd import hqp as Hq

    B, NS = 1024, 12
    q, fl, fs = cases.synth_batch(B, seed=31)
    q[0], q[1] = cases.Q_CASE[1], cases.Q_CASE[2]
    for i, case in enumerate((1, 2)):
        fs[i] = list(cases.FSTAR_CASE[case][0]) + list(cases.FSTAR_CASE[case][1])
    fs[2:] *= 3.0
    wbc = D.Batch(D.Model.from_urdf(cases.URDF), B, device=0)
    for c in cases.CONTACTS_2:
        wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    wbc.add_task(0, D.TASK_LINK_6D, 0)
    wbc.add_task(1, D.TASK_LINK_ROTATION, 15)
    wbc.set_torque_limit(np.array(cases.TAU_LIM))
    wbc.enable_dump(True)
    wbc.set_state(q)
    wbc.set_contact(fl)
    wbc.set_fstar_all(fs)
    wbc.solve()
    hq = D.HQP.for_lqp(wbc, 12)
    hq.configure_lqp(wbc)
    hq.solveSequential()
    tau = hq.lqp_torque(wbc)
    assert hq.num_levels() == 4
    for lv in range(4):
        assert (hq.get(lv, Hq.STATUS) == 1).all()
    assert (hq.get(0, Hq.NULL_SIZE) == 45).all() and (hq.get(3, Hq.NULL_SIZE) == 24).all()
    y = hq.y_ans(3)
    for b in range(NS):
        ref, ok, tau_ref, _ = _lqp_oracle(q[b], fs[b])
        assert ok == 1
        for lv, h in enumerate(ref.hqp_hs_):
            assert (np.abs(hq.y_ans(lv)[b] - h.y_ans_) / (1.0 + np.abs(h.y_ans_))).max() < TOL, (b, lv)
        assert np.abs(tau[b] - tau_ref).max() < 1e-5
    # whole batch: floating-base dynamics (level-0 equality) hold for the final answer, torque inside the LQP's own limit,
    # level-1 rows hold with their slack, contact forces unilateral
    B0, b0 = hq.get(0, Hq.MAT_B).reshape(B, 6, 51), hq.get(0, Hq.VEC_b)
    assert np.abs(np.einsum("bij,bj->bi", B0, y) + b0).max() < 1e-8
    assert np.abs(tau).max() < 200.0 + 1e-6
    A1, a1, v1 = hq.get(1, Hq.MAT_A).reshape(B, 86, 51), hq.get(1, Hq.VEC_a), hq.v_ans(1)
    assert (np.einsum("bij,bj->bi", A1, y) + a1 - v1).max() < 1e-7
    assert (y[:, 39 + 2] < 1e-6).all() and (y[:, 39 + 8] < 1e-6).all()  # unilateral contact (f_z <= 0 in the reference's convention)
    # a second configure + solve on a new state reuses the object (CalcControlTorqueLQP per cycle)
    q2 = q.copy()
    q2[:, 6:39] += 0.01
    wbc.set_state(q2)
    wbc.solve()
    hq.configure_lqp(wbc)
    hq.solveSequential(init=False)
    assert np.abs(hq.y_ans(3) - y).max() > 1e-6 and (hq.get(3, Hq.STATUS) == 1).all()


# ------------------------------------------------------------------------------------------------------------------ JACC
def _jacc_oracle(q, fstar, tasks=cases.TASKS_2LEVEL):
    """CalcSingleTaskTorqueWithJACC_QP level after level (reference tests/sp_test/dof_comparison_jacc.cpp:296-300)"""
    m = cases.tocabi_model()
    c = Dn.Cycle(m)
    for cc in cases.CONTACTS_2:
        c.add_contact(cc["link"], cc["point"], cc["lx"], cc["ly"], cc["mu"], cc["muz"])
    for lv, links in enumerate(tasks):
        for mode, link, pt in links:
            c.add_task(lv, mode, link, pt)
    c.update_kinematics(q)
    c.set_contact([1, 1])
    Js = [c.task_jacobian(i) for i in range(len(tasks))]
    fs, off = [], 0
    for J in Js:
        fs.append(np.asarray(fstar[off : off + J.shape[0]], float))
        off += J.shape[0]
    res, fqp = [], []
    for lv in range(len(tasks)):
        ok, acc, tau, f, s, hq = H.jacc_qp(c, lv, Js, fs, fqp)
        fqp.append(s)
        res.append(dict(ok=ok, acc=acc, tau=tau, f=f, s=s, hq=hq))
    return c, Js, fs, res


def _check_jacc_kkt(c, Js, fs, res, lv, tol=1e-6):
    """KKT of the reference's ORIGINAL QP over x = [qddot; tau; f_c; s] (src/dwbc.cpp:3806-3905) at the answer -- the elimination of
    tau and s used by the solver is checked by going back to the form the reference writes down."""
    n, m, cd = c.n, c.m, c.cdof
    r = res[lv]
    J, t = Js[lv], Js[lv].shape[0]
    x = np.concatenate([r["acc"], r["tau"], r["f"], r["s"]])
    nx = x.size
    Hq = np.zeros((nx, nx))
    Hq[:n, :n] = c.A
    Hq[n + m + cd :, n + m + cd :] = 100.0 * np.eye(t)
    ST = np.zeros((n, m))
    ST[6:] = np.eye(m)
    rows, rhs = [np.hstack([c.A, -ST, c.J_C.T, np.zeros((n, t))])], [-c.G]
    rows.append(np.hstack([c.J_C, np.zeros((cd, m + cd + t))])); rhs.append(np.zeros(cd))
    for i in range(lv):
        ti = Js[i].shape[0]
        rows.append(np.hstack([Js[i], np.zeros((ti, m + cd + t))])); rhs.append(fs[i] + res[i]["s"])
    rows.append(np.hstack([J, np.zeros((t, m + cd)), -np.eye(t)])); rhs.append(fs[lv])
    Aeq, beq = np.vstack(rows), np.concatenate(rhs)
    assert np.abs(Aeq @ x - beq).max() < 1e-8
    # inequalities G x <= h: cones, joint acceleration bounds, torque bounds
    Cm = -c.cone_matrix()
    G = [np.hstack([np.zeros((Cm.shape[0], n + m)), Cm, np.zeros((Cm.shape[0], t))])]
    hh = [np.zeros(Cm.shape[0])]
    Ia = np.zeros((m, nx)); Ia[:, 6:n] = np.eye(m)
    It = np.zeros((m, nx)); It[:, n : n + m] = np.eye(m)
    for Mx, lim in ((Ia, H.JACC_ACC_LIM), (It, H.JACC_TAU_LIM)):
        G += [Mx, -Mx]; hh += [np.full(m, lim), np.full(m, lim)]
    G, hh = np.vstack(G), np.concatenate(hh)
    sl = hh - G @ x
    assert sl.min() > -1e-6
    act = sl < 1e-6
    # stationarity: H x + Aeq^T nu + G_act^T lam = 0 for SOME lam >= 0 (the active rows are degenerate: bounds, cones and the
    # equalities are linearly dependent at these points, so existence is checked by non-negative least squares; the Tikhonov term of
    # the canon enters at 1e-6 |y|)
    from scipy.optimize import nnls

    Mk = np.hstack([Aeq.T, -Aeq.T, G[act].T])
    sol, rn = nnls(Mk, -Hq @ x, maxiter=20000)
    assert rn < 1e-3 * (1 + np.abs(Hq @ x).max()), rn


@pytest.mark.parametrize("case", [1, 2])
def test_oracle_jacc_qp_satisfies_the_reference_qp(case):
    q = np.array(cases.Q_CASE[case])
    fs = np.array(list(cases.FSTAR_CASE[case][0]) + list(cases.FSTAR_CASE[case][1]))
    c, Js, fsl, res = _jacc_oracle(q, fs)
    for lv in range(2):
        assert res[lv]["ok"] == 1
        _check_jacc_kkt(c, Js, fsl, res, lv)
    # the second level keeps the first level's task at f* + f*_qp (hard equality)
    assert np.abs(Js[0] @ res[1]["acc"] - fsl[0] - res[0]["s"]).max() < 1e-9


def test_emulated_jacc_matches_oracle():
    from tests.emu.emu import Emu, EmuHQP

    B = 4
    q, fl, fs = cases.synth_batch(B, seed=29, yaw=True)
    q[0], q[1] = cases.Q_CASE[1], cases.Q_CASE[2]
    for i, case in enumerate((1, 2)):
        fs[i] = list(cases.FSTAR_CASE[case][0]) + list(cases.FSTAR_CASE[case][1])
    e = Emu(cases.URDF, cases.CONTACTS_2, cases.TASKS_2LEVEL, cases.TAU_LIM)
    r = e.run(q, fl, fs, dump=True)
    prev = []
    outs = []
    for lv, (e0, t) in enumerate(((18, 6), (24, 3))):
        eh = EmuHQP(B, 51, [152, 0], [e0, t], [0, 1])
        eh.set_exact(0)
        out, st = eh.jacc_solve(e, [0, 1], lv, r["dump"], fs, prev)
        assert st.all()
        prev.append(out)
        outs.append(out)
    for b in range(B):
        c, Js, fsl, res = _jacc_oracle(q[b], fs[b])
        for lv in range(2):
            o = outs[lv][b]
            rel = lambda a, ref: (np.abs(a - ref) / (1 + np.abs(ref))).max()
            # rows within the 1e-6 row tolerance of their bound may or may not enter the working set: answers agree to that scale
            # ... and the internal-wrench part of f_c has no cost in this formulation: it is set by the Tikhonov weight alone and
            # moves by 1e-5 relative with such a choice, tau = A qddot + J_C^T f_c + G with it
            assert rel(o[:39], res[lv]["acc"]) < 1e-5 and rel(o[39:72], res[lv]["tau"]) < 1e-3
            assert rel(o[72:84], res[lv]["f"]) < 1e-3
            t = Js[lv].shape[0]
            assert rel(o[84 : 84 + t], res[lv]["s"]) < 1e-5


@pytest.mark.gpu
def test_gpu_jacc_matches_oracle_and_keeps_its_constraints():
    """CalcSingleTaskTorqueWithJACC_QP for both task levels of a batch on the device, against the restatement on a subset and
    through the constraints of the formulation on the whole batch."""
    import libdwbc_amd as D

    B, NS = 256, 6
    q, fl, fs = cases.synth_batch(B, seed=33)
    q[0], q[1] = cases.Q_CASE[1], cases.Q_CASE[2]
    for i, case in enumerate((1, 2)):
        fs[i] = list(cases.FSTAR_CASE[case][0]) + list(cases.FSTAR_CASE[case][1])
    fs[NS:] *= 2.0  # beyond the compared subset: many rows active (vertex solutions), checked through the constraints only
    wbc = D.Batch(D.Model.from_urdf(cases.URDF), B, device=0)
    for c in cases.CONTACTS_2:
        wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    wbc.add_task(0, D.TASK_LINK_6D, 0)
    wbc.add_task(1, D.TASK_LINK_ROTATION, 15)
    wbc.set_torque_limit(np.array(cases.TAU_LIM))
    wbc.enable_dump(True)
    wbc.set_state(q); wbc.set_contact(fl); wbc.set_fstar_all(fs)
    wbc.solve()
    hq = D.HQP.for_lqp(wbc, 12)
    with pytest.raises(D.DwbcError, match="in order"):
        hq.solve_jacc(wbc, 1)
    out = []
    for lv in range(2):
        hq.solve_jacc(wbc, lv)
        out.append(D.HQP.jacc_result(wbc, lv))
        assert out[lv]["status"].all()
    rel = lambda a, ref: (np.abs(a - ref) / (1 + np.abs(ref))).max()
    for b in range(NS):
        c, Js, fsl, res = _jacc_oracle(q[b], fs[b])
        for lv in range(2):
            assert rel(out[lv]["acc_qp"][b], res[lv]["acc"]) < 1e-5
            assert rel(out[lv]["torque_qp"][b], res[lv]["tau"]) < 1e-3  # (see test_emulated_jacc_matches_oracle)
            assert rel(out[lv]["contact_qp"][b], res[lv]["f"]) < 1e-3
            t = Js[lv].shape[0]
            assert rel(out[lv]["f_star_qp"][b][:t], res[lv]["s"]) < 1e-5
    A, JC, G = wbc.get("A"), wbc.get("J_C"), wbc.get("G")
    for lv in range(2):
        acc, tau, f = out[lv]["acc_qp"], out[lv]["torque_qp"], out[lv]["contact_qp"]
        dyn = np.einsum("bij,bj->bi", A, acc) + np.einsum("bji,bj->bi", JC, f) + G
        dyn[:, 6:] -= tau
        assert np.abs(dyn).max() < 1e-7                                   # rigid-body dynamics
        assert np.abs(np.einsum("bij,bj->bi", JC, acc)).max() < 1e-8      # contact constraint
        assert np.abs(acc[:, 6:]).max() < 10 + 1e-6 and np.abs(tau).max() < 200 + 1e-6
        assert (f[:, 2] < 1e-4).all() and (f[:, 8] < 1e-4).all()  # unilateral (implied by the cone rows, each held to 1e-6 on the normalised row)


# ------------------------------------------------------------------------------------------- capacity of the device solver
def _capacity_problem(nrow, nv=40):
    """one level whose optimum keeps EVERY inequality row active: B x = 0 asks the eight sums of five variables to vanish while
    the rows ask x_i <= -1; least squares on both gives x_i = -1/2, slack 1/2 on all `nrow` rows (the working set of the exact
    active-set solve then holds nrow rows)"""
    A = np.zeros((nrow, nv))
    A[np.arange(nrow), np.arange(nrow)] = 1.0
    return dict(m=nrow, e=8, A=A, a=np.ones(nrow), B=np.kron(np.eye(8), np.ones((1, 5))) / np.sqrt(5.0), b=np.zeros(8))


def test_emulated_hqp_level_beyond_working_set_capacity_fails_with_status_0():
    """kHqpMaxQ = 32 working-set rows (dwbc_hqp.h) is a capacity of the device solver.  A level that needs
    30 active rows is solved; one that needs 40 returns status 0 and a zero answer -- never a truncated working set's point."""
    from tests.emu.emu import EmuHQP

    nv = 40
    for nrow, want in ((30, 1), (40, 0)):
        d = _capacity_problem(nrow, nv)
        hq, ok = _oracle_generic([d], nv)
        assert ok == 1 and len(hq.hqp_hs_[0].working_set_) == nrow  # the restatement has no capacity: all rows are active
        h = hq.hqp_hs_[0]
        eh = EmuHQP(1, nv, [nrow], [8], [0], share_cost=False, solve_first=True)
        eh.block(0, 0, (nrow, nv))[0] = h.A_
        eh.block(0, 1, (nrow,))[0] = h.a_
        eh.block(0, 2, (8, nv))[0] = h.B_
        eh.block(0, 3, (8,))[0] = h.b_
        eh.solve()
        assert eh.status(0)[0] == want
        if want:
            assert np.abs(eh.block(0, 5, (nv,))[0] - h.y_ans_).max() < TOL
        else:
            assert np.abs(eh.block(0, 5, (nv,))[0]).max() == 0.0


@pytest.mark.gpu
def test_gpu_hqp_level_beyond_working_set_capacity_fails_with_status_0():
    import libdwbc_amd as D
    from libdwbc_amd import hqp as Hq

    nv, B = 40, 4
    for nrow, want in ((30, 1), (40, 0)):
        d = _capacity_problem(nrow, nv)
        ref, ok = _oracle_generic([d], nv)
        hq = D.HQP(B, nv, 0, 0)
        hq.addHierarchy(nrow, 8)
        rep = lambda x: np.repeat(x[None], B, axis=0)
        hq.updateConstraintMatrix(0, rep(d["A"]), rep(d["a"]), rep(d["B"]), rep(d["b"]))
        hq.normalizeConstraintMatrix(0)
        hq.prepare()
        hq.solvefirst()
        assert (hq.get(0, Hq.STATUS) == want).all()
        if want:
            assert np.abs(hq.y_ans(0) - ref.hqp_hs_[0].y_ans_[None]).max() < TOL
        else:
            assert np.abs(hq.y_ans(0)).max() == 0.0

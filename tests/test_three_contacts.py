"""Three simultaneously active 6D contacts: the reference stacks every flagged contact (src/dwbc.cpp:445-453) and its
own tests register both hands next to the feet (tests/dwbc_test.cpp:68-69).  The product kernels stack two; a batch that opts in with
``set_max_active_contacts(3)`` runs the general-contact kernel (libdwbc_amd/csrc/dwbc_cycle_gc.h).

not-gpu: the two restatements (plain C and numpy) against each other on feet + left hand -- the reference's golden fixtures hold
two-contact states only, so three contacts are pinned by the two independent restatements of the same formulas, not by a fixture --
and the kernel source compiled for the host (tests/emu) against the C restatement.  gpu: the kernel through the C-ABI."""
import numpy as np
import pytest

from oracle import orc
from tests import cases
from tests.emu.emu import Emu

TOL_TAU, TOL_WR = 1e-6, 1e-5

CONTACT_SETS = {
    "feet": [1, 1, 0, 0],
    "left_foot": [1, 0, 0, 0],
    "feet_left_hand": [1, 1, 1, 0],
    "feet_right_hand": [1, 1, 0, 1],
    "left_foot_both_hands": [1, 0, 1, 1],
}


def _oracle(q, fl, fs, tasks=cases.TASKS_2LEVEL, lim=cases.TAU_LIM):
    M = orc.make_model(cases.tocabi_model())
    S = orc.make_setup(cases.CONTACTS_4, tasks, lim)
    return orc.cycle_batch(M, S, q, fl, fs, 0)


def test_c_and_numpy_restatements_agree_on_feet_plus_left_hand():
    from oracle import dwbc_np

    B = 3
    q, _, fs = cases.synth_batch(B, seed=5, yaw=True)
    fl = np.tile(np.array(CONTACT_SETS["feet_left_hand"], np.uint8), (B, 1))
    tau_c, wr_c, st_c, _ = _oracle(q, fl, fs)
    assert st_c.all()
    for i in range(B):
        cy = dwbc_np.Cycle(cases.tocabi_model())
        for c in cases.CONTACTS_4:
            cy.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
        for lv, links in enumerate(cases.TASKS_2LEVEL):
            for mode, link, pt in links:
                cy.add_task(lv, mode, link, pt)
        cy.set_torque_limit(cases.TAU_LIM)
        tau = cy.run(q[i], list(fl[i].astype(bool)), [fs[i, :6], fs[i, 6:9]])
        assert cy.status == 1 and cy.cdof == 18
        assert np.abs(tau - tau_c[i].sum(axis=0)).max() < 1e-6
        assert np.abs(cy.contact_force(tau) - wr_c[i, :18]).max() < 1e-5


@pytest.mark.parametrize("name", list(CONTACT_SETS))
def test_emulated_general_contact_kernel_vs_oracle(name):
    """one, two and three active contacts through the same kernel source (LDS poisoned with NaN per instance)"""
    B = 16
    q, _, fs = cases.synth_batch(B, seed=11, yaw=True)
    fl = np.tile(np.array(CONTACT_SETS[name], np.uint8), (B, 1))
    e = Emu(cases.URDF, cases.CONTACTS_4, cases.TASKS_2LEVEL, cases.TAU_LIM)
    r = e.run_gc(q, fl, fs)
    tau_r, wr_r, st_r, _ = _oracle(q, fl, fs)
    assert (r["status"] == st_r).all() and st_r.mean() > 0.9
    ok = st_r == 1
    assert np.isfinite(r["tau"]).all() and np.isfinite(r["wrench"]).all()
    assert np.abs(r["tau"][ok] - tau_r[ok]).max() < TOL_TAU
    assert np.abs(r["wrench"][ok] - wr_r[ok][:, :18]).max() < TOL_WR
    if sum(CONTACT_SETS[name]) == 3:
        assert r["diag"][:, 4].max() > 0  # the 18-variable QP of the pelvis level really iterates


def test_emulated_general_contact_kernel_three_levels_and_mixed_flags():
    """left foot + both hands with the right foot swinging as a third task level (QPs of 18, 15 and 18 variables), and a batch whose
    instances differ in their contact sets"""
    B = 12
    q, _, fs = cases.synth_batch(B, seed=21, yaw=True, contact_mode="L", levels=3)
    fl = np.tile(np.array(CONTACT_SETS["left_foot_both_hands"], np.uint8), (B, 1))
    e = Emu(cases.URDF, cases.CONTACTS_4, cases.TASKS_3LEVEL_SWING_R, cases.TAU_LIM)
    r = e.run_gc(q, fl, fs)
    tau_r, wr_r, st_r, _ = _oracle(q, fl, fs, tasks=cases.TASKS_3LEVEL_SWING_R)
    ok = st_r == 1
    assert (r["status"] == st_r).all() and ok.mean() > 0.5
    assert np.abs(r["tau"][ok] - tau_r[ok]).max() < TOL_TAU
    assert np.abs(r["wrench"][ok] - wr_r[ok][:, :18]).max() < TOL_WR
    q, _, fs = cases.synth_batch(B, seed=22, yaw=True)
    sets = list(CONTACT_SETS.values())
    fl = np.array([sets[i % len(sets)] for i in range(B)], np.uint8)
    e = Emu(cases.URDF, cases.CONTACTS_4, cases.TASKS_2LEVEL, cases.TAU_LIM)
    r = e.run_gc(q, fl, fs)
    tau_r, wr_r, st_r, _ = _oracle(q, fl, fs)
    ok = st_r == 1
    assert (r["status"] == st_r).all() and ok.mean() > 0.9
    assert np.abs(r["tau"][ok] - tau_r[ok]).max() < TOL_TAU
    assert np.abs(r["wrench"][ok] - wr_r[ok][:, :18]).max() < TOL_WR


def test_emulated_general_contact_kernel_refuses_a_fourth_contact():
    """four flags: never solved with a subset -- status 0, zero torques and wrench"""
    q, _, fs = cases.synth_batch(2, seed=3)
    fl = np.array([[1, 1, 1, 1], [1, 1, 1, 0]], np.uint8)
    e = Emu(cases.URDF, cases.CONTACTS_4, cases.TASKS_2LEVEL, cases.TAU_LIM)
    r = e.run_gc(q, fl, fs)
    assert r["status"][0] == 0 and np.abs(r["tau"][0]).max() == 0.0 and np.abs(r["wrench"][0]).max() == 0.0
    assert r["status"][1] == 1


def _make_gpu(B, tasks=cases.TASKS_2LEVEL):
    import libdwbc_amd as D

    wbc = D.Batch(D.Model.from_urdf(cases.URDF), B, device=0)
    for c in cases.CONTACTS_4:
        wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    for lv, links in enumerate(tasks):
        for mode, link, pt in links:
            wbc.add_task(lv, mode, link, pt)
    wbc.set_torque_limit(np.array(cases.TAU_LIM))
    return wbc


@pytest.mark.gpu
def test_gpu_feet_and_left_hand_vs_oracle():
    """feet + left hand against the restatement at 1e-6 through the C-ABI; a batch with one, two and three active
    contacts side by side; the wrench output is (B, 18)"""
    B = 250
    q, _, fs = cases.synth_batch(B, seed=41, yaw=True)
    wbc = _make_gpu(B)
    wbc.set_max_active_contacts(3)
    assert wbc.max_active_contacts == 3 and "dwbc_cycle_kernel_gc<39, 34, 64, 6>" in wbc.kernel_name()
    for name in ("feet_left_hand", None):
        if name:
            fl = np.tile(np.array(CONTACT_SETS[name], np.uint8), (B, 1))
        else:
            sets = list(CONTACT_SETS.values())
            fl = np.array([sets[i % len(sets)] for i in range(B)], np.uint8)
        wbc.set_state(q); wbc.set_contact(fl); wbc.set_fstar_all(fs)
        wbc.solve()
        tau, wr, st = wbc.get("tau"), wbc.get("wrench"), wbc.get("status")
        assert wr.shape == (B, 18)
        tau_r, wr_r, st_r, _ = _oracle(q, fl, fs)
        ok = st_r == 1
        assert (st == st_r).all() and ok.mean() > 0.9
        assert np.abs(tau[ok] - tau_r[ok]).max() < TOL_TAU
        assert np.abs(wr[ok] - wr_r[ok][:, :18]).max() < TOL_WR


@pytest.mark.gpu
def test_gpu_three_contacts_three_levels_and_scope():
    import libdwbc_amd as D

    B = 64
    q, _, fs = cases.synth_batch(B, seed=43, yaw=True, contact_mode="L", levels=3)
    fl = np.tile(np.array(CONTACT_SETS["left_foot_both_hands"], np.uint8), (B, 1))
    wbc = _make_gpu(B, tasks=cases.TASKS_3LEVEL_SWING_R)
    with pytest.raises(D.batch.DwbcError, match="set_max_active_contacts"):
        wbc.set_contact(fl)  # without the opt-in a third flag is refused
    wbc.set_max_active_contacts(3)
    wbc.set_state(q); wbc.set_contact(fl); wbc.set_fstar_all(fs)
    wbc.solve()
    tau_r, wr_r, st_r, _ = _oracle(q, fl, fs, tasks=cases.TASKS_3LEVEL_SWING_R)
    ok = st_r == 1
    assert (wbc.get("status") == st_r).all() and ok.mean() > 0.5
    assert np.abs(wbc.get("tau")[ok] - tau_r[ok]).max() < TOL_TAU
    four = fl.copy()
    four[0] = 1
    with pytest.raises(D.batch.DwbcError, match="more than 3"):
        wbc.set_contact(four)
    with pytest.raises(D.batch.DwbcError, match="hqp = true"):
        wbc.solve(hqp=False)
    # back to the default: the product kernels serve the batch again, the wrench is (B, 12)
    two = np.tile(np.array(CONTACT_SETS["left_foot"], np.uint8), (B, 1))
    wbc.set_contact(two)
    wbc.set_max_active_contacts(2)
    wbc.solve()
    assert "kernel_gc" not in wbc.kernel_name() and wbc.get("wrench").shape == (B, 12)
    tau_r2, _, st_r2, _ = _oracle(q, two, fs, tasks=cases.TASKS_3LEVEL_SWING_R)
    ok = st_r2 == 1
    assert np.abs(wbc.get("tau")[ok] - tau_r2[ok]).max() < TOL_TAU


# ---- the two-contact product kernels on pairs that include a hand.  Found while pinning the three-contact kernel: a foot and a hand give
#      J A^-1 J^T rows of 1e-1 next to rows of 1e3 (the unscaled 12 x 12 sweep left the torques 1e-5 Nm off on every such pair) and working
#      sets whose contact block loses rank (the lexicographic solve blew up and fell back: 8e-3 Nm on one instance in 500)
PAIRS = [[1, 0, 1, 0], [0, 1, 0, 1], [0, 0, 1, 1], [1, 0, 0, 1], [0, 1, 1, 0], [1, 1, 0, 0]]


@pytest.mark.parametrize("compact", [False, True, "pair"])
def test_emulated_product_kernels_on_foot_and_hand_pairs(compact):
    B = 192
    q, _, fs = cases.synth_batch(B, seed=9100, yaw=True)
    rng = np.random.default_rng(0)
    fl = np.array([PAIRS[i] for i in rng.integers(0, len(PAIRS), B)], np.uint8)
    # instances of the 9100.. sweep that were 4e-3 .. 8e-3 Nm off before the rank rule of the lexicographic solve (QP canon 3)
    q2, _, fs2 = cases.synth_batch(1024, seed=9101, yaw=True)
    q, fs, fl = np.vstack([q, q2[[67, 901]]]), np.vstack([fs, fs2[[67, 901]]]), np.vstack([fl, [[0, 1, 1, 0], [1, 0, 1, 0]]]).astype(np.uint8)
    e = Emu(cases.URDF, cases.CONTACTS_4, cases.TASKS_2LEVEL, cases.TAU_LIM)
    r = e.run(q, fl, fs, compact=compact)
    tau_r, wr_r, st_r, _ = _oracle(q, fl, fs)
    ok = st_r == 1
    assert (r["status"] == st_r).all() and ok.mean() > 0.95
    assert np.abs(r["tau"][ok] - tau_r[ok]).max() < TOL_TAU
    assert np.abs(r["wrench"][ok] - wr_r[ok][:, [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11]]).max() < TOL_WR


@pytest.mark.gpu
def test_gpu_product_kernels_on_foot_and_hand_pairs():
    for B in (1024, 4096):  # the two-wave kernel, the compact one-wave kernel
        q, _, fs = cases.synth_batch(B, seed=9101, yaw=True)
        rng = np.random.default_rng(1)
        fl = np.array([PAIRS[i] for i in rng.integers(0, len(PAIRS), B)], np.uint8)
        fl[67], fl[901] = [0, 1, 1, 0], [1, 0, 1, 0]
        wbc = _make_gpu(B)
        wbc.set_state(q); wbc.set_contact(fl); wbc.set_fstar_all(fs)
        wbc.solve()
        assert "kernel_gc" not in wbc.kernel_name()
        tau_r, wr_r, st_r, _ = _oracle(q, fl, fs)
        ok = st_r == 1
        assert (wbc.get("status") == st_r).all() and ok.mean() > 0.95
        assert np.abs(wbc.get("tau")[ok] - tau_r[ok]).max() < TOL_TAU


# ---- three contacts on a model of another size (kernel packs carry the general-contact kernel for models of at most 40 dof)
def _variant37(tmp_path):
    from oracle import urdf_model

    path = cases.variant_urdf(tmp_path / "fixed_head.urdf", cases.HEAD_JOINTS)
    mo = urdf_model.load_urdf(path)
    names = list(mo["names"])
    links = [names.index(n) for n in ("L_AnkleRoll_Link", "R_AnkleRoll_Link", "L_Wrist2_Link", "Upperbody_Link")]
    return path, mo, links


def _oracle37(mo, links, q, fl, fs, lim):
    M = orc.make_model(mo)
    contacts = [dict(c, link=l) for c, l in zip(cases.CONTACTS_4[:3], links[:3])]
    S = orc.make_setup(contacts, [[(0, 0, (0, 0, 0))], [(6, links[3], (0, 0, 0))]], lim)
    return contacts, orc.cycle_batch(M, S, q, fl, fs, 0)


def test_emulated_general_contact_kernel_on_a_37_dof_model(tmp_path):
    from tests.test_model_packs import variant_states

    path, mo, links = _variant37(tmp_path)
    B = 8
    q, fs = variant_states(mo, B, seed=13)
    fl = np.tile(np.array([1, 1, 1], np.uint8), (B, 1))
    fl[1], fl[2] = [1, 0, 1], [1, 1, 0]
    lim = np.full(31, 300.0)
    contacts, (tau_r, wr_r, st_r, _) = _oracle37(mo, links, q, fl, fs, lim)
    e = Emu(path, contacts, [[(0, 0, (0, 0, 0))], [(6, links[3], (0, 0, 0))]], lim)
    r = e.run_gc(q, fl, fs)
    ok = st_r == 1
    assert (r["status"] == st_r).all() and ok.all()
    assert np.abs(r["tau"] - tau_r).max() < TOL_TAU
    assert np.abs(r["wrench"] - wr_r[:, :18]).max() < TOL_WR


@pytest.mark.gpu
def test_gpu_three_contacts_on_a_37_dof_model_through_its_kernel_pack(tmp_path):
    import libdwbc_amd as D
    from tests.test_model_packs import variant_states

    path, mo, links = _variant37(tmp_path)
    md = D.Model.from_urdf(path)
    cases.ensure_pack(md)
    B = 96
    q, fs = variant_states(mo, B, seed=17)
    fl = np.tile(np.array([1, 1, 1], np.uint8), (B, 1))
    fl[::5] = [1, 0, 1]
    lim = np.full(31, 300.0)
    contacts, (tau_r, wr_r, st_r, _) = _oracle37(mo, links, q, fl, fs, lim)
    wbc = D.Batch(md, B, device=0)
    for c in contacts:
        wbc.add_contact(md.link_id(mo["names"][c["link"]]), c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    wbc.add_task(0, D.TASK_LINK_6D, 0)
    wbc.add_task(1, D.TASK_LINK_ROTATION, md.link_id("Upperbody_Link"))
    wbc.set_torque_limit(lim)
    wbc.set_max_active_contacts(3)
    wbc.set_state(q); wbc.set_contact(fl); wbc.set_fstar_all(fs)
    wbc.solve()
    assert "dwbc_cycle_kernel_gc<37, 32, 64, 6>" in wbc.kernel_name()
    ok = st_r == 1
    assert (wbc.get("status") == st_r).all() and ok.mean() > 0.9
    assert np.abs(wbc.get("tau")[ok] - tau_r[ok]).max() < TOL_TAU
    assert np.abs(wbc.get("wrench")[ok] - wr_r[ok][:, :18]).max() < TOL_WR

"""Models other than TOCABI's size run through kernel packs (libdwbc_amd/csrc/dwbc_pack.hip): the reference is model-generic
(any URDF RBDL reads, src/dwbc.cpp:140-277; its harnesses load tests/dof_test/dyros_tocabi_dof{18..43}.urdf), the device kernels
are compiled per (system dof, bodies).  The variants here are made from the TOCABI fixture by fixing joints (RBDL's fixed-joint
merge, the same step both URDF readers restate), solved on the device and compared with the numpy restatement, which is
model-generic as written."""
import os
import re

import numpy as np
import pytest

from oracle import dwbc_np as Dn
from oracle import urdf_model
from tests import cases

HEAD = cases.HEAD_JOINTS
ARMS = [f"{s}_{j}_Joint" for s in "LR" for j in ("Shoulder1", "Shoulder2", "Shoulder3", "Armlink", "Elbow", "Forearm", "Wrist1", "Wrist2")]
VARIANTS = {"fixed_head": (HEAD, 37, 32), "fixed_arms": (ARMS, 23, 18)}
TOL_TAU = 1e-6  # BASELINE north star
variant_urdf = cases.variant_urdf


def variant_states(model, B, seed):
    """TOCABI's nominal stance restricted to the joints the variant keeps, + jitter; f* of the golden CASE 1 + jitter"""
    toc = cases.tocabi_model()
    idx = [toc["names"].index(nm) for nm in model["names"]]
    q0 = np.array(cases.Q_CASE[1])
    rng = np.random.default_rng(seed)
    n = model["ndof"]
    q = np.zeros((B, n + 1))
    q[:, :6] = q0[:6]
    for i in range(1, model["nb"]):
        q[:, 6 + i - 1] = q0[6 + idx[i] - 1]
    q[:, n] = 1.0
    q[:, 6:n] += 0.02 * rng.uniform(-1, 1, size=(B, n - 6))
    fs = np.concatenate([np.array(cases.FSTAR_CASE[1][0]) + 0.1 * rng.uniform(-1, 1, size=(B, 6)),
                         np.array(cases.FSTAR_CASE[1][1]) + 0.1 * rng.uniform(-1, 1, size=(B, 3))], axis=1)
    return q, fs


def oracle_cycle(model, links, q, fs, tau_lim):
    c = Dn.Cycle(model)
    for cc, l in zip(cases.CONTACTS_2, links[:2]):
        c.add_contact(l, cc["point"], cc["lx"], cc["ly"], cc["mu"], cc["muz"])
    c.add_task(0, 0, 0, (0, 0, 0))
    c.add_task(1, 6, links[2], (0, 0, 0))
    c.set_torque_limit(tau_lim)
    c.run(q, [1, 1], [fs[:6], fs[6:9]])
    return dict(status=c.status, tau_grav=c.tau_grav, tau_task=c.tau_task, tau_contact=c.tau_contact)


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_variant_models_parse_the_same_on_both_sides_of_the_boundary(tmp_path, name):
    """not-gpu: the C-ABI URDF reader and the oracle's reader agree on the merged model (sizes, tree, inertial parameters)"""
    import libdwbc_amd as D

    fixed, n, nb = VARIANTS[name]
    path = variant_urdf(tmp_path / f"{name}.urdf", fixed)
    mo = urdf_model.load_urdf(path)
    md = D.Model.from_urdf(path)
    assert (md.ndof, md.nb) == (n, nb) == (mo["ndof"], mo["nb"])
    a = md.arrays()
    assert [md.link_name(i) for i in range(nb)] == list(mo["names"])
    assert (a["parent"] == np.asarray(mo["parent"])).all()
    for k in ("R_T", "p_T", "axis", "mass", "com", "inertia"):
        assert np.abs(a[k] - np.asarray(mo[k])).max() < 1e-12, k
    assert abs(md.total_mass - cases.tocabi_model()["mass"].sum()) < 1e-9  # fixing joints moves mass, it does not remove it


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_emulated_other_model_sizes_match_oracle(tmp_path, name):
    """not-gpu: the kernel source instantiated for the pack sizes (tests/emu, TopoGeneric, two levels) against the restatement.
    The 32-body variant is the case that found the composite-inertia scan stopping one doubling short when the body count is a
    power of two."""
    from tests.emu.emu import Emu

    fixed, n, nb = VARIANTS[name]
    path = variant_urdf(tmp_path / f"{name}.urdf", fixed)
    mo = urdf_model.load_urdf(path)
    names = list(mo["names"])
    links = [names.index("L_AnkleRoll_Link"), names.index("R_AnkleRoll_Link"), names.index("Upperbody_Link")]
    lim = np.full(n - 6, 300.0)
    e = Emu(path, [dict(cc, link=l) for cc, l in zip(cases.CONTACTS_2, links[:2])], [[(0, 0, (0, 0, 0))], [(6, links[2], (0, 0, 0))]], lim)
    B = 6
    q, fs = variant_states(mo, B, seed=11)
    r = e.run(q, np.ones((B, 2), np.uint8), fs, dump=True)
    c = Dn.Cycle(mo)
    c.update_kinematics(q[0])
    assert np.abs(e.dump_field(r["dump"], "A", (n, n))[0] - c.A).max() < 1e-10 * np.abs(c.A).max()
    assert np.abs(e.dump_field(r["dump"], "A_inv", (n, n))[0] - c.A_inv).max() < 1e-9 * np.abs(c.A_inv).max()
    for b in range(B):
        o = oracle_cycle(mo, links, q[b], fs[b], lim)
        assert r["status"][b] == o["status"] == 1
        ref = np.stack([o["tau_grav"], o["tau_task"], o["tau_contact"]])
        assert np.abs(r["tau"][b] - ref).max() < TOL_TAU


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_gpu_other_model_sizes_match_oracle(tmp_path, name):
    import libdwbc_amd as D

    fixed, n, nb = VARIANTS[name]
    path = variant_urdf(tmp_path / f"{name}.urdf", fixed)
    mo = urdf_model.load_urdf(path)
    md = D.Model.from_urdf(path)
    cases.ensure_pack(md)
    B = 48
    q, fs = variant_states(mo, B, seed=7)
    links = [md.link_id("L_AnkleRoll_Link"), md.link_id("R_AnkleRoll_Link"), md.link_id("Upperbody_Link")]
    assert min(links) > 0
    lim = np.full(n - 6, 300.0)
    wbc = D.Batch(md, B, device=0)
    for cc, l in zip(cases.CONTACTS_2, links[:2]):
        wbc.add_contact(l, cc["point"], cc["lx"], cc["ly"], cc["mu"], cc["muz"])
    wbc.add_task(0, D.TASK_LINK_6D, 0)
    wbc.add_task(1, D.TASK_LINK_ROTATION, links[2])
    wbc.set_torque_limit(lim)
    fl = np.ones((B, 2), np.uint8)
    for dump in (False, True):  # the lean and the full build of the pack
        wbc.enable_dump(dump)
        wbc.set_state(q); wbc.set_contact(fl); wbc.set_fstar_all(fs)
        wbc.solve()
        tau, st = wbc.get("tau"), wbc.get("status")
        assert f"<{n}, {nb}," in wbc.kernel_name()  # (generic pack, or the tree-specific one when it has been built for this variant)
        nok = 0
        for b in range(B):
            r = oracle_cycle(mo, links, q[b], fs[b], lim)
            assert st[b] == r["status"]
            if not st[b]:
                continue
            nok += 1
            ref = np.stack([r["tau_grav"], r["tau_task"], r["tau_contact"]])
            assert np.abs(tau[b] - ref).max() < TOL_TAU, (name, b, np.abs(tau[b] - ref).max())
        assert nok >= B - 2
    A = wbc.get("A")
    c = Dn.Cycle(mo)
    c.update_kinematics(q[0])
    assert np.abs(A[0] - c.A).max() < 1e-9 * np.abs(c.A).max()
    wbc.set_torque_limit(None)
    with pytest.raises(D.DwbcError, match="no kernel"):
        wbc.solve(reduced=True)  # packs hold the full-model cycle only


@pytest.mark.gpu
def test_gpu_missing_pack_is_reported_with_the_command_that_builds_it(tmp_path, monkeypatch):
    import libdwbc_amd as D

    path = variant_urdf(tmp_path / "m.urdf", HEAD + ["Waist1_Joint"])
    md = D.Model.from_urdf(path)
    assert (md.ndof, md.nb) == (36, 31)
    with pytest.raises(D.DwbcError, match="make -C libdwbc_amd/csrc pack N=36 NB=31"):
        D.Batch(md, 4, device=0)


# ---- a model LARGER than TOCABI (43 dof / 38 bodies, the size of the reference's tests/dof_test/dyros_tocabi_dof43.urdf): four links
#      added to the right hand by model surgery (AddLink with revolute joints; tests/test_model_surgery.py checks the surgery itself)
def model_43():
    import libdwbc_amd as D

    md = D.Model.from_urdf(cases.URDF)
    par = "R_Wrist2_Link"
    for k in range(4):
        ax = [[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0]][k]
        ang = 0.3 * (k + 1)
        R = np.array([[np.cos(ang), -np.sin(ang), 0.0], [np.sin(ang), np.cos(ang), 0.0], [0.0, 0.0, 1.0]])
        md = md.add_link(par, f"finger{k}", 1, ax, R, [0.02 * (k + 1), -0.01 * k, -0.04], 0.2 + 0.05 * k, [0.005, 0.0, -0.01 * (k + 1)],
                         np.diag([2e-4, 3e-4, 1e-4]) * (k + 1))
        par = f"finger{k}"
    a = md.arrays()
    a["names"] = [md.link_name(i) for i in range(md.nb)]
    return md, a


def states_43(B, seed):
    rng = np.random.default_rng(seed)
    q = np.zeros((B, 44))
    q0 = np.array(cases.Q_CASE[1])
    q[:, :39] = q0[:39]
    q[:, 39:43] = rng.uniform(-0.5, 0.5, size=(B, 4))
    q[:, 43] = 1.0
    q[:, 6:39] += 0.02 * rng.uniform(-1, 1, size=(B, 33))
    fs = np.concatenate([np.array(cases.FSTAR_CASE[1][0]) + 0.1 * rng.uniform(-1, 1, size=(B, 6)),
                         np.array(cases.FSTAR_CASE[1][1]) + 0.1 * rng.uniform(-1, 1, size=(B, 3))], axis=1)
    return q, fs


def test_emulated_43_dof_model_matches_oracle():
    from tests.emu.emu import Emu

    md, mo = model_43()
    assert (md.ndof, md.nb) == (43, 38)
    links = [6, 12, 15]
    lim = np.full(37, 300.0)
    e = Emu(mo, [dict(cc, link=l) for cc, l in zip(cases.CONTACTS_2, links[:2])], [[(0, 0, (0, 0, 0))], [(6, 15, (0, 0, 0))]], lim)
    B = 4
    q, fs = states_43(B, 5)
    r = e.run(q, np.ones((B, 2), np.uint8), fs, dump=True)
    c = Dn.Cycle(mo)
    c.update_kinematics(q[0])
    assert np.abs(e.dump_field(r["dump"], "A", (43, 43))[0] - c.A).max() < 1e-10 * np.abs(c.A).max()
    assert np.abs(e.dump_field(r["dump"], "A_inv", (43, 43))[0] - c.A_inv).max() < 1e-9 * np.abs(c.A_inv).max()
    for b in range(B):
        o = oracle_cycle(mo, links, q[b], fs[b], lim)
        assert r["status"][b] == o["status"] == 1
        assert np.abs(r["tau"][b] - np.stack([o["tau_grav"], o["tau_task"], o["tau_contact"]])).max() < TOL_TAU


@pytest.mark.gpu
def test_gpu_43_dof_model_matches_oracle():
    import libdwbc_amd as D

    md, mo = model_43()
    cases.ensure_pack(md)
    B = 32
    q, fs = states_43(B, 9)
    links = [6, 12, 15]
    lim = np.full(37, 300.0)
    wbc = D.Batch(md, B, device=0)
    for cc, l in zip(cases.CONTACTS_2, links[:2]):
        wbc.add_contact(l, cc["point"], cc["lx"], cc["ly"], cc["mu"], cc["muz"])
    wbc.add_task(0, D.TASK_LINK_6D, 0)
    wbc.add_task(1, D.TASK_LINK_ROTATION, 15)
    wbc.set_torque_limit(lim)
    for dump in (False, True):
        wbc.enable_dump(dump)
        wbc.set_state(q); wbc.set_contact(np.ones((B, 2), np.uint8)); wbc.set_fstar_all(fs)
        wbc.solve()
        tau, st = wbc.get("tau"), wbc.get("status")
        assert "<43, 38," in wbc.kernel_name()
        for b in range(B):
            o = oracle_cycle(mo, links, q[b], fs[b], lim)
            assert st[b] == o["status"] == 1
            assert np.abs(tau[b] - np.stack([o["tau_grav"], o["tau_task"], o["tau_contact"]])).max() < TOL_TAU, (b, dump)


@pytest.mark.gpu
def test_gpu_tree_specific_pack_is_preferred_and_only_for_its_own_tree(tmp_path):
    """A pack built for ONE parent table (TopoPack: tree-sparse sweep like the built-in TOCABI kernels) serves exactly that tree;
    another 37-dof / 32-body tree keeps the generic pack of the size."""
    import libdwbc_amd as D

    path = variant_urdf(tmp_path / "fh.urdf", HEAD)
    mo = urdf_model.load_urdf(path)
    md = D.Model.from_urdf(path)
    cases.ensure_pack(md)
    cases.ensure_pack(md, tree=True)
    other = D.Model.from_urdf(variant_urdf(tmp_path / "fw.urdf", ["L_Wrist2_Joint", "R_Wrist2_Joint"]))  # same size, other tree
    assert (other.ndof, other.nb) == (md.ndof, md.nb) and list(other.arrays()["parent"]) != list(md.arrays()["parent"])
    B = 32
    q, fs = variant_states(mo, B, seed=13)
    links = [md.link_id("L_AnkleRoll_Link"), md.link_id("R_AnkleRoll_Link"), md.link_id("Upperbody_Link")]
    lim = np.full(31, 300.0)

    def run(model, qq):
        wbc = D.Batch(model, B, device=0)
        for cc, l in zip(cases.CONTACTS_2, ("L_AnkleRoll_Link", "R_AnkleRoll_Link")):
            wbc.add_contact(model.link_id(l), cc["point"], cc["lx"], cc["ly"], cc["mu"], cc["muz"])
        wbc.add_task(0, D.TASK_LINK_6D, 0)
        wbc.add_task(1, D.TASK_LINK_ROTATION, model.link_id("Upperbody_Link"))
        wbc.set_torque_limit(lim)
        wbc.set_state(qq); wbc.set_contact(np.ones((B, 2), np.uint8)); wbc.set_fstar_all(fs)
        wbc.solve()
        return wbc.kernel_name(), wbc.get("tau"), wbc.get("status")

    name, tau, st = run(md, q)
    assert "TopoPack" in name and "<37, 32," in name
    for b in range(B):
        o = oracle_cycle(mo, links, q[b], fs[b], lim)
        assert st[b] == o["status"] == 1
        assert np.abs(tau[b] - np.stack([o["tau_grav"], o["tau_task"], o["tau_contact"]])).max() < TOL_TAU
    mo2 = urdf_model.load_urdf(str(tmp_path / "fw.urdf"))
    q2, _ = variant_states(mo2, B, seed=13)
    name2, tau2, st2 = run(other, q2)
    assert "TopoGeneric" in name2
    links2 = [other.link_id("L_AnkleRoll_Link"), other.link_id("R_AnkleRoll_Link"), other.link_id("Upperbody_Link")]
    for b in range(B):
        o = oracle_cycle(mo2, links2, q2[b], fs[b], lim)
        assert st2[b] == o["status"] == 1
        assert np.abs(tau2[b] - np.stack([o["tau_grav"], o["tau_task"], o["tau_contact"]])).max() < TOL_TAU

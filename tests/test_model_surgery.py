"""Init-time model surgery -- SURVEY 8 row f4: RobotData::DeleteLink / AddLink / ChangeLinkToFixedJoint / ChangeLinkInertia
(reference include/dwbc.h:206-226, src/dwbc.cpp:1764-2382, 2707-2730; only caller: tests/sp_test/test_mod.cpp).  The reference
edits its RBDL model in place; here every edit returns a new model (dwbc_model_*), checked against the URDF readers (the same
fixed-joint merge, RBDL Body::Join) and against the numpy rigid-body restatement."""
import numpy as np
import pytest

from oracle import dwbc_np as Dn
from oracle import urdf_model
from tests import cases
from tests.test_model_packs import HEAD, variant_urdf

LEFT_ARM = ["L_Shoulder1_Link", "L_Shoulder2_Link", "L_Shoulder3_Link", "L_Armlink_Link", "L_Elbow_Link", "L_Forearm_Link", "L_Wrist1_Link", "L_Wrist2_Link"]


def _tocabi():
    import libdwbc_amd as D

    return D.Model.from_urdf(cases.URDF)


def _as_oracle_model(md):
    a = md.arrays()
    a["names"] = [md.link_name(i) for i in range(md.nb)]
    return a


def test_change_link_to_fixed_joint_equals_the_urdf_with_those_joints_fixed(tmp_path):
    md = _tocabi()
    m2 = md.change_link_to_fixed_joint("Head_Link").change_link_to_fixed_joint("Neck_Link")  # leaf first: the reference deletes descendants
    ref = urdf_model.load_urdf(variant_urdf(tmp_path / "fh.urdf", HEAD))
    assert (m2.ndof, m2.nb) == (37, 32)
    assert [m2.link_name(i) for i in range(m2.nb)] == list(ref["names"])
    a = m2.arrays()
    assert (a["parent"] == np.asarray(ref["parent"])).all()
    for k in ("R_T", "p_T", "axis", "mass", "com", "inertia"):
        assert np.abs(a[k] - np.asarray(ref[k])).max() < 1e-12, k
    assert abs(m2.total_mass - md.total_mass) < 1e-9
    # the reference's ChangeLinkToFixedJoint on a link WITH descendants deletes them (src/dwbc.cpp:2375 DeleteLink, then AddLink of the
    # link alone): fixing the neck first loses the head's mass
    m3 = md.change_link_to_fixed_joint("Neck_Link")
    head_mass = md.arrays()["mass"][md.link_id("Head_Link")]
    assert m3.nb == 32 and abs(m3.total_mass - (md.total_mass - head_mass)) < 1e-9


def test_delete_link_removes_the_subtree_and_add_link_restores_the_dynamics():
    import libdwbc_amd as D

    md = _tocabi()
    a0 = md.arrays()
    ids = [md.link_id(n) for n in LEFT_ARM]
    m2 = md.delete_link("L_Shoulder1_Link")
    assert m2.nb == md.nb - 8 and all(m2.link_id(n) < 0 for n in LEFT_ARM)
    assert abs(m2.total_mass - (md.total_mass - a0["mass"][ids].sum())) < 1e-9
    keep = [i for i in range(md.nb) if i not in ids]
    a2 = m2.arrays()
    remap = {old: new for new, old in enumerate(keep)}
    assert [remap[p] for p in a0["parent"][keep][1:]] == list(a2["parent"][1:])
    for k in ("R_T", "p_T", "axis", "mass", "com", "inertia"):
        assert np.abs(a2[k] - a0[k][keep]).max() == 0.0
    # re-attach the arm link by link: the new links are the LAST ones, their joints the last coordinates (as with RBDL's AddBody)
    m3 = m2
    for n, i in zip(LEFT_ARM, ids):
        par = md.link_name(int(a0["parent"][i]))
        m3 = m3.add_link(par, n, 1, a0["axis"][i], a0["R_T"][i], a0["p_T"][i], a0["mass"][i], a0["com"][i], a0["inertia"][i])
    assert m3.nb == md.nb and abs(m3.total_mass - md.total_mass) < 1e-9
    assert [m3.link_name(i) for i in range(m3.nb)] == [md.link_name(i) for i in keep] + LEFT_ARM
    # same robot, other numbering: the joint-space mass matrix is the original's under the permutation of the coordinates
    order = keep + ids  # new body -> old body
    q_old = np.array(cases.Q_CASE[2])
    q_new = q_old.copy()
    for nb_, ob in enumerate(order):
        if nb_ > 0:
            q_new[6 + nb_ - 1] = q_old[6 + ob - 1]
    c_old, c_new = Dn.Cycle(_as_oracle_model(md)), Dn.Cycle(_as_oracle_model(m3))
    c_old.update_kinematics(q_old)
    c_new.update_kinematics(q_new)
    perm = list(range(6)) + [6 + ob - 1 for ob in order[1:]]
    assert np.abs(c_new.A - c_old.A[np.ix_(perm, perm)]).max() < 1e-10 * np.abs(c_old.A).max()
    assert np.abs(c_new.G - c_old.G[perm]).max() < 1e-9
    # a revolute link somewhere in the middle of the depth-first numbering is refused, with the reason
    i = ids[0]
    with pytest.raises(D.DwbcError, match="last link or one of its ancestors"):
        m2.add_link("L_AnkleRoll_Link", "extra", 1, a0["axis"][i], a0["R_T"][i], a0["p_T"][i], 1.0, a0["com"][i], a0["inertia"][i])
    with pytest.raises(D.DwbcError, match="base link cannot be deleted"):
        md.delete_link(0)


def test_add_link_with_a_fixed_joint_and_change_link_inertia():
    md = _tocabi()
    a0 = md.arrays()
    hand = md.link_id("R_Wrist2_Link")
    # a 0.5 kg tool bolted to the right hand: RBDL Body::Join
    R = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    p, c, I = np.array([0.0, 0.05, -0.1]), np.array([0.01, 0.0, -0.02]), np.diag([1e-3, 2e-3, 3e-3])
    m2 = md.add_link(hand, "tool", 0, [0, 0, 1], R, p, 0.5, c, I)
    a2 = m2.arrays()
    assert m2.nb == md.nb and abs(m2.total_mass - md.total_mass - 0.5) < 1e-12
    m1, c1, I1 = a0["mass"][hand], a0["com"][hand], a0["inertia"][hand]
    c2 = R @ c + p
    ct = (m1 * c1 + 0.5 * c2) / (m1 + 0.5)
    sk = lambda d: np.array([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]])
    It = I1 + m1 * sk(c1 - ct) @ sk(c1 - ct).T + R @ I @ R.T + 0.5 * sk(c2 - ct) @ sk(c2 - ct).T
    assert np.abs(a2["com"][hand] - ct).max() < 1e-14 and np.abs(a2["inertia"][hand] - It).max() < 1e-14
    others = [i for i in range(md.nb) if i != hand]
    assert np.abs(a2["inertia"][others] - a0["inertia"][others]).max() == 0.0
    m3 = md.change_link_inertia("Head_Link", np.diag([0.01, 0.02, 0.03]), [0.0, 0.0, 0.1], 2.5)
    a3 = m3.arrays()
    h = md.link_id("Head_Link")
    assert a3["mass"][h] == 2.5 and np.abs(a3["com"][h] - [0, 0, 0.1]).max() == 0 and np.abs(a3["inertia"][h] - np.diag([0.01, 0.02, 0.03])).max() == 0
    assert abs(m3.total_mass - (md.total_mass - a0["mass"][h] + 2.5)) < 1e-9


@pytest.mark.gpu
def test_gpu_solve_on_a_surgically_edited_model(tmp_path):
    """the 37-dof model made by ChangeLinkToFixedJoint solves like the one read from the URDF with those joints fixed"""
    import libdwbc_amd as D
    from tests.test_model_packs import variant_states

    md = _tocabi().change_link_to_fixed_joint("Head_Link").change_link_to_fixed_joint("Neck_Link")
    mu = D.Model.from_urdf(variant_urdf(tmp_path / "fh.urdf", HEAD))
    cases.ensure_pack(md)
    mo = urdf_model.load_urdf(str(tmp_path / "fh.urdf"))
    B = 32
    q, fs = variant_states(mo, B, seed=3)
    out = []
    for model in (md, mu):
        wbc = D.Batch(model, B, device=0)
        for cc, l in zip(cases.CONTACTS_2, ("L_AnkleRoll_Link", "R_AnkleRoll_Link")):
            wbc.add_contact(model.link_id(l), cc["point"], cc["lx"], cc["ly"], cc["mu"], cc["muz"])
        wbc.add_task(0, D.TASK_LINK_6D, 0)
        wbc.add_task(1, D.TASK_LINK_ROTATION, model.link_id("Upperbody_Link"))
        wbc.set_torque_limit(np.full(31, 300.0))
        wbc.set_state(q); wbc.set_contact(np.ones((B, 2), np.uint8)); wbc.set_fstar_all(fs)
        wbc.solve()
        out.append((wbc.get("tau"), wbc.get("status")))
    assert out[0][1].all() and (out[0][1] == out[1][1]).all()
    assert np.abs(out[0][0] - out[1][0]).max() < 1e-8

"""On-device task reference: TaskLink quintic / slerp trajectory + PD -> f*
(reference src/task.cpp:223-339, src/math.cpp:127-182,275-291, src/dwbc.cpp:708-780).

PARITY UNPINNED in the reference (nothing calls SetTrajectory* there).  The numpy restatement
oracle/task_reference_np.py is pinned by first-principles properties; the kernel (libdwbc_amd/csrc/dwbc_fstar.h) is
checked against it, through the f* it produces AND through the torques that follow from it."""
import numpy as np
import pytest

from oracle import dwbc_np as D
from oracle import task_reference_np as TR
from tests import cases

GAINS = dict(pos_p=np.array([400.0, 380, 360]), pos_d=np.array([40.0, 38, 36]), pos_a=np.array([1.0, 0.9, 1.1]),
             rot_p=np.array([300.0, 310, 320]), rot_d=np.array([30.0, 31, 32]))
G15 = np.concatenate([GAINS["pos_p"], GAINS["pos_d"], GAINS["pos_a"], GAINS["rot_p"], GAINS["rot_d"]])


def _rot(axis, ang):
    return D.axis_angle_R(np.asarray(axis, float) / np.linalg.norm(axis), ang)


def test_quintic_end_conditions():
    x = [TR.quintic_spline(t, 1.0, 3.0, 0.2, 0.5, 0.0, 1.4, -0.3, 0.0) for t in (1.0, 3.0, 0.5, 3.5)]
    assert np.allclose(x[0], [0.2, 0.5, 0.0]) and np.allclose(x[1], [1.4, -0.3, 0.0], atol=1e-12)
    assert np.allclose(x[2], [0.2, 0.5, 0.0]) and np.allclose(x[3], [1.4, -0.3, 0.0])
    h = 1e-6
    a, b, c = (TR.quintic_spline(2.0 + d, 1.0, 3.0, 0.2, 0.5, 0.0, 1.4, -0.3, 0.0) for d in (-h, 0.0, h))
    assert abs((c[0] - a[0]) / (2 * h) - b[1]) < 1e-8 and abs((c[1] - a[1]) / (2 * h) - b[2]) < 1e-7


def test_rotation_reference_properties():
    Ri, Rd = _rot([1, 2, 3], 0.4), _rot([1, 2, 3], 0.4) @ _rot([0.2, -1, 0.5], 1.1)
    qi, qd = TR.quat_from_R(Ri), TR.quat_from_R(Rd)
    assert np.abs(TR.quat_to_R(qi) - Ri).max() < 1e-14
    assert np.abs(TR.quat_to_R(TR.quat_slerp(qi, 0.0, qd)) - Ri).max() < 1e-14
    assert np.abs(TR.quat_to_R(TR.quat_slerp(qi, 1.0, qd)) - Rd).max() < 1e-14
    ang, ax = TR.angle_axis_from_quat(TR.quat_mul(qd, TR.quat_inverse(qi)))
    assert np.abs(_rot(ax, ang) @ Ri - Rd).max() < 1e-14
    # trace branch of Quaternion(Matrix3): a rotation by ~pi
    Rp = _rot([0.3, 1, -0.2], 3.1)
    assert np.abs(TR.quat_to_R(TR.quat_from_R(Rp)) - Rp).max() < 1e-14
    # GetPhi of a small relative rotation is its rotation vector
    e = np.array([1e-5, -2e-5, 3e-5])
    assert np.abs(TR.get_phi(Ri, _rot(e, np.linalg.norm(e)) @ Ri) - e).max() < 1e-9


def _setup(B, seed):
    rng = np.random.default_rng(seed)
    q, fl, fs = cases.synth_batch(B, seed=seed, yaw=True)
    qd = 0.3 * rng.uniform(-1, 1, (B, 39))
    m = cases.tocabi_model()
    ctime = rng.uniform(0.2, 1.8, B)
    ctime[0] = -0.1  # before the start
    if B > 1:
        ctime[1] = 2.5  # after the end
    # level 0: pelvis 6D (pos + rot trajectory); level 1: upper body rotation (rot trajectory)
    traj0, traj1 = np.zeros((B, 34)), np.zeros((B, 34))
    fexp = fs.copy()
    for b in range(B):
        R, p = D.forward_kinematics(m, q[b])
        v, w, _ = D.link_velocities(m, R, p, qd[b])
        tr0 = dict(t0=0.0, t1=2.0, pos_init=p[0], vel_init=rng.uniform(-0.1, 0.1, 3), pos_des=p[0] + rng.uniform(-0.05, 0.05, 3),
                   vel_des=np.zeros(3), rot_init=R[0], rot_des=R[0] @ _rot(rng.uniform(-1, 1, 3), 0.3), has_pos=1, has_rot=1)
        tr1 = dict(t0=0.0, t1=2.0, pos_init=np.zeros(3), vel_init=np.zeros(3), pos_des=np.zeros(3), vel_des=np.zeros(3),
                   rot_init=R[15], rot_des=_rot(rng.uniform(-1, 1, 3), 2.9) @ R[15], has_pos=0, has_rot=1)
        traj0[b], traj1[b] = TR.pack_traj(tr0), TR.pack_traj(tr1)
        fexp[b, 0:6] = TR.link_fstar(D.TASK_LINK_6D, ctime[b], tr0, GAINS, 0, np.zeros(3), m["com"][0], R, p, v, w, fs[b, 0:6])
        fexp[b, 6:9] = TR.link_fstar(D.TASK_LINK_ROTATION, ctime[b], tr1, GAINS, 15, np.zeros(3), m["com"][15], R, p, v, w, fs[b, 6:9])
    return q, qd, fl, fs, ctime, traj0, traj1, fexp


def test_emulated_task_reference_vs_oracle():
    from tests.emu.emu import Emu

    B = 6
    q, qd, fl, fs, ctime, traj0, traj1, fexp = _setup(B, 31)
    e = Emu(cases.URDF, cases.CONTACTS_2, cases.TASKS_2LEVEL, cases.TAU_LIM)
    e.set_traj(0, 0, 0, G15)
    e.set_traj(1, 0, 1, G15)
    traj = np.stack([traj0, traj1], axis=1)  # (B, n_traj, 34)
    r = e.run(q, fl, fs, qdot=qd, traj=traj, ctime=ctime)
    e2 = Emu(cases.URDF, cases.CONTACTS_2, cases.TASKS_2LEVEL, cases.TAU_LIM)
    r2 = e2.run(q, fl, fexp)  # the same cycle fed with the oracle's f*
    assert np.abs(fexp - fs).max() > 0.1
    assert (r["status"] == r2["status"]).all()
    ok = r2["status"] == 1
    assert np.abs(r["tau"][ok] - r2["tau"][ok]).max() < 1e-7


@pytest.mark.gpu
def test_gpu_task_reference_vs_oracle():
    import libdwbc_amd as Dw

    B = 16
    q, qd, fl, fs, ctime, traj0, traj1, fexp = _setup(B, 32)

    def mk():
        wbc = Dw.Batch(Dw.Model.from_urdf(cases.URDF), B, device=0)
        for c in cases.CONTACTS_2:
            wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
        wbc.add_task(0, Dw.TASK_LINK_6D, 0)
        wbc.add_task(1, Dw.TASK_LINK_ROTATION, 15)
        wbc.set_torque_limit(np.array(cases.TAU_LIM))
        wbc.set_contact(fl)
        return wbc

    a = mk()
    a.set_state(q, qd)
    a.set_fstar_all(fs)
    for lv, tr in ((0, traj0), (1, traj1)):
        a.set_task_gain(lv, 0, GAINS["pos_p"], GAINS["pos_d"], GAINS["pos_a"], GAINS["rot_p"], GAINS["rot_d"])
        a.set_trajectory(lv, 0, tr)
    a.set_control_time(ctime)
    a.solve()
    b = mk()
    b.set_state(q)
    b.set_fstar_all(fexp)
    b.solve()
    sa, sb = a.get("status"), b.get("status")
    assert (sa == sb).all()
    ok = sb == 1
    assert np.abs(a.get("tau")[ok] - b.get("tau")[ok]).max() < 1e-7
    # clearing a trajectory returns that link to the SetTaskSpace values
    a.set_trajectory(0, 0, None)
    a.set_trajectory(1, 0, None)
    a.solve()
    c = mk()
    c.set_state(q)
    c.set_fstar_all(fs)
    c.solve()
    assert np.abs(a.get("tau") - c.get("tau")).max() < 1e-9

"""Velocity-dependent outputs of UpdateKinematics(q, qdot, qddot) (reference src/dwbc.cpp:279-371, src/link.cpp:76-96):
B_ = C qdot + g (RBDL NonlinearEffects [ext]) and the link velocities.  The reference holds no fixture for them
(PARITY UNPINNED); the numpy restatement is pinned here by first principles instead -- Lagrange's equations on the true
coordinates, Euler-Poincare on the base rotation, finite differences of the forward kinematics -- and the kernel
(libdwbc_amd/csrc/dwbc_velocity.h) is checked against the restatement."""
import numpy as np
import pytest

from oracle import dwbc_np as D
from tests import cases


def _integrate(q, dq, h):
    """q (+) h dq on the configuration manifold: base rotation advanced by the BODY-frame rotation vector h dq[3:6]"""
    qn = q.copy()
    qn[0:3] += h * dq[0:3]
    qn[6:39] += h * dq[6:39]
    w = h * dq[3:6]
    ang = np.linalg.norm(w)
    d = np.array([0, 0, 0, 1.0]) if ang < 1e-14 else np.concatenate([np.sin(ang / 2) * w / ang, [np.cos(ang / 2)]])
    x, y, z, wq = q[3], q[4], q[5], q[39]
    a, b, c, dd = d
    qn[3] = wq * a + x * dd + y * c - z * b
    qn[4] = wq * b - x * c + y * dd + z * a
    qn[5] = wq * c + x * b - y * a + z * dd
    qn[39] = wq * dd - x * a - y * b - z * c
    return qn


def _state(seed):
    rng = np.random.default_rng(seed)
    q = np.array(cases.Q_CASE[2], float) + 0.05 * rng.uniform(-1, 1, 40)
    qu = cases.yaw_quat(0.7, 0.1, -0.2)
    q[3:6], q[39] = qu[:3], qu[3]
    return q, rng.uniform(-1, 1, 39)


def test_nonlinear_effects_satisfy_lagrange_and_euler_poincare():
    m = cases.tocabi_model()
    q, qd = _state(0)
    cy = D.Cycle(m)
    cy.update_kinematics(q)
    assert np.abs(D.nonlinear_effects(m, q, np.zeros(39)) - cy.G).max() < 1e-10  # B_(q, 0) = G_ (pinned: torque_grav_ goldens)
    B = D.nonlinear_effects(m, q, qd)
    h = 1e-6
    T = lambda q_: 0.5 * qd @ D.crba(m, q_) @ qd
    p = lambda q_: D.crba(m, q_) @ qd
    ddt = (p(_integrate(q, qd, h)) - p(_integrate(q, qd, -h))) / (2 * h)  # d/dt (A qdot) at zero generalised acceleration
    dT = np.zeros(39)
    for j in range(39):
        e = np.zeros(39)
        e[j] = 1
        dT[j] = (T(_integrate(q, e, h)) - T(_integrate(q, e, -h))) / (2 * h)
    res = ddt - dT + cy.G - B
    res[3:6] += np.cross(qd[3:6], p(q)[3:6])  # Euler-Poincare term of the body-frame angular velocity
    assert np.abs(B - cy.G).max() > 1.0 and np.abs(res).max() < 1e-6


def test_link_velocities_match_finite_differences():
    m = cases.tocabi_model()
    q, qd = _state(1)
    R, p = D.forward_kinematics(m, q)
    v, w, vi = D.link_velocities(m, R, p, qd)
    h = 1e-6
    Rp, pp = D.forward_kinematics(m, _integrate(q, qd, h))
    Rm, pm = D.forward_kinematics(m, _integrate(q, qd, -h))
    assert np.abs((pp - pm) / (2 * h) - v).max() < 1e-8
    for i in (0, 6, 15, 33):
        Wd = (Rp[i] - Rm[i]) / (2 * h) @ R[i].T  # skew(w)
        assert np.abs(np.array([Wd[2, 1], Wd[0, 2], Wd[1, 0]]) - w[i]).max() < 1e-8


def _check(out_B, out_v, out_w, q, qd):
    m = cases.tocabi_model()
    for b in range(q.shape[0]):
        R, p = D.forward_kinematics(m, q[b])
        v, w, _ = D.link_velocities(m, R, p, qd[b])
        assert np.abs(out_B[b] - D.nonlinear_effects(m, q[b], qd[b])).max() < 1e-9
        assert np.abs(out_v[b][:34] - v).max() < 1e-11 and np.abs(out_w[b][:34] - w).max() < 1e-11


def test_emulated_kernel_velocity_outputs():
    from tests.emu.emu import Emu

    B = 3
    q, fl, fs = cases.synth_batch(B, seed=5, yaw=True)
    qd = np.random.default_rng(5).uniform(-1, 1, (B, 39))
    e = Emu(cases.URDF, cases.CONTACTS_2, cases.TASKS_2LEVEL, cases.TAU_LIM)
    r = e.run(q, fl, fs, dump=True, qdot=qd)
    d = r["dump"]
    _check(e.dump_field(d, "B", (39,)), e.dump_field(d, "link_v", (48, 3)), e.dump_field(d, "link_w", (48, 3)), q, qd)
    r0 = e.run(q, fl, fs)  # the torque path ignores qdot
    assert np.abs(r["tau"] - r0["tau"]).max() == 0.0


@pytest.mark.gpu
def test_gpu_velocity_outputs():
    import libdwbc_amd as Dw

    B = 8
    q, fl, fs = cases.synth_batch(B, seed=6, yaw=True)
    qd = np.random.default_rng(6).uniform(-1, 1, (B, 39))
    wbc = Dw.Batch(Dw.Model.from_urdf(cases.URDF), B, device=0)
    for c in cases.CONTACTS_2:
        wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    wbc.add_task(0, Dw.TASK_LINK_6D, 0)
    wbc.add_task(1, Dw.TASK_LINK_ROTATION, 15)
    wbc.set_torque_limit(np.array(cases.TAU_LIM))
    wbc.enable_dump(True)
    wbc.set_state(q, qd)
    wbc.set_contact(fl)
    wbc.set_fstar_all(fs)
    wbc.solve()
    _check(wbc.get("B"), wbc.get("link_v"), wbc.get("link_w"), q, qd)

"""The TOCABI set-up and the seeded synthetic state batches of the BASELINE configurations.

Product-side home of what `bench.py`'s HIP engine, `__graft_entry__.smoke()` and the tools need to pose the workload: the robot
description that every reference test, example and BASELINE config loads, the contact / task definitions of the reference's
harness (tests/dwbc_test.cpp:48-77,152-181 in the reference tree) and the input recipe.  `tests/cases.py` re-exports these names,
so the parity tests and the bench pose exactly the same problems.  Nothing here imports `oracle/` or `tests/`.
"""
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
TOCABI_URDF = os.path.join(_HERE, "data", "dyros_tocabi.urdf")
URDF = TOCABI_URDF

TASK_LINK_6D, TASK_LINK_ROTATION = 0, 6

# reference tests/dwbc_test.cpp:48-54 (case 1) and :152-158 (case 2)
Q_CASE = {
    1: [0, 0, 0.92983, 0, 0, 0, 0.0, 0.0, -0.24, 0.6, -0.36, 0.0, 0.0, 0.0, -0.24, 0.6, -0.36, 0.0, 0, 0, 0,
        0.3, 0.3, 1.5, -1.27, -1, 0, -1, 0, 0, 0, -0.3, -0.3, -1.5, 1.27, 1, 0, 1, 0, 1],
    2: [0, 0, 0.92983, 0, 0, 0, 0.1, 0.0, -0.24, 0.5, -0.6, 0.0, 0.05, 0.0, -0.21, 0.7, -0.31, 0.0, 0, 0, 0,
        0.2, 0.5, 1.5, -1.27, -1.2, 0, -1, 0, 0, 0, -0.3, -0.3, -1.5, 1.27, 1.3, 0.1, 1.3, 0, 1],
}
FSTAR_CASE = {1: ([0.1, 4.0, 0.1, 0.1, -0.1, 0.1], [0.1, -0.1, 0.1]), 2: ([0.4, 2.0, 0.1, 0.3, -0.1, 0.1], [0.1, 0.1, 0.1])}

FOOT_POINT = (0.03, 0.0, -0.1585)
# reference tests/dwbc_test.cpp:66-69: four registered contacts (feet, hands); the reference's harness only ever enables the feet
CONTACTS_4 = [
    dict(link=6, point=FOOT_POINT, lx=0.15, ly=0.075, mu=0.2, muz=0.2),
    dict(link=12, point=FOOT_POINT, lx=0.15, ly=0.075, mu=0.2, muz=0.2),
    dict(link=23, point=FOOT_POINT, lx=0.04, ly=0.04, mu=0.2, muz=0.2),
    dict(link=31, point=FOOT_POINT, lx=0.04, ly=0.04, mu=0.2, muz=0.2),
]
CONTACTS_2 = CONTACTS_4[:2]
TASKS_2LEVEL = [[(TASK_LINK_6D, 0, (0, 0, 0))], [(TASK_LINK_ROTATION, 15, (0, 0, 0))]]
# single support + swing foot as a third level
TASKS_3LEVEL_SWING_R = TASKS_2LEVEL + [[(TASK_LINK_6D, 12, (0, 0, 0))]]
TASKS_3LEVEL_SWING_L = TASKS_2LEVEL + [[(TASK_LINK_6D, 6, (0, 0, 0))]]
TAU_LIM = [300.0] * 33


def yaw_quat(yaw, roll=0.0, pitch=0.0):
    """quaternion (x,y,z,w) of Rx(roll)*Ry(pitch)*Rz(yaw) -- the composition used by reference
    tests/dwbc_test.cpp:268-271 (AngleAxis X * Y * Z)."""
    def q_axis(a, ang):
        s = np.sin(ang / 2)
        return np.array([a[0] * s, a[1] * s, a[2] * s, np.cos(ang / 2)])

    def qmul(p, q):
        px, py, pz, pw = p
        qx, qy, qz, qw = q
        return np.array([
            pw * qx + px * qw + py * qz - pz * qy,
            pw * qy - px * qz + py * qw + pz * qx,
            pw * qz + px * qy - py * qx + pz * qw,
            pw * qw - px * qx - py * qy - pz * qz,
        ])

    return qmul(qmul(q_axis((1, 0, 0), roll), q_axis((0, 1, 0), pitch)), q_axis((0, 0, 1), yaw))


def synth_batch(B, seed=20251226, yaw=False, contact_mode="LR", levels=2):
    """Seeded synthetic TOCABI states: q = nominal stance + 0.01 U(-1,1), identity or random-yaw
    base orientation, f* = fixture values + 0.1 U.  contact_mode: 'LR' | 'L' | 'R' | 'mixed'.
    Returns q (B,40), flags (B,2) uint8, fstar (B, 9 or 15)."""
    rng = np.random.Generator(np.random.Philox(seed))
    q0 = np.array(Q_CASE[1], dtype=np.float64)
    q = q0[None, :] + 0.01 * rng.uniform(-1, 1, size=(B, 40))
    q[:, 3:6] = 0.0
    q[:, 39] = 1.0
    if yaw:
        ya = rng.uniform(-np.pi, np.pi, size=B)
        ro = rng.uniform(-0.1, 0.1, size=B)
        pi = rng.uniform(-0.1, 0.1, size=B)
        for b in range(B):
            qu = yaw_quat(ya[b], ro[b], pi[b])
            q[b, 3:6] = qu[:3]
            q[b, 39] = qu[3]
    f0 = np.array(FSTAR_CASE[1][0]) + 0.1 * rng.uniform(-1, 1, size=(B, 6))
    f1 = np.array(FSTAR_CASE[1][1]) + 0.1 * rng.uniform(-1, 1, size=(B, 3))
    fs = [f0, f1]
    if levels == 3:
        fs.append(np.array([0, 0, 0.5, 0, 0, 0.0]) + 0.1 * rng.uniform(-1, 1, size=(B, 6)))
    fstar = np.concatenate(fs, axis=1)
    flags = np.ones((B, 2), dtype=np.uint8)
    if contact_mode == "L":
        flags[:, 1] = 0
    elif contact_mode == "R":
        flags[:, 0] = 0
    elif contact_mode == "mixed":
        u = rng.uniform(0, 1, size=B)
        flags[(u >= 0.5) & (u < 0.75), 1] = 0
        flags[u >= 0.75, 0] = 0
    return q, flags, fstar

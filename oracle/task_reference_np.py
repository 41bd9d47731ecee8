"""numpy restatement of libdwbc's task-reference front end (TaskLink trajectory + PD -> f*).   TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

PARITY UNPINNED: no test, example or fixture of the reference calls SetTrajectoryQuintic / SetTrajectoryRotation /
SetTaskGain (they are only declared, include/dwbc_task.h:104-108); the restatement follows the source and is pinned by
first-principles properties in tests/test_task_reference.py (end-point / derivative conditions of the quintic,
slerp end points, orientation error of a small rotation).

Reference functions followed (file:line in /root/reference):
  src/math.cpp:127-182   QuinticSpline
  src/math.cpp:275-291   GetPhi
  src/task.cpp:223-266   SetTrajectoryQuintic / SetTrajectoryRotation / SetTaskGain
  src/task.cpp:268-293   TaskLink::GetFstarPosPD
  src/task.cpp:295-339   TaskLink::GetFstarRotPD
  src/dwbc.cpp:685-793   RobotData::UpdateTaskSpace (which point / velocity of the link each mode feeds in)
Eigen arithmetic restated from its published algorithms [ext]: Quaternion(Matrix3) (trace method), Quaternion::slerp,
Quaternion::toRotationMatrix, AngleAxis(Quaternion).
"""
import numpy as np

from .dwbc_np import (TASK_LINK_6D, TASK_LINK_6D_COM_FRAME, TASK_LINK_6D_CUSTOM_FRAME, TASK_LINK_POSITION,
                      TASK_LINK_POSITION_COM_FRAME, TASK_LINK_POSITION_CUSTOM_FRAME, TASK_LINK_ROTATION,
                      TASK_LINK_ROTATION_CUSTOM_FRAME)

TRAJ_STRIDE = 34  # t0 t1 | pos_init vel_init pos_des vel_des (12) | rot_init (9) rot_des (9) | has_pos has_rot


def quintic_spline(t, t0, tf, x0, xd0, xdd0, xf, xdf, xddf):
    """src/math.cpp:127-182 -> (position, velocity, acceleration)"""
    if t < t0:
        return np.array([x0, xd0, xdd0])
    if t > tf:
        return np.array([xf, xdf, xddf])
    ts = tf - t0
    a1, a2, a3 = x0, xd0, xdd0 / 2.0
    Tm = np.array([[ts**3, ts**4, ts**5], [3 * ts**2, 4 * ts**3, 5 * ts**4], [6 * ts, 12 * ts**2, 20 * ts**3]])
    Rt = np.array([xf - x0 - xd0 * ts - xdd0 * ts**2 / 2.0, xdf - xd0 - xdd0 * ts, xddf - xdd0])
    a4, a5, a6 = np.linalg.solve(Tm, Rt)
    s = t - t0
    return np.array([
        a1 + a2 * s + a3 * s**2 + a4 * s**3 + a5 * s**4 + a6 * s**5,
        a2 + 2 * a3 * s + 3 * a4 * s**2 + 4 * a5 * s**3 + 5 * a6 * s**4,
        2 * a3 + 6 * a4 * s + 12 * a5 * s**2 + 20 * a6 * s**3,
    ])


def quat_from_R(m):
    """Eigen Quaternion(Matrix3) [ext]: (x, y, z, w)"""
    t = m[0, 0] + m[1, 1] + m[2, 2]
    q = np.zeros(4)
    if t > 0.0:
        t = np.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0] = (m[2, 1] - m[1, 2]) * t
        q[1] = (m[0, 2] - m[2, 0]) * t
        q[2] = (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (m[k, j] - m[j, k]) * t
        q[j] = (m[j, i] + m[i, j]) * t
        q[k] = (m[k, i] + m[i, k]) * t
    return q


def quat_slerp(qa, t, qb):
    """Eigen QuaternionBase::slerp [ext]"""
    d = float(qa @ qb)
    ad = abs(d)
    if ad >= 1.0 - np.finfo(float).eps:
        s0, s1 = 1.0 - t, t
    else:
        th = np.arccos(ad)
        st = np.sin(th)
        s0 = np.sin((1.0 - t) * th) / st
        s1 = np.sin(t * th) / st
    if d < 0:
        s1 = -s1
    return s0 * qa + s1 * qb


def quat_to_R(q):
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
    ])


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by + ay * bw + az * bx - ax * bz,
        aw * bz + az * bw + ax * by - ay * bx,
        aw * bw - ax * bx - ay * by - az * bz,
    ])


def quat_inverse(q):
    n2 = float(q @ q)
    return np.array([-q[0], -q[1], -q[2], q[3]]) / n2


def angle_axis_from_quat(q):
    """Eigen AngleAxis(Quaternion) [ext] -> (angle, axis)"""
    n = np.linalg.norm(q[:3])
    if n != 0.0:
        ang = 2.0 * np.arctan2(n, abs(q[3]))
        if q[3] < 0:
            n = -n
        return ang, q[:3] / n
    return 0.0, np.array([1.0, 0.0, 0.0])


def get_phi(Rc, Rd):
    """src/math.cpp:275-291 (the two sign flips cancel): 1/2 sum_i (col_i(Rc) x col_i(Rd))"""
    return 0.5 * sum(np.cross(Rc[:, i], Rd[:, i]) for i in range(3))


def fstar_pos_pd(t, tr, gains, pos, vel):
    """TaskLink::GetFstarPosPD (src/task.cpp:268-293)"""
    pt, vt, at = np.zeros(3), np.zeros(3), np.zeros(3)
    for j in range(3):
        pt[j], vt[j], at[j] = quintic_spline(t, tr["t0"], tr["t1"], tr["pos_init"][j], tr["vel_init"][j], 0.0, tr["pos_des"][j], tr["vel_des"][j], 0.0)
    return gains["pos_a"] * at + gains["pos_p"] * (pt - pos) + gains["pos_d"] * (vt - vel)


def fstar_rot_pd(t, tr, gains, rot, w):
    """TaskLink::GetFstarRotPD (src/task.cpp:295-339)"""
    qs = quintic_spline(t, tr["t0"], tr["t1"], 0, 0, 0, 1, 0, 0)
    qi, qd = quat_from_R(tr["rot_init"]), quat_from_R(tr["rot_des"])
    rot_traj = quat_to_R(quat_slerp(qi, qs[0], qd))
    ang, ax = angle_axis_from_quat(quat_mul(qd, quat_inverse(qi)))
    w_traj = ang * qs[1] * ax
    return gains["rot_p"] * get_phi(rot, rot_traj) + gains["rot_d"] * (w_traj - w)


def link_fstar(mode, t, tr, gains, link, task_point, com_l, R, p, v, w, f_in):
    """the f* segment of one task link as RobotData::UpdateTaskSpace fills it (src/dwbc.cpp:708-780); f_in is the
    segment SetTaskSpace left there, kept where no trajectory is set"""
    out = np.array(f_in, float).copy()
    six = mode in (TASK_LINK_6D, TASK_LINK_6D_COM_FRAME, TASK_LINK_6D_CUSTOM_FRAME)
    posm = mode in (TASK_LINK_POSITION, TASK_LINK_POSITION_COM_FRAME, TASK_LINK_POSITION_CUSTOM_FRAME)
    if (six or posm) and tr["has_pos"]:
        if mode in (TASK_LINK_6D_COM_FRAME, TASK_LINK_POSITION_COM_FRAME):
            cp, cv = p[link] + R[link] @ com_l, v[link] + np.cross(w[link], R[link] @ com_l)  # xipos, vi (link.cpp:83,94)
        elif mode in (TASK_LINK_6D_CUSTOM_FRAME, TASK_LINK_POSITION_CUSTOM_FRAME):
            cp, cv = p[link] + R[link] @ task_point, v[link] + np.cross(w[link], task_point)  # sic: w x task_point_, not rotated (dwbc.cpp:742)
        else:
            cp, cv = p[link], v[link]
        out[0:3] = fstar_pos_pd(t, tr, gains, cp, cv)
    if (six or not posm) and tr["has_rot"]:
        o = 3 if six else 0
        out[o : o + 3] = fstar_rot_pd(t, tr, gains, R[link], w[link])
    return out


def pack_traj(tr):
    rec = np.zeros(TRAJ_STRIDE)
    rec[0], rec[1] = tr["t0"], tr["t1"]
    rec[2:5], rec[5:8], rec[8:11], rec[11:14] = tr["pos_init"], tr["vel_init"], tr["pos_des"], tr["vel_des"]
    rec[14:23] = np.asarray(tr["rot_init"]).reshape(9)
    rec[23:32] = np.asarray(tr["rot_des"]).reshape(9)
    rec[32], rec[33] = float(tr["has_pos"]), float(tr["has_rot"])
    return rec

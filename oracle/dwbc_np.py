"""numpy restatement of the libdwbc per-cycle OSF/HQP torque solve.   TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
It is the slow, readable twin of oracle/dwbc_oracle.c (same algorithm, written independently
in array form); the two are cross-checked in tests/test_oracle_*.py and both are pinned by the
reference's binary goldens tests/golden/cases/{1,2} (reference tests/cases/*, format
reference tests/dwbc_test_util.h:15-28).

Third-party arithmetic that is NOT in /root/reference and is restated from its published
algorithm [ext]:
  RBDL (saga0619/rbdl-orb, unpinned): UpdateKinematicsCustom, CompositeRigidBodyAlgorithm
      (Featherstone CRBA), CalcPointJacobian6D, CalcBodyToBaseCoordinates
  Eigen (>=3.0, unpinned): LLT solve, CompleteOrthogonalDecomposition::pseudoInverse
  qpOASES (saga0619/qpOASES, unpinned): replaced by a dual active-set QP (Goldfarb-Idnani)
      with the canonical lexicographic min-norm tie-break (DESIGN.md "QP canon").

Reference call sites followed (file:line in /root/reference):
  src/dwbc.cpp:279-371   UpdateKinematics (A_, A_inv_, com, CMM_, G_)
  src/link.cpp:76-119, src/contact_constraint.cpp:51-77   poses + point Jacobians [lin;ang]
  src/dwbc.cpp:433-478 + src/wbd.cpp:108-143   contact constraint algebra
  src/wbd.cpp:5-53       COD pseudo-inverse, V2
  src/wbd.cpp:59-97      ZMP / friction cone rows
  src/wbd.cpp:186-192    gravity compensation
  src/dwbc.cpp:685-816 + src/wbd.cpp:207-261   task Jacobians, J_kt, Lambda, null-space
  src/dwbc.cpp:818-873, 941-1127   cascade + per-level QP assembly
  src/dwbc.cpp:1372-1568 contact redistribution QP
  src/wbd.cpp:268-271    contact force
"""
import struct

import numpy as np

COD_THRESHOLD = 1.0e-6  # reference include/dwbc_wbd.h:10
GRAV = 9.81
INFTY = 1.0e20

TASK_LINK_6D = 0
TASK_LINK_6D_COM_FRAME = 1
TASK_LINK_6D_CUSTOM_FRAME = 2
TASK_LINK_POSITION = 3
TASK_LINK_POSITION_COM_FRAME = 4
TASK_LINK_POSITION_CUSTOM_FRAME = 5
TASK_LINK_ROTATION = 6
TASK_LINK_ROTATION_CUSTOM_FRAME = 7


def read_golden(path):
    """int64 rows, int64 cols, float64 column-major (reference tests/dwbc_test_util.h:15-28)."""
    with open(path, "rb") as f:
        b = f.read()
    r, c = struct.unpack("<qq", b[:16])
    a = np.frombuffer(b[16:], dtype="<f8").reshape((c, r)).T.copy()
    return a


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def quat_to_R(x, y, z, w):
    """Standard rotation matrix of unit quaternion (body -> world).  RBDL's Quaternion::toMatrix
    returns the transpose (world -> body coordinate transform E) [ext]."""
    return np.array(
        [
            [1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * x * z + 2 * w * y],
            [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
            [2 * x * z - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y],
        ]
    )


def axis_angle_R(a, q):
    """Rodrigues: rotation by q about unit axis a (child -> joint frame)."""
    K = skew(a)
    return np.eye(3) + np.sin(q) * K + (1 - np.cos(q)) * (K @ K)


# ----------------------------------------------------------------------------------------------
# kinematics + CRBA (body-coordinate Featherstone recursion, as RBDL does it)
# ----------------------------------------------------------------------------------------------
def forward_kinematics(model, q):
    """returns R[i] (body->world), p[i] (body origin in world)."""
    nb = model["nb"]
    R = np.zeros((nb, 3, 3))
    p = np.zeros((nb, 3))
    R[0] = quat_to_R(q[3], q[4], q[5], q[model["ndof"]])
    p[0] = q[0:3]
    for i in range(1, nb):
        par = model["parent"][i]
        Rj = axis_angle_R(model["axis"][i], q[6 + i - 1])
        R[i] = R[par] @ model["R_T"][i] @ Rj
        p[i] = p[par] + R[par] @ model["p_T"][i]
    return R, p


def _spatial_inertia(m, c, I):
    """6x6 spatial inertia about the body origin, [ang; lin] ordering (RBDL convention)."""
    C = skew(c)
    out = np.zeros((6, 6))
    out[:3, :3] = I + m * (C @ C.T)
    out[:3, 3:] = m * C
    out[3:, :3] = m * C.T
    out[3:, 3:] = m * np.eye(3)
    return out


def _X(E, r):
    """Plücker motion transform for frame B given in A by rotation E (A->B coords) and origin r (in A)."""
    X = np.zeros((6, 6))
    X[:3, :3] = E
    X[3:, 3:] = E
    X[3:, :3] = -E @ skew(r)
    return X


def crba(model, q):
    """Composite rigid body algorithm in body coordinates -> joint-space inertia A (ndof x ndof).
    DoF order: 0-2 base translation (world axes), 3-5 base angular velocity (BODY frame),
    6.. revolute joints in body order.  [ext: RBDL CompositeRigidBodyAlgorithm]"""
    nb, n = model["nb"], model["ndof"]
    par = model["parent"]
    Rb = quat_to_R(q[3], q[4], q[5], q[n])
    Xl = [None] * nb  # parent -> body motion transform
    S = [None] * nb
    Ic = [None] * nb
    for i in range(nb):
        Ic[i] = _spatial_inertia(model["mass"][i], model["com"][i], model["inertia"][i])
        if i == 0:
            continue
        Rj = axis_angle_R(model["axis"][i], q[6 + i - 1])
        E = (model["R_T"][i] @ Rj).T
        Xl[i] = _X(E, model["p_T"][i])
        S[i] = np.concatenate([model["axis"][i], np.zeros(3)])
    for i in range(nb - 1, 0, -1):
        Ic[par[i]] = Ic[par[i]] + Xl[i].T @ Ic[i] @ Xl[i]
    A = np.zeros((n, n))
    # base: spherical joint (body-frame omega) on body 0, preceded by world-axis translation.
    # body-0 frame motion subspace: [omega_body ; v_body]; translation dofs move the origin with
    # world velocity e_k -> body coords R^T e_k.
    S0 = np.zeros((6, 6))
    S0[3:, 0:3] = Rb.T  # translation columns (linear part, body coords)
    S0[:3, 3:6] = np.eye(3)  # spherical columns
    A[:6, :6] = S0.T @ Ic[0] @ S0
    for i in range(1, nb):
        di = 6 + i - 1
        F = Ic[i] @ S[i]
        A[di, di] = S[i] @ F
        j = i
        while par[j] > 0:
            F = Xl[j].T @ F
            j = par[j]
            dj = 6 + j - 1
            A[di, dj] = A[dj, di] = F @ S[j]
        F = Xl[j].T @ F  # now in body-0 coordinates
        A[di, :6] = A[:6, di] = S0.T @ F
    return A


def point_jacobian(model, R, p, body, point_local):
    """6 x ndof world-frame Jacobian of a body-fixed point, rows [linear; angular]
    (reference src/link.cpp:98-119, src/contact_constraint.cpp:59-61 swap RBDL's [ang;lin])."""
    n = model["ndof"]
    par = model["parent"]
    P = p[body] + R[body] @ point_local
    J = np.zeros((6, n))
    J[0:3, 0:3] = np.eye(3)
    for k in range(3):
        w = R[0][:, k]
        J[0:3, 3 + k] = np.cross(w, P - p[0])
        J[3:6, 3 + k] = w
    j = body
    while j > 0:
        w = R[j] @ model["axis"][j]
        J[0:3, 6 + j - 1] = np.cross(w, P - p[j])
        J[3:6, 6 + j - 1] = w
        j = par[j]
    return J


# ----------------------------------------------------------------------------------------------
# dense helpers that stand in for Eigen
# ----------------------------------------------------------------------------------------------
def llt_inverse(A):
    L = np.linalg.cholesky(A)
    Li = np.linalg.solve(L, np.eye(A.shape[0]))
    return Li.T @ Li


def pinv_cod(M, want_v2=False, thr=COD_THRESHOLD):
    """Moore-Penrose pseudo-inverse by column-pivoted Householder QR + complete orthogonal
    decomposition; rank = #{|R_ii| > thr * |R_00|} (Eigen COD::setThreshold semantics [ext]);
    V2 = rows [rank:] of Q^T (reference src/wbd.cpp:32-53)."""
    import scipy.linalg as sla

    Q, Rm, piv = sla.qr(M, pivoting=True)
    d = np.abs(np.diag(Rm))
    rank = int(np.sum(d > thr * d[0])) if d[0] > 0 else 0
    n = M.shape[1]
    R1 = Rm[:rank, :]
    # min-norm solution operator of R1 (rank x n, full row rank): R1^T (R1 R1^T)^-1 via QR of R1^T
    Qz, Tz = np.linalg.qr(R1.T)  # R1^T = Qz Tz  -> R1 = Tz^T Qz^T
    R1p = Qz @ np.linalg.solve(Tz.T, np.eye(rank)) if rank > 0 else np.zeros((n, 0))
    P = np.zeros((n, n))
    P[piv, np.arange(n)] = 1.0
    pinv = P @ R1p @ Q[:, :rank].T
    if want_v2:
        return pinv, Q.T[rank:, :], rank
    return pinv


def zmp_const(lx, ly):
    Z = np.zeros((4, 6))
    Z[0, 2], Z[0, 4] = -lx, -1
    Z[1, 2], Z[1, 4] = -lx, 1
    Z[2, 2], Z[2, 3] = -ly, -1
    Z[3, 2], Z[3, 3] = -ly, 1
    return Z


def force_const(mu, muz):
    F = np.zeros((6, 6))
    F[0, 0], F[0, 2] = 1, -mu
    F[1, 0], F[1, 2] = -1, -mu
    F[2, 1], F[2, 2] = 1, -mu
    F[3, 1], F[3, 2] = -1, -mu
    F[4, 5], F[4, 2] = 1, -muz
    F[5, 5], F[5, 2] = -1, -muz
    return F


# ----------------------------------------------------------------------------------------------
# QP:  lexmin ( 1/2|x[:t]|^2 , 1/2|x[t:]|^2 )  s.t.  A x <= ub     (H = diag(I_t, 0_k), g = 0)
# ----------------------------------------------------------------------------------------------
QP_SCALE = 1.0e4  # c = QP_SCALE * c_hat ; eps = 1/QP_SCALE^2 Tikhonov weight used to FIND the active set
QP_TOL = 1.0e-9
QP_ZERO_ROW = 1.0e-9  # rows with a smaller norm are treated as 0 . x <= b (their slack is taken unnormalised)


def _gi_least_distance(G, b, max_iter, tol=1.0e-9):
    """Goldfarb-Idnani dual active set for  min 1/2|x|^2  s.t. G x <= b.
    Returns (status, x, active list, multipliers).  Written for clarity: every projection is
    recomputed from scratch with lstsq (the HIP kernel updates a QR instead)."""
    m, n = G.shape
    x = np.zeros(n)
    act = []
    u = np.zeros(0)
    gnorm = np.linalg.norm(G, axis=1)
    gnorm[gnorm < QP_ZERO_ROW] = 1.0  # a numerically zero row is the constraint 0 <= b: never normalised by its noise
    it = 0
    while True:
        s = b - G @ x
        viol = s / gnorm
        viol[act] = 0.0
        p = int(np.argmin(viol))
        if viol[p] >= -tol:
            return 1, x, act, u
        up = 0.0
        while True:
            it += 1
            if it > max_iter:
                return 0, x, act, u
            gp = -G[p]  # GI is stated for n^T x >= b': n = -g
            if act:
                N = -G[act].T
                r = np.linalg.lstsq(N, gp, rcond=None)[0]
                z = gp - N @ r
            else:
                r = np.zeros(0)
                z = gp.copy()
            zn = np.linalg.norm(z)
            t1, l = np.inf, -1
            for j in range(len(act)):
                if r[j] > 1e-13 * max(1.0, np.abs(r).max()):
                    tj = u[j] / r[j]
                    if tj < t1:
                        t1, l = tj, j
            sp = b[p] - G[p] @ x
            if zn > 1e-10 * gnorm[p]:
                t2 = -sp / (z @ gp)
            else:
                t2 = np.inf
            t = min(t1, t2)
            if not np.isfinite(t):
                return 0, x, act, u
            if not np.isfinite(t2):
                u = u - t * r
                up += t
                u = np.delete(u, l)
                act.pop(l)
                continue
            x = x + t * z
            u = u - t * r
            up += t
            if t2 <= t1:
                act.append(p)
                u = np.append(u, up)
                break
            u = np.delete(u, l)
            act.pop(l)


def _lex_eqp(Ad, Ac, b):
    """Exact lexicographic equality-constrained solve on a working set:
       stage 1  min |d|^2  s.t. exists c: Ad d + Ac c = b ;  stage 2  min |c|^2 on that set."""
    import scipy.linalg as sla

    q, k = Ac.shape
    t = Ad.shape[1]
    if q == 0:
        return np.zeros(t), np.zeros(k)
    if k == 0:
        d = Ad.T @ np.linalg.solve(Ad @ Ad.T, b)
        return d, np.zeros(0)
    Q, Rm, piv = sla.qr(Ac, pivoting=True)  # Ac[:,piv] = Q Rm
    dg = np.abs(np.diag(Rm))
    rho = int(np.sum(dg > 1e-9 * dg[0])) if dg.size and dg[0] > 0 else 0
    Ab = Q.T @ Ad
    bb = Q.T @ b
    if q - rho > 0:
        A2 = Ab[rho:]
        d = A2.T @ np.linalg.solve(A2 @ A2.T, bb[rho:])
    else:
        d = np.zeros(t)
    c = np.zeros(k)
    if rho > 0:
        R1 = Rm[:rho, :]
        f = bb[:rho] - Ab[:rho] @ d
        cp = R1.T @ np.linalg.solve(R1 @ R1.T, f)
        c[piv] = cp
    return d, c


QP_FEAS_TOL = 1.0e-7  # acceptance of the lexicographic point: slack / |row| >= -QP_FEAS_TOL on every row


def _worst_slack(A, ub, x):
    nrm = np.linalg.norm(A, axis=1)
    nrm[nrm < QP_ZERO_ROW] = 1.0
    return float(((ub - A @ x) / nrm).min()) if A.shape[0] else 0.0


def solve_qp(A, ub, t, max_iter=1000, tol=None):
    """x = [d (t); c (k)].  Returns (status, x, active_rows).   (DESIGN.md "QP canon")
      1. working set W := active set of  min 1/2|d|^2 + 1/2 eps |c|^2  (eps = QP_SCALE^-2), Goldfarb-Idnani;
      2. x := lexicographic least-norm point on W (min |d| first, then min |c|) if it is feasible to QP_FEAS_TOL,
      3. otherwise x := the Tikhonov point on W (near-singular contact blocks make the lexicographic point blow up)."""
    m, nv = A.shape
    k = nv - t
    G = A.copy()
    G[:, t:] *= QP_SCALE
    # tol: violation below which a row counts as satisfied during the search: QP_TOL, except for the contact redistribution QP, which
    # starts from a point the last task QP ACCEPTED at QP_FEAS_TOL and searches with that tolerance (canon rule 5)
    st, xh, act, u = _gi_least_distance(G, ub, max_iter, QP_TOL if tol is None else tol)
    if st == 0:
        return 0, np.zeros(nv), []
    act_sorted = list(act)

    def tikhonov_point():
        if not act_sorted:
            return np.zeros(nv)
        N = G[act_sorted]
        xs = N.T @ np.linalg.solve(N @ N.T, ub[act_sorted])
        xs = xs.copy()
        xs[t:] *= QP_SCALE
        return xs

    if k > 0 and t > 0:
        d, c = _lex_eqp(A[act_sorted, :t], A[act_sorted, t:], ub[act_sorted])
        x = np.concatenate([d, c])
        if _worst_slack(A, ub, x) < -QP_FEAS_TOL:
            x = tikhonov_point()
    else:
        x = tikhonov_point()  # strictly convex: re-solve on the final working set to shed drift
    return 1, x, sorted(act_sorted)


# ----------------------------------------------------------------------------------------------
# the control cycle
# ----------------------------------------------------------------------------------------------
class Cycle:
    """One robot instance; mirrors the call sequence of reference tests/dwbc_test.cpp:61-130."""

    def __init__(self, model):
        self.model = model
        self.n = model["ndof"]
        self.m = self.n - 6
        self.contacts = []  # dict(link, point, lx, ly, mu, muz)
        self.tasks = []  # list of levels; each level = list of (mode, link, point)
        self.tau_lim = None

    def add_contact(self, link, point, lx, ly, mu=0.2, muz=0.2):
        self.contacts.append(dict(link=link, point=np.asarray(point, float), lx=lx, ly=ly, mu=mu, muz=muz))

    def add_task(self, level, mode, link, point=(0, 0, 0)):
        while len(self.tasks) <= level:
            self.tasks.append([])
        self.tasks[level].append((mode, link, np.asarray(point, float)))

    def add_custom_task(self, level, dof):
        """AddTaskSpace(heirarchy, TASK_CUSTOM, task_dof) (dwbc.cpp:522-530); the Jacobian comes from set_custom_J"""
        while len(self.tasks) <= level:
            self.tasks.append([])
        self.tasks[level] = ("custom", dof)
        self.custom_J = getattr(self, "custom_J", {})

    def set_custom_J(self, level, J):
        """SetTaskSpace(heirarchy, f*, J_task) (dwbc.cpp:664-681)"""
        self.custom_J[level] = np.asarray(J, float)

    def set_torque_limit(self, lim):
        self.tau_lim = np.asarray(lim, float)

    # -- UpdateKinematics (dwbc.cpp:279-371)
    def update_kinematics(self, q):
        mdl = self.model
        self.q = np.asarray(q, float)
        self.R, self.p = forward_kinematics(mdl, self.q)
        self.A = crba(mdl, self.q)
        self.A_inv = llt_inverse(self.A)
        mtot = mdl["mass"].sum()
        skm = self.R[0] @ self.A[3:6, 0:3] / mtot
        self.com_from_pelv = np.array([skm[2, 1], skm[0, 2], skm[1, 0]])
        self.com = self.com_from_pelv + self.q[0:3]
        cm = np.eye(6)
        cm[3:, 3:] = self.R[0]
        cm[3:, :3] = skew(self.com_from_pelv).T
        self.CMM = cm @ self.A[:6, :]
        Icom = self.R[0] @ self.A[3:6, 3:6] @ self.R[0].T - mtot * skew(self.com_from_pelv) @ skew(self.com_from_pelv).T
        SI = np.zeros((6, 6))
        SI[:3, :3] = np.eye(3) * mtot
        SI[3:, 3:] = Icom
        self.J_com = np.linalg.solve(SI, self.CMM)
        self.G = -self.J_com[:3].T @ (mtot * np.array([0, 0, -GRAV]))

    # -- SetContact / UpdateContactConstraint (dwbc.h:432-474, dwbc.cpp:433-454)
    def set_contact(self, flags):
        flags = list(flags) + [False] * (len(self.contacts) - len(flags))
        self.cflags = flags
        rows = []
        self.c_rot = []
        for cc, f in zip(self.contacts, flags):
            J = point_jacobian(self.model, self.R, self.p, cc["link"], cc["point"])
            if f:
                rows.append(J)
                self.c_rot.append(self.R[cc["link"]])
        self.J_C = np.vstack(rows) if rows else np.zeros((0, self.n))
        self.cdof = self.J_C.shape[0]
        self.act_contacts = [cc for cc, f in zip(self.contacts, flags) if f]

    # -- CalcContactConstraint (wbd.cpp:108-143)
    def calc_contact_constraint(self):
        J, Ai = self.J_C, self.A_inv
        n = self.n
        self.Lambda_c = np.linalg.inv(J @ Ai @ J.T)
        self.J_C_INV_T = self.Lambda_c @ J @ Ai
        self.N_C = np.eye(n) - J.T @ self.J_C_INV_T
        self.A_inv_N_C = Ai @ self.N_C
        self.W = self.A_inv_N_C[6:, 6:]
        if self.cdof > 6:
            self.W_inv, self.V2, rank = pinv_cod(self.W, want_v2=True)
            ok = self.V2.shape[0] == self.cdof - 6
            self.NwJw = self.V2.T @ np.linalg.inv(self.J_C_INV_T[: self.cdof - 6, 6:] @ self.V2.T)
            return 1 if ok else 0
        self.W_inv = pinv_cod(self.W)
        self.V2 = np.zeros((0, self.m))
        self.NwJw = np.zeros((self.m, 0))
        return 1

    # -- CalcGravCompensation (wbd.cpp:186-192)
    def calc_grav(self):
        self.tau_grav = self.W_inv @ (self.A_inv[6:, :] @ (self.N_C @ self.G))
        self.P_C = self.J_C_INV_T @ self.G
        return self.tau_grav

    # -- UpdateTaskSpace (dwbc.cpp:685-793)
    def task_jacobian(self, level):
        if isinstance(self.tasks[level], tuple) and self.tasks[level][0] == "custom":
            return self.custom_J[level]
        rows = []
        for mode, link, pt in self.tasks[level]:
            if link == self.model["nb"]:  # the synthetic "COM" link: jac_ = jac_com_ = SI_body^-1 CMM_ (dwbc.cpp:230-231,352-353)
                J = self.J_com
                rows.append(J if mode <= TASK_LINK_6D_CUSTOM_FRAME else (J[:3] if mode <= TASK_LINK_POSITION_CUSTOM_FRAME else J[3:]))
                continue
            com_l = self.model["com"][link]
            if mode in (TASK_LINK_6D, TASK_LINK_POSITION, TASK_LINK_ROTATION, TASK_LINK_ROTATION_CUSTOM_FRAME):
                J = point_jacobian(self.model, self.R, self.p, link, np.zeros(3))
            elif mode in (TASK_LINK_6D_COM_FRAME, TASK_LINK_POSITION_COM_FRAME):
                J = point_jacobian(self.model, self.R, self.p, link, com_l)
            else:
                J = point_jacobian(self.model, self.R, self.p, link, pt)
            if mode in (TASK_LINK_6D, TASK_LINK_6D_COM_FRAME, TASK_LINK_6D_CUSTOM_FRAME):
                rows.append(J)
            elif mode in (TASK_LINK_POSITION, TASK_LINK_POSITION_COM_FRAME, TASK_LINK_POSITION_CUSTOM_FRAME):
                rows.append(J[:3])
            else:
                rows.append(J[3:])
        return np.vstack(rows)

    def cone_matrix(self):
        """C~ = A_const_a * A_rot  (dwbc.cpp:1018-1039)."""
        nc = len(self.act_contacts)
        Ca = np.zeros((10 * nc, 6 * nc))
        Ar = np.zeros((6 * nc, 6 * nc))
        for i, cc in enumerate(self.act_contacts):
            Rt = self.c_rot[i].T
            Ar[6 * i : 6 * i + 3, 6 * i : 6 * i + 3] = Rt
            Ar[6 * i + 3 : 6 * i + 6, 6 * i + 3 : 6 * i + 6] = Rt
            Ca[10 * i : 10 * i + 4, 6 * i : 6 * i + 6] = zmp_const(cc["lx"], cc["ly"])
            Ca[10 * i + 4 : 10 * i + 10, 6 * i : 6 * i + 6] = force_const(cc["mu"], cc["muz"])
        return Ca @ Ar

    # -- per-level QP assembly (dwbc.cpp:941-1053)
    def build_task_qp(self, Ntorque_task, fstar, tau_prev):
        m = self.m
        t = Ntorque_task.shape[1]
        k = max(self.cdof - 6, 0)
        nlim = 2 * m if self.tau_lim is not None else 0
        ncone = 10 * len(self.act_contacts)
        A = np.zeros((nlim + ncone, t + k))
        ub = np.zeros(nlim + ncone)
        base = tau_prev + Ntorque_task @ fstar
        if nlim:
            A[:m, :t] = Ntorque_task
            A[m : 2 * m, :t] = -Ntorque_task
            if k:
                A[:m, t:] = self.NwJw
                A[m : 2 * m, t:] = -self.NwJw
            ub[:m] = self.tau_lim - base
            ub[m : 2 * m] = self.tau_lim + base
        if ncone:
            Ct = self.cone_matrix()
            Atemp = Ct @ self.J_C_INV_T[:, 6:]
            A[nlim:, :t] = -Atemp @ Ntorque_task
            if k:
                A[nlim:, t:] = -Atemp @ self.NwJw
            bA = Ct @ self.P_C - Atemp @ base
            ub[nlim:] = -bA
        return A, ub

    # -- CalcTaskSpace + CalcTaskControlTorque(hqp=true) (dwbc.cpp:795-873, wbd.cpp:207-261)
    def calc_task_torque(self, fstars):
        m = self.m
        k = max(self.cdof - 6, 0)
        L = len(self.tasks)
        self.J_task, self.Lambda_t, self.J_kt, self.Null, self.qp = [], [], [], [], []
        AiNc = self.A_inv @ self.N_C  # wbd.cpp:210 recomputes it
        for i in range(L):
            Jt = self.task_jacobian(i)
            lam = np.linalg.inv(Jt @ AiNc @ Jt.T)
            Q = (lam @ Jt @ AiNc)[:, 6:]
            Jkt = self.W_inv @ Q.T @ pinv_cod(Q @ self.W_inv @ Q.T)
            self.J_task.append(Jt)
            self.Lambda_t.append(lam)
            self.J_kt.append(Jkt)
            if i != L - 1:
                prev = np.eye(m) if i == 0 else self.Null[i - 1]
                self.Null.append(prev @ (np.eye(m) - Jkt @ lam @ Jt @ self.A_inv_N_C[:, 6:]))
        self.tau_task = np.zeros(m)
        self.tau_contact = np.zeros(m)
        self.fstar_qp, self.contact_qp, self.qp_active = [], [], []
        for i in range(L):
            Nprev = np.eye(m) if i == 0 else self.Null[i - 1]
            Nt = Nprev @ self.J_kt[i] @ self.Lambda_t[i]
            tau_prev = self.tau_grav + self.tau_task
            A, ub = self.build_task_qp(Nt, fstars[i], tau_prev)
            t = Nt.shape[1]
            st, x, act = solve_qp(A, ub, t, 1000)
            self.qp.append((A, ub))
            if st == 0:
                self.fstar_qp.append(np.zeros(t))
                self.contact_qp.append(np.zeros(k))
                return 0
            self.fstar_qp.append(x[:t])
            self.contact_qp.append(x[t:])
            self.qp_active.append(act)
            torque_h = self.J_kt[i] @ self.Lambda_t[i] @ (fstars[i] + x[:t])
            self.tau_task = self.tau_task + Nprev @ torque_h
            self.tau_contact = self.NwJw @ x[t:] if k else np.zeros(m)
        return 1

    # -- CalcContactRedistribute(hqp=true) (dwbc.cpp:1372-1568)
    def calc_contact_redistribute(self):
        m = self.m
        k = max(self.cdof - 6, 0)
        if k == 0:
            self.tau_contact = np.zeros(m)
            return 1
        tau_in = self.tau_grav + self.tau_task + self.tau_contact
        nlim = 2 * m if self.tau_lim is not None else 0
        ncone = 10 * len(self.act_contacts)
        A = np.zeros((nlim + ncone, k))
        ub = np.zeros(nlim + ncone)
        if nlim:
            A[:m] = self.NwJw
            A[m : 2 * m] = -self.NwJw
            ub[:m] = self.tau_lim - tau_in
            ub[m : 2 * m] = self.tau_lim + tau_in
        CM = -self.cone_matrix()
        A[nlim:] = CM @ self.J_C_INV_T[:, 6:] @ self.NwJw
        ub[nlim:] = CM @ self.P_C - CM @ self.J_C_INV_T[:, 6:] @ tau_in
        st, x, act = solve_qp(A, ub, k, 300, tol=QP_FEAS_TOL)  # H = I over all k variables: t := k
        self.redis_qp = (A, ub)
        if st == 0:
            self.tau_contact = np.zeros(m)
            return 0
        self.cf_redis = x
        self.tau_contact = self.tau_contact + self.NwJw @ x
        return 1

    def contact_force(self, tau):
        return self.J_C_INV_T[:, 6:] @ tau - self.P_C

    def get_zmp(self, cf):
        """RobotData::getZMP (dwbc.cpp:898-939) on the packed wrench of the active contacts -> (zmp, per-contact zmp_pos)"""
        pts = [self.p[cc["link"]] + self.R[cc["link"]] @ cc["point"] for cc in self.act_contacts]
        tot = sum(cf[6 * i + 2] for i in range(len(pts)))
        z = np.zeros(3)
        zs = []
        for i, xc in enumerate(pts):
            fz = cf[6 * i + 2]
            zp = xc.copy()
            if not fz > -1.0e-3:
                zp[0] += -cf[6 * i + 4] / fz
                zp[1] += cf[6 * i + 3] / fz
            zs.append(zp)
            z = z + zp * fz / tot
        return z, zs

    def run(self, q, flags, fstars):
        self.update_kinematics(q)
        self.set_contact(flags)
        ok = self.calc_contact_constraint()
        self.calc_grav()
        ok_t = self.calc_task_torque(fstars)
        ok_c = self.calc_contact_redistribute()
        self.status = int(ok and ok_t and ok_c)
        return self.tau_grav + self.tau_task + self.tau_contact


# ----------------------------------------------------------------------------------------------
# velocity-dependent outputs of UpdateKinematics (reference src/dwbc.cpp:340-344,362-367, src/link.cpp:76-96)
# ----------------------------------------------------------------------------------------------
def nonlinear_effects(model, q, qd, gravity=GRAV):
    """B_ = C(q, qd) qd + g(q): recursive Newton-Euler with zero joint acceleration in body coordinates
    ([ext] RBDL NonlinearEffects; spatial vectors [ang; lin]).  DoF conventions as crba()."""
    nb, n = model["nb"], model["ndof"]
    par = model["parent"]
    Rb = quat_to_R(q[3], q[4], q[5], q[n])

    def crm(v):  # spatial motion cross product matrix
        out = np.zeros((6, 6))
        out[:3, :3] = skew(v[:3])
        out[3:, :3] = skew(v[3:])
        out[3:, 3:] = skew(v[:3])
        return out

    I = [_spatial_inertia(model["mass"][i], model["com"][i], model["inertia"][i]) for i in range(nb)]
    Xl, S, v, a, f = [None] * nb, [None] * nb, [None] * nb, [None] * nb, [None] * nb
    # base: world-axis translation then body-frame spherical joint.  In body-0 coordinates the velocity is
    # [omega_b ; R^T v_world]; the spatial acceleration with zero generalised acceleration is [0 ; -omega_b x (R^T v)]
    # plus the fictitious upward acceleration that stands for gravity.
    wb = np.asarray(qd[3:6], float)
    vb = Rb.T @ np.asarray(qd[0:3], float)
    v[0] = np.concatenate([wb, vb])
    a[0] = np.concatenate([np.zeros(3), -np.cross(wb, vb) + Rb.T @ np.array([0, 0, gravity])])
    for i in range(1, nb):
        Rj = axis_angle_R(model["axis"][i], q[6 + i - 1])
        Xl[i] = _X((model["R_T"][i] @ Rj).T, model["p_T"][i])
        S[i] = np.concatenate([model["axis"][i], np.zeros(3)])
        vj = S[i] * qd[6 + i - 1]
        v[i] = Xl[i] @ v[par[i]] + vj
        a[i] = Xl[i] @ a[par[i]] + crm(v[i]) @ vj
    for i in range(nb):
        f[i] = I[i] @ a[i] - crm(v[i]).T @ (I[i] @ v[i])
    tau = np.zeros(n)
    for i in range(nb - 1, 0, -1):
        tau[6 + i - 1] = S[i] @ f[i]
        f[par[i]] = f[par[i]] + Xl[i].T @ f[i]
    tau[0:3] = Rb @ f[0][3:]  # world-axis translation dofs
    tau[3:6] = f[0][:3]       # body-frame spherical dofs
    return tau


def link_velocities(model, R, p, qd):
    """link_[i].v / .w / .vi (reference src/link.cpp:85-95): 6-D velocity of each link origin and the linear
    velocity of its centre of mass, world frame, from the point Jacobians."""
    nb = model["nb"]
    v, w, vi = np.zeros((nb, 3)), np.zeros((nb, 3)), np.zeros((nb, 3))
    for i in range(nb):
        J = point_jacobian(model, R, p, i, np.zeros(3))
        v[i] = J[:3] @ qd
        w[i] = J[3:] @ qd
        vi[i] = v[i] + np.cross(w[i], R[i] @ model["com"][i])
    return v, w, vi


# ----------------------------------------------------------------------------------------------
# hqp = false: plain hierarchy + closed-form two-contact redistribution (reference src/dwbc.cpp:856-873, 1570-1619,
# src/wbd.cpp:273-404).  PARITY UNPINNED (no fixture; link_[0].rpy comes from Eigen's eulerAngles [ext]).
# ----------------------------------------------------------------------------------------------
def euler_angles_zyx(m):
    """Eigen MatrixBase::eulerAngles(2, 1, 0) (Eigen >= 3.3 [ext]): (a0 about Z in [0, pi], a1 about Y, a2 about X)"""
    r0 = np.arctan2(m[1, 0], m[0, 0])
    c2 = np.hypot(m[2, 2], m[2, 1])
    if r0 < 0.0:
        r0 += np.pi
        r1 = np.arctan2(-m[2, 0], -c2)
    else:
        r1 = np.arctan2(-m[2, 0], c2)
    s1, c1 = np.sin(r0), np.cos(r0)
    r2 = np.arctan2(s1 * m[0, 2] - c1 * m[1, 2], c1 * m[1, 1] - s1 * m[0, 1])
    return np.array([r0, r1, r2])


def rotate_with_z(a):
    return np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])


def contact_redistribute_two_mod(eta_cust, footlength, footwidth, mu_s, ratio_x, ratio_y, P1, P2, F12):
    """ContactRedistributetwomod (src/wbd.cpp:273-404) -> (ForceRedistribution (12), eta)"""
    W = np.zeros((6, 12))
    W[:, :6] = np.eye(6)
    W[:, 6:] = np.eye(6)
    W[3:, 0:3] = skew(P1)
    W[3:, 6:9] = skew(P2)
    Rf = W @ F12
    lb, ub = 1.0 - eta_cust, eta_cust

    def bound(A, B, C, lb, ub):
        a, b, c = A * A, 2.0 * A * B, B * B - C * C
        with np.errstate(all="ignore"):
            disc = np.sqrt(b * b - 4.0 * a * c)
            s1, s2 = (-b + disc) / 2.0 / a, (-b - disc) / 2.0 / a
        hi, lo = (s1, s2) if s1 > s2 else (s2, s1)
        if hi < ub:
            ub = hi
        if lo > lb:
            lb = lo
        return lb, ub

    d = P1 - P2
    lb, ub = bound(d[2] * Rf[1] - d[1] * Rf[2], Rf[3] + P2[2] * Rf[1] - P2[1] * Rf[2], ratio_y * footwidth / 2.0 * abs(Rf[2]), lb, ub)
    lb, ub = bound(-d[2] * Rf[0] + d[0] * Rf[2], Rf[4] - P2[2] * Rf[0] + P2[0] * Rf[2], ratio_x * footlength / 2.0 * abs(Rf[2]), lb, ub)
    lb, ub = bound(-d[0] * Rf[1] + d[1] * Rf[0], Rf[5] + P2[1] * Rf[0] - P2[0] * Rf[1], mu_s * abs(Rf[2]), lb, ub)
    with np.errstate(all="ignore"):
        eta_s = (-Rf[3] - P2[2] * Rf[1] + P2[1] * Rf[2]) / (d[2] * Rf[1] - d[1] * Rf[2])
    eta = eta_s
    if eta_s > ub:
        eta = ub
    elif eta_s < lb:
        eta = lb
    if (eta > eta_cust) or (eta < 1.0 - eta_cust) or not np.isfinite(eta):
        eta = 0.5
    A3, B3 = d[2] * Rf[1] - d[1] * Rf[2], Rf[3] + P2[2] * Rf[1] - P2[1] * Rf[2]
    A4, B4 = -d[2] * Rf[0] + d[0] * Rf[2], Rf[4] - P2[2] * Rf[0] + P2[0] * Rf[2]
    A5, B5 = -d[0] * Rf[1] + d[1] * Rf[0], Rf[5] + P2[1] * Rf[0] - P2[0] * Rf[1]
    out = np.zeros(12)
    out[0:3] = eta * Rf[0:3]
    out[3], out[4], out[5] = A3 * eta * eta + B3 * eta, A4 * eta * eta + B4 * eta, A5 * eta * eta + B5 * eta
    out[6:9] = (1.0 - eta) * Rf[0:3]
    out[9], out[10], out[11] = (1.0 - eta) * (A3 * eta + B3), (1.0 - eta) * (A4 * eta + B4), (1.0 - eta) * (A5 * eta + B5)
    return out, eta


def run_no_hqp(cyc, q, flags, fstars):
    """UpdateKinematics .. CalcTaskControlTorque(false) .. CalcContactRedistribute(false) on a Cycle object"""
    cyc.update_kinematics(q)
    cyc.set_contact(flags)
    ok = cyc.calc_contact_constraint()
    cyc.calc_grav()
    m = cyc.m
    L = len(cyc.tasks)
    AiNc = cyc.A_inv @ cyc.N_C
    cyc.tau_task = np.zeros(m)
    null_prev = np.eye(m)
    for i in range(L):  # CalcTaskSpace + the hqp = false branch of CalcTaskControlTorque (dwbc.cpp:856-873)
        Jt = cyc.task_jacobian(i)
        lam = np.linalg.inv(Jt @ AiNc @ Jt.T)
        Q = (lam @ Jt @ AiNc)[:, 6:]
        Jkt = cyc.W_inv @ Q.T @ pinv_cod(Q @ cyc.W_inv @ Q.T)
        cyc.tau_task = cyc.tau_task + null_prev @ (Jkt @ lam @ np.asarray(fstars[i], float))
        null_prev = null_prev @ (np.eye(m) - Jkt @ lam @ Jt @ cyc.A_inv_N_C[:, 6:])
    cyc.tau_contact = np.zeros(m)
    if cyc.cdof != 12:  # dwbc.cpp:1612-1617
        cyc.status = 0
        return cyc.tau_grav + cyc.tau_task
    tau_in = cyc.tau_grav + cyc.tau_task
    cf = cyc.J_C_INV_T[:, 6:] @ tau_in - cyc.P_C
    xc = [cyc.p[cc["link"]] + cyc.R[cc["link"]] @ cc["point"] for cc in cyc.act_contacts]
    Ry = rotate_with_z(-euler_angles_zyx(cyc.R[0])[2])  # sic: rpy(2) of eulerAngles(2,1,0) is the angle about X (dwbc.cpp:1580)
    F12 = np.concatenate([Ry @ cf[0:3], Ry @ cf[3:6], Ry @ cf[6:9], Ry @ cf[9:12]])
    red, eta = contact_redistribute_two_mod(0.99, 0.26, 0.1, 1.0, 0.9, 0.9, Ry @ (xc[0] - cyc.com), Ry @ (xc[1] - cyc.com), F12)
    fc = np.concatenate([Ry.T @ red[0:3], Ry.T @ red[3:6], Ry.T @ red[6:9], Ry.T @ red[9:12]])
    desired = -cf[6:12] + fc[6:12]
    V2t = cyc.V2.T
    cyc.tau_contact = V2t @ np.linalg.solve(cyc.J_C_INV_T[6:12, 6:] @ V2t, desired)
    cyc.eta = eta
    cyc.status = int(ok)
    return cyc.tau_grav + cyc.tau_task + cyc.tau_contact

"""TEST INFRASTRUCTURE (oracle): numpy restatement of the reference's generic hierarchical-QP class and of its LQP
configurator -- SURVEY 8 rows a16 / f3.  Only tests/ may import this.

Follows, statement by statement where the reference is deterministic:
    HQP_Hierarch::initialize / update* / normalizeConstraintMatrix   src/dwbc_hqp.cpp:436-581
    HQP::addHierarchy / prepare (null-space chain)                    src/dwbc_hqp.cpp:425-434, 23-85
    getNullSpace                                                      src/math.cpp:349-360
    HQP::solvefirst / solveSequentialSingle / solveSequential         src/dwbc_hqp.cpp:222-403
    RobotData::ConfigureLQP / CalcControlTorqueLQP                    src/dwbc.cpp:4304-4430, 4432-4452
    torque from the LQP answer                                        tests/sp_test/jacc_compare.cpp:416-418

PARITY UNPINNED.  The reference solves every level with OSQP (OsqpEigen, an un-vendored, unpopulated submodule:
.gitmodules:1-3; src/dwbc_hqp.cpp:583-631) at its default ADMM tolerances on a sparsified copy of the matrices
(`sparseView(1e-5)`), and no test or fixture in the reference asserts any number of this path (herzog_test / jacc_compare
only print).  What is restated here is the PROBLEM each level poses; it is solved exactly (dual active set), with one
stated canon where the problem itself is not unique:

    level i:   min_{u, v}  1/2 |B_i Z u + (B_i y_prev + b_i)|^2  [+ 1/2 u^T Z^T H Z u + (Z^T H y_prev)^T u]  + 1/2 |v|^2
               s.t.        A_i Z u - v <= -(A_i y_prev) - a_i                       (own inequalities, slack v)
                           A_j Z u     <= -(A_j y_prev) + v_ans_j - a_j   (j < i)   (earlier levels, slacks frozen)
               y_i = y_prev + Z u,   Z = Z_{i-1}

  * The bound v >= 0 that solveSequentialSingle writes into qp_lb_ is never handed to OSQP (solveOSQP passes only qp_A_,
    qp_lbA_ = -inf, qp_ubA_), so v is free; at the optimum v = max(0, A Z u - ub) either way.
  * The Hessian in u is only positive SEMI-definite in general (TOCABI LQP: the six internal-wrench directions of the two
    feet are in every null space and carry no cost), so u is not unique and OSQP returns whatever its iteration converges
    to.  Canon here (and in the HIP kernel): Tikhonov term 1/2 eps |u|^2, eps = HQP_EPS -- the least-norm member of the
    solution set in the limit.  Properties checked by tests: KKT of every level, hierarchy (a later level never degrades
    an earlier level's equality residual or slack beyond tolerance), and on the golden CASE 1 / 2 states that the LQP
    torque satisfies the rows it was asked to satisfy.
"""
import numpy as np

HQP_EPS = 1.0e-6       # Tikhonov weight of the canon (conditioning 1e6: answers reproducible to ~1e-9 relative)
HQP_TOL = 1.0e-6       # constraint violation tolerance of the active-set solver (rows are normalised by the configurator)
HQP_TOL_DEP = 1.0e-5   # a violated row that depends on the working set and has nothing to trade against is let go below this
HQP_MAX_ITER = 400
HQP_DEP = 1.0e-12      # linear-dependence threshold of the active-set step (relative curvature)
COD_EPS = np.finfo(float).eps


# ----------------------------------------------------------------------------------------------
# getNullSpace (src/math.cpp:349-360): orthonormal basis of null(A) from a rank-revealing decomposition.  Eigen's COD with its
# default threshold (epsilon * diagonal size) [ext]; any orthonormal basis of the same subspace gives the same y_ans_, because
# every quantity downstream is a function of span(Z) only.
# ----------------------------------------------------------------------------------------------
def pivoted_qr(A):
    """column-pivoted Householder QR: A[:, piv] = Q R; returns Q (m x m), R (m x n), piv, |diag| before thresholding"""
    A = np.array(A, dtype=float)
    m, n = A.shape
    Q = np.eye(m)
    piv = np.arange(n)
    steps = min(m, n)
    for s in range(steps):
        norms = (A[s:, s:] ** 2).sum(axis=0)
        j = s + int(np.argmax(norms))
        if j != s:
            A[:, [s, j]] = A[:, [j, s]]
            piv[[s, j]] = piv[[j, s]]
        x = A[s:, s].copy()
        nx = np.linalg.norm(x)
        if nx == 0.0:
            continue
        alpha = -nx if x[0] > 0 else nx
        v = x
        v[0] -= alpha
        vn2 = v @ v
        if vn2 == 0.0:
            continue
        beta = 2.0 / vn2
        A[s:, s:] -= beta * np.outer(v, v @ A[s:, s:])
        Q[:, s:] -= beta * np.outer(Q[:, s:] @ v, v)
    return Q, A, piv


def get_null_space(A, threshold=None):
    """orthonormal columns spanning null(A) (A: r x c) -> c x (c - rank)"""
    A = np.asarray(A, dtype=float)
    r, c = A.shape
    if r == 0:
        return np.eye(c)
    # null(A) = orthogonal complement of range(A^T): QR of A^T with column pivoting, rank by the pivot threshold
    Q, R, _ = pivoted_qr(A.T)
    d = np.abs(np.diag(R[: min(r, c), : min(r, c)]))
    thr = (COD_EPS * min(r, c)) if threshold is None else threshold
    rank = int((d > thr * (d.max() if d.size and d.max() > 0 else 1.0)).sum())
    return Q[:, rank:]


# ----------------------------------------------------------------------------------------------
# exact solver of one level: strictly convex QP (after the Tikhonov term) in u with soft rows (slack, unit weight) and hard rows
#     min 1/2 u^T H u + g^T u + 1/2 |max(0, As u - ds)|^2   s.t.  Ah u <= dh
# Dual active set in range-space form on the augmented variable x = (u, v): the working set holds soft rows (their slack
# positive) and hard rows (at their bound); M = C_W H^-1 C_W^T (+ 1 on the diagonal for soft rows) is refactorised every step.
# ----------------------------------------------------------------------------------------------
def solve_level_qp(H, g, As, ds, Ah, dh, max_iter=HQP_MAX_ITER, tol=HQP_TOL):
    k = H.shape[0]
    ms, mh = As.shape[0], Ah.shape[0]
    C = np.vstack([As, Ah]) if ms + mh else np.zeros((0, k))
    d = np.concatenate([ds, dh]) if ms + mh else np.zeros(0)
    soft = np.concatenate([np.ones(ms, bool), np.zeros(mh, bool)])
    L = np.linalg.cholesky(H)
    Hinv = lambda b: np.linalg.solve(L.T, np.linalg.solve(L, b))
    u = -Hinv(g)
    v = np.zeros(ms)
    W, lam, ignored = [], [], []
    it = 0
    status = 1
    p = -1
    while True:
        if p < 0:
            if C.shape[0] == 0:
                break
            sl = d - C @ u
            sl[:ms] += v
            sl_w = sl.copy()
            sl_w[W] = np.inf
            sl_w[ignored] = np.inf
            p = int(np.argmin(sl_w))
            if not sl_w[p] < -tol:
                break
            lam_p = 0.0
        it += 1
        if it > max_iter:
            status = 0
            break
        cp = C[p]
        # direction: z = H^-1 (c_p - C_W^T r) in u, slack part separately; r solves M r = C_W Haug^-1 n_p
        if W:
            CW = C[W]
            T = np.array([Hinv(c) for c in CW])  # q x k
            M = CW @ T.T
            for a, j in enumerate(W):
                if soft[j]:
                    M[a, a] += 1.0
            rhs = T @ cp
            if soft[p]:
                pass  # different rows: the slack columns are orthogonal
            r = np.linalg.solve(M, rhs)
            zu = Hinv(cp - CW.T @ r)
        else:
            r = np.zeros(0)
            zu = Hinv(cp)
        # slack directions: soft row j in W has v_j moving by +r_j (its normal has -1 on v_j: dv = -(-1) * (-r_j) ... derived below)
        # augmented normal of row j: n_j = (c_j, -e_j [soft]);  Haug^-1 = blkdiag(H^-1, I);  z = Haug^-1 (n_p - N_W r)
        zv = np.zeros(ms)
        if soft[p]:
            zv[p] += -1.0
        for a, j in enumerate(W):
            if soft[j]:
                zv[j] -= -1.0 * r[a]
        # curvature along the violated normal: n_p . z
        nz = cp @ zu - (zv[p] if soft[p] else 0.0)
        slack_p = d[p] - cp @ u + (v[p] if soft[p] else 0.0)  # negative
        if not slack_p < 0.0:  # a dual-only step (a row left the working set) has already satisfied it
            p = -1
            continue
        # the violated normal counts as independent of the working set only if a fraction > HQP_DEP of its H^-1-norm survives
        # the projection (a dependent row gives a dual-only step: some active row leaves first)
        full = cp @ Hinv(cp) + (1.0 if soft[p] else 0.0)
        t2 = (-slack_p) / nz if nz > HQP_DEP * full else np.inf
        t1, drop = np.inf, -1
        for a in range(len(W)):
            if r[a] > 1e-14:
                ta = lam[a] / r[a]
                if ta < t1:
                    t1, drop = ta, a
        t = min(t1, t2)
        if not np.isfinite(t):
            if -slack_p < HQP_TOL_DEP:  # round-off on a redundant row: not a violation
                ignored.append(p)
                p = -1
                continue
            status = 0
            break
        # primal step moves AGAINST the violated normal (rows are c x <= d): x -= t z ; multipliers: lam_W -= t r, lam_p += t
        u = u - t * zu
        v = v - t * zv
        lam = [l - t * ra for l, ra in zip(lam, r)]
        lam_p += t
        if t2 <= t1:
            W.append(p)
            lam.append(lam_p)
            p = -1
        else:
            W.pop(drop)
            lam.pop(drop)
    return status, u, v, it, sorted(W)


# ----------------------------------------------------------------------------------------------
# the classes (field names as in include/dwbc_hqp.h)
# ----------------------------------------------------------------------------------------------
class HQP_Hierarch:
    def __init__(self, level, acc, torque, contact, ineq, eq):  # HQP_Hierarch::initialize, dwbc_hqp.cpp:436-481
        self.hierarchy_level_ = level
        self.variable_size_ = acc + torque + contact
        self.ineq_const_size_, self.eq_const_size_ = ineq, eq
        n = self.variable_size_
        self.A_, self.a_ = np.zeros((ineq, n)), np.zeros(ineq)
        self.v_ans_ = np.zeros(ineq)
        self.B_, self.b_ = np.zeros((eq, n)), np.zeros(eq)
        self.y_ans_, self.w_ans_ = np.zeros(n), np.zeros(eq)
        self.enable_cost_ = False
        self.H_, self.g_ = None, None
        self.Z_ = None
        self.null_space_size_ = 0
        self.qp_iter_ = 0
        self.qp_status_ = 1
        self.exact_ = False  # JACC: equalities hold exactly (least-norm step), inequalities only declared (slack 0)
        self.V_, self.W_ = np.eye(ineq), np.eye(eq)  # row weights, read by solvefirst only (dwbc_hqp.cpp:245-254, 503-553)

    def updateInequalityCostWeight(self, V):  # dwbc_hqp.cpp:503-510 (a vector is a diagonal)
        V = np.array(V, float)
        self.V_ = np.diag(V) if V.ndim == 1 else V

    def updateEqualityCostWeight(self, W):  # dwbc_hqp.cpp:521-528
        W = np.array(W, float)
        self.W_ = np.diag(W) if W.ndim == 1 else W

    def updateConstraintWeight(self, V, W):  # dwbc_hqp.cpp:549-553
        self.V_, self.W_ = np.array(V, float), np.array(W, float)

    def updateConstraintMatrix(self, A, a, B, b):  # dwbc_hqp.cpp:530-547
        if self.ineq_const_size_ > 0:
            assert A.shape == (self.ineq_const_size_, self.variable_size_) and a.shape == (self.ineq_const_size_,)
            self.A_, self.a_ = np.array(A, float), np.array(a, float)
        assert B.shape == (self.eq_const_size_, self.variable_size_) and b.shape == (self.eq_const_size_,)
        self.B_, self.b_ = np.array(B, float), np.array(b, float)

    def updateCostMatrix(self, H, g):  # dwbc_hqp.cpp:483-493
        self.enable_cost_ = True
        self.H_, self.g_ = np.array(H, float), np.array(g, float)

    def normalizeConstraintMatrix(self):  # dwbc_hqp.cpp:555-581
        for M, v in ((self.A_, self.a_), (self.B_, self.b_)):
            for i in range(M.shape[0]):
                nrm = np.linalg.norm(M[i])
                if nrm > 0:
                    M[i] /= nrm
                    v[i] /= nrm


class HQP:
    def __init__(self):
        self.hqp_hs_ = []

    def initialize(self, acceleration_size, torque_size, contact_size):  # dwbc_hqp.cpp:16-21
        self.acceleration_size_, self.torque_size_, self.contact_size_ = acceleration_size, torque_size, contact_size

    def addHierarchy(self, ineq_const_size, eq_const_size):  # dwbc_hqp.cpp:425-434
        self.hqp_hs_.append(HQP_Hierarch(len(self.hqp_hs_), self.acceleration_size_, self.torque_size_, self.contact_size_, ineq_const_size, eq_const_size))

    def prepare(self):  # dwbc_hqp.cpp:23-85
        hs = self.hqp_hs_
        hs[0].Z_ = get_null_space(hs[0].B_)
        hs[0].null_space_size_ = hs[0].Z_.shape[1]
        for i in range(1, len(hs)):
            nullB = get_null_space(hs[i].B_ @ hs[i - 1].Z_)
            hs[i].Z_ = hs[i - 1].Z_ @ nullB
            hs[i].null_space_size_ = hs[i].Z_.shape[1]

    def _solve_exact(self, i, y_prev, Z):
        h = self.hqp_hs_[i]
        BZ = h.B_ @ Z
        r = h.B_ @ y_prev + h.b_
        u = -np.linalg.lstsq(BZ, r, rcond=1e-12)[0]  # least-norm solution of B Z u = -r
        h.y_ans_ = y_prev + Z @ u
        h.v_ans_ = np.zeros(h.ineq_const_size_)
        h.w_ans_ = h.B_ @ h.y_ans_ + h.b_
        h.qp_iter_, h.working_set_ = 0, []
        h.qp_status_ = int(np.abs(h.w_ans_).max() < 1e-6) if h.eq_const_size_ else 1
        return h.qp_status_

    def _solve(self, i, y_prev, Z):
        h = self.hqp_hs_[i]
        if h.exact_:
            return self._solve_exact(i, y_prev, Z)
        Bz = h.B_ @ Z
        r = h.B_ @ y_prev + h.b_
        k = Z.shape[1]
        H = Bz.T @ Bz + HQP_EPS * np.eye(k)
        g = Bz.T @ r
        if h.enable_cost_:
            H = H + Z.T @ h.H_ @ Z
            g = g + Z.T @ (h.H_ @ y_prev)
        As = h.A_ @ Z if h.ineq_const_size_ > 0 else np.zeros((0, k))
        ds = -(h.A_ @ y_prev) - h.a_ if h.ineq_const_size_ > 0 else np.zeros(0)
        Ah, dh = [np.zeros((0, k))], [np.zeros(0)]
        for j in range(i):
            hj = self.hqp_hs_[j]
            if hj.ineq_const_size_ > 0:
                Ah.append(hj.A_ @ Z)
                dh.append(-(hj.A_ @ y_prev) + hj.v_ans_ - hj.a_)
        st, u, v, it, W = solve_level_qp(H, g, As, ds, np.vstack(Ah), np.concatenate(dh))
        h.qp_iter_, h.qp_status_, h.working_set_ = it, st, W
        h.y_ans_ = y_prev + Z @ u
        if h.ineq_const_size_ > 0:
            h.v_ans_ = v
        h.w_ans_ = h.B_ @ h.y_ans_ + h.b_
        return st

    def solvefirst(self):  # dwbc_hqp.cpp:222-289: level 0 over the full variable (no null-space restriction), posed on V A, V a, W B, W b
        h = self.hqp_hs_[0]
        n = h.variable_size_
        keep = (h.A_, h.a_, h.B_, h.b_)
        if h.ineq_const_size_ > 0:
            h.A_, h.a_ = h.V_ @ h.A_, h.V_ @ h.a_
        h.B_, h.b_ = h.W_ @ h.B_, h.W_ @ h.b_
        st = self._solve(0, np.zeros(n), np.eye(n))
        h.A_, h.a_, h.B_, h.b_ = keep  # (every later level reads the unweighted matrices, as the reference does)
        return st

    def solveSequentialSingle(self, level):  # dwbc_hqp.cpp:291-395
        prev = self.hqp_hs_[level - 1]
        return self._solve(level, prev.y_ans_, prev.Z_)

    def solveSequential(self):  # dwbc_hqp.cpp:397-403 (an exact level 0 is evaluated here as well)
        ok = 1
        if self.hqp_hs_[0].exact_:
            ok &= self.solvefirst()
        for i in range(1, len(self.hqp_hs_)):
            ok &= self.solveSequentialSingle(i)
        return ok


# ----------------------------------------------------------------------------------------------
# RobotData::ConfigureLQP (src/dwbc.cpp:4304-4430) on the fields of a dwbc_np.Cycle after update_kinematics / set_contact
# ----------------------------------------------------------------------------------------------
LQP_TAU_LIM = 200.0  # `tlim`, hard-coded in the reference (dwbc.cpp:4360)
LQP_ACC_LIM = 5.0    # `alim` (dwbc.cpp:4398)


def configure_lqp(c, B_nle, J_tasks, f_stars, cost_norm=None, tau_lim=None):
    """c: dwbc_np.Cycle (A, A_inv, J_C, cone_matrix()); B_nle = RobotData::B_ (n); J_tasks / f_stars per task level.
    cost_norm / tau_lim: what ConfigureLQP_R changes (the norm of the FULL A_ scales the cost, one torque limit is 600)"""
    n, m, cd = c.n, c.m, c.cdof
    nv = n + cd
    hqp = HQP()
    hqp.initialize(n, 0, cd)
    cost_h = np.zeros((nv, nv))
    cost_h[:n, :n] = c.A / (np.linalg.norm(c.A) if cost_norm is None else cost_norm) * 5.0
    tl = np.full(m, LQP_TAU_LIM) if tau_lim is None else np.asarray(tau_lim, float)
    cost_g = np.zeros(nv)
    JCt = c.J_C.T
    # priority 1: torque limit (inequality), floating-base dynamics (equality); "solved" analytically
    hqp.addHierarchy(2 * m, 6)
    A = np.zeros((2 * m, nv))
    a = np.zeros(2 * m)
    Bm = np.zeros((6, nv))
    Bm[:, :n] = c.A[:6]
    Bm[:, n:] = JCt[:6]
    b = B_nle[:6].copy()
    A[:m, :n] = c.A[6:]
    A[:m, n:] = JCt[6:]
    A[m:, :n] = -c.A[6:]
    A[m:, n:] = -JCt[6:]
    a[:m] = -tl + B_nle[6:]
    a[m:] = -tl - B_nle[6:]
    h0 = hqp.hqp_hs_[0]
    h0.updateConstraintMatrix(A, a, Bm, b)
    h0.normalizeConstraintMatrix()
    h0.v_ans_ = np.zeros(2 * m)
    h0.w_ans_ = np.zeros(6)
    h0.y_ans_ = np.zeros(nv)
    h0.y_ans_[:n] = -c.A_inv @ B_nle
    # priority 2: contact cones + joint acceleration limit (inequality), contact constraint (equality), cost
    ncc = 10 * len(c.act_contacts)
    hqp.addHierarchy(ncc + 2 * m, cd)
    A = np.zeros((ncc + 2 * m, nv))
    a = np.zeros(ncc + 2 * m)
    A[:ncc, n:] = -c.cone_matrix()  # getContactConstraintMatrix(): C = -A_const_a A_rot (dwbc.cpp:512)
    A[ncc : ncc + m, 6 : 6 + m] = np.eye(m)
    A[ncc + m :, 6 : 6 + m] = -np.eye(m)
    a[ncc:] = -LQP_ACC_LIM
    Bm = np.zeros((cd, nv))
    Bm[:, :n] = c.J_C
    h1 = hqp.hqp_hs_[1]
    h1.updateConstraintMatrix(A, a, Bm, np.zeros(cd))
    h1.updateCostMatrix(cost_h, cost_g)
    h1.normalizeConstraintMatrix()
    # priorities 3..: one equality level per task space
    for i, (J, f) in enumerate(zip(J_tasks, f_stars)):
        t = J.shape[0]
        hqp.addHierarchy(0, t)
        Bm = np.zeros((t, nv))
        Bm[:, :n] = J
        hi = hqp.hqp_hs_[2 + i]
        hi.updateConstraintMatrix(None, None, Bm, -np.asarray(f, float))
        hi.updateCostMatrix(cost_h, cost_g)
        hi.normalizeConstraintMatrix()
    hqp.prepare()
    return hqp


def lqp_torque(c, B_nle, y):
    """tests/sp_test/jacc_compare.cpp:416-418: tau = A[6:] qdd + J_C^T[6:] f_c + B_[6:]"""
    n = c.n
    return c.A[6:] @ y[:n] + c.J_C.T[6:] @ y[n:] + B_nle[6:]


# ----------------------------------------------------------------------------------------------
# RobotData::CalcSingleTaskTorqueWithJACC_QP (src/dwbc.cpp:3772-3945) on the same machinery: tau and the task slack are
# eliminated from the reference's single QP over [qddot; tau; f_c; s] (see libdwbc_amd/csrc/dwbc_hqp.h, jacc_configure_instance)
# ----------------------------------------------------------------------------------------------
JACC_ACC_LIM, JACC_TAU_LIM = 10.0, 200.0


def jacc_qp(c, level, J_tasks, f_stars, fqp_prev, tau_rows=None):
    """c: dwbc_np.Cycle after update_kinematics / set_contact.  Returns (ok, acc_qp, torque_qp, contact_qp, f_star_qp, hqp).
    tau_rows: how many of the m torques carry the +-200 bound (JACC_QP_R: all but the six virtual ones, dwbc.cpp:4096-4097)"""
    n, m, cd = c.n, c.m, c.cdof
    mt = m if tau_rows is None else tau_rows
    nv = n + cd
    JCt = c.J_C.T
    ncc = 10 * len(c.act_contacts)
    hq = HQP()
    hq.initialize(n, 0, cd)
    e0 = 6 + cd + sum(J_tasks[i].shape[0] for i in range(level))
    hq.addHierarchy(ncc + 2 * m + 2 * mt, e0)
    A = np.zeros((ncc + 2 * m + 2 * mt, nv))
    a = np.zeros(ncc + 2 * m + 2 * mt)
    A[:ncc, n:] = -c.cone_matrix()
    A[ncc : ncc + m, 6 : 6 + m] = np.eye(m)
    A[ncc + m : ncc + 2 * m, 6 : 6 + m] = -np.eye(m)
    a[ncc : ncc + 2 * m] = -JACC_ACC_LIM
    D = np.hstack([c.A[6 : 6 + mt], JCt[6 : 6 + mt]])
    A[ncc + 2 * m : ncc + 2 * m + mt] = D
    A[ncc + 2 * m + mt :] = -D
    a[ncc + 2 * m : ncc + 2 * m + mt] = -JACC_TAU_LIM + c.G[6 : 6 + mt]
    a[ncc + 2 * m + mt :] = -JACC_TAU_LIM - c.G[6 : 6 + mt]
    B = np.zeros((e0, nv))
    b = np.zeros(e0)
    B[:6, :n] = c.A[:6]
    B[:6, n:] = JCt[:6]
    b[:6] = c.G[:6]
    B[6 : 6 + cd, :n] = c.J_C
    row = 6 + cd
    for i in range(level):
        t = J_tasks[i].shape[0]
        B[row : row + t, :n] = J_tasks[i]
        b[row : row + t] = -(np.asarray(f_stars[i]) + np.asarray(fqp_prev[i]))
        row += t
    h0 = hq.hqp_hs_[0]
    h0.updateConstraintMatrix(A, a, B, b)
    h0.normalizeConstraintMatrix()
    h0.exact_ = True
    J = J_tasks[level]
    t = J.shape[0]
    hq.addHierarchy(0, t)
    B1 = np.zeros((t, nv))
    B1[:, :n] = 10.0 * J
    H = np.zeros((nv, nv))
    H[:n, :n] = c.A
    h1 = hq.hqp_hs_[1]
    h1.updateConstraintMatrix(None, None, B1, -10.0 * np.asarray(f_stars[level], float))
    h1.updateCostMatrix(H, np.zeros(nv))
    hq.prepare()
    ok = hq.solveSequential()
    y = h1.y_ans_
    tau = c.A[6:] @ y[:n] + JCt[6:] @ y[n:] + c.G[6:]
    return ok, y[:n].copy(), tau, y[n:].copy(), J @ y[:n] - np.asarray(f_stars[level], float), hq


# ----------------------------------------------------------------------------------------------
# The same formulations on the REDUCED system (contact chains + 6 centroidal coordinates of the other bodies):
#   RobotData::ConfigureLQP_R / CalcControlTorqueLQP_R              src/dwbc.cpp:4504-4632, 4455-4477
#   RobotData::ConfigureLQP_R_NC / CalcControlTorqueLQP_R_NC        src/dwbc.cpp:4634-4760, 4479-4502
#   RobotData::CalcSingleTaskTorqueWithJACC_QP_R                    src/dwbc.cpp:3946-4122
#   RobotData::CalcSingleTaskTorqueWithJACC_QP_R_NC                 src/dwbc.cpp:4124-4302
# The _R functions are the full-model ones with (A_, J_C, G_ / B_, J_task) replaced by (A_R, J_CR, G_R, J_task J_R_INV_T^T);
# the differences are the arguments of configure_lqp / jacc_qp above.  PARITY UNPINNED like everything in this file; the
# reference's only callers are the timing harnesses tests/sp_test/{jacc_compare, dof_comparison, dof_comparison_jacc}.cpp.
# ----------------------------------------------------------------------------------------------
LQP_R_TAU_LIM_SPECIAL = 600.0  # `tlim(tlim_size - 4) = 600`, src/dwbc.cpp:4552: the vertical centroidal force of the body group
LQP_NC_TAU_LIM, LQP_NC_ACC_LIM = 200.0, 5.0  # src/dwbc.cpp:4666, 4744-4748
JACC_NC_W_TASK = 5.0  # src/dwbc.cpp:4163


class ReducedView:
    """the reduced system of an oracle ReducedCycle (after reduced_dynamics / reduced_contact_constraint) under the names
    configure_lqp / jacc_qp read from a Cycle"""

    def __init__(self, r):
        self.A, self.A_inv, self.J_C, self.G = r.A_R, r.A_R_inv, r.J_CR, r.G_R
        self.n = r.r_sys_dof
        self.m = r.r_model_dof
        self.cdof = r.cdof
        self.act_contacts = r.act_contacts
        self.cone_matrix = r.cone_matrix


def reduced_task_jacobians(r, J_tasks, noncont):
    """J_task J_R_INV_T^T of the contact-chain levels (src/dwbc.cpp:4619), in level order"""
    return [J @ r.J_R_INV_T.T for J, nc in zip(J_tasks, noncont) if not nc]


def configure_lqp_r(r, J_tasks, f_stars, noncont):
    """ConfigureLQP_R.  The reference writes hierarchy 2 + i for task i and so skips a slot when a non-contact level
    precedes a contact-chain one (src/dwbc.cpp:4608-4626); here the contact-chain levels are packed in order."""
    v = ReducedView(r)
    tl = np.full(v.m, LQP_TAU_LIM)
    tl[v.m - 4] = LQP_R_TAU_LIM_SPECIAL
    Jr = reduced_task_jacobians(r, J_tasks, noncont)
    fr = [f for f, nc in zip(f_stars, noncont) if not nc]
    return configure_lqp(v, r.G_R, Jr, fr, cost_norm=np.linalg.norm(r.A), tau_lim=tl)


def lqp_r_torque(r, y):
    """A_R[6:] qdd_R + J_CR^T[6:] f_c + G_R[6:]: the rows ConfigureLQP_R bounds (12 chain torques + the 6-D wrench on the
    centroidal coordinates)"""
    v = ReducedView(r)
    return lqp_torque(v, r.G_R, y)


def _nc_task_local(r, J_task, f_star, link_pos, base_acc):
    """fstar_local = Ja (f* - base acceleration), Ja = [[I, skew(x_link - x_pelvis)], [0, I]] (src/dwbc.cpp:4144-4147, 4728-4731)"""
    d = np.asarray(link_pos, float) - r.p[0]
    Ja = np.eye(6)
    Ja[0:3, 3:6] = np.array([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]])
    return Ja @ (np.asarray(f_star, float) - base_acc)


def configure_lqp_r_nc(r, q_acc, J_task, f_star, link_pos):
    """ConfigureLQP_R_NC for ONE 6-D non-contact level (the reference hard-codes ts_[1]).  q_acc: the reduced answer
    [base 6 | chain joints | centroidal 6]."""
    ncd, vcd = r.nc_dof, r.vc_dof
    Ann = r.A[vcd:, vcd:]
    Gn = r.G[vcd:]
    hq = HQP()
    hq.initialize(ncd, 0, 0)
    cost = Ann / np.linalg.norm(Ann) * 5.0
    hq.addHierarchy(2 * ncd, 6)
    A = np.vstack([Ann, -Ann])
    a = np.concatenate([-LQP_NC_TAU_LIM + Gn, -LQP_NC_TAU_LIM - Gn])
    h0 = hq.hqp_hs_[0]
    h0.updateConstraintMatrix(A, a, r.J_I_nc.copy(), -np.asarray(q_acc[-6:], float))
    h0.updateCostMatrix(cost, np.zeros(ncd))
    hq.addHierarchy(2 * ncd, 6)
    A = np.vstack([np.eye(ncd), -np.eye(ncd)])
    a = np.full(2 * ncd, -LQP_NC_ACC_LIM)
    h1 = hq.hqp_hs_[1]
    h1.updateConstraintMatrix(A, a, J_task[:, vcd:].copy(), -_nc_task_local(r, J_task, f_star, link_pos, np.asarray(q_acc[:6], float)))
    h1.updateCostMatrix(cost, np.zeros(ncd))
    hq.prepare()
    return hq


def solve_lqp_r_nc(hq):
    """CalcControlTorqueLQP_R_NC: solvefirst, then the remaining levels"""
    ok = hq.solvefirst()
    for i in range(1, len(hq.hqp_hs_)):
        ok = hq.solveSequentialSingle(i) and ok
    return ok


def jacc_qp_r(r, level, J_tasks, f_stars, fqp_prev):
    """CalcSingleTaskTorqueWithJACC_QP_R; J_tasks: full-model Jacobians of the contact-chain levels"""
    v = ReducedView(r)
    Jr = [J @ r.J_R_INV_T.T for J in J_tasks]
    return jacc_qp(v, level, Jr, f_stars, fqp_prev, tau_rows=v.m - 6)


def jacc_qp_r_nc(r, prev_acc, J_task, f_star, link_pos):
    """CalcSingleTaskTorqueWithJACC_QP_R_NC.  The bounds the reference prepares are removed again before the solve
    (`DeleteSubjectToX`, src/dwbc.cpp:4277) and tau only appears in its own defining equality, so the problem is
        min 1/2 |J_I_nc a - gacc_prev|^2 + 5/2 |J_task[:, nc] a - fstar_local|^2   over a (nc_dof),
    under-determined (12 residuals, 21 unknowns): canon = least norm (Tikhonov HQP_EPS).
    Returns (acc_qp, torque_qp, gacc_qp, f_star_qp)."""
    ncd, vcd = r.nc_dof, r.vc_dof
    fl = _nc_task_local(r, J_task, f_star, link_pos, np.asarray(prev_acc[:6], float))
    t = J_task.shape[0]
    w = np.sqrt(JACC_NC_W_TASK)
    hq = HQP()
    hq.initialize(ncd, 0, 0)
    hq.addHierarchy(0, 6 + t)
    B = np.vstack([r.J_I_nc, w * J_task[:, vcd:]])
    b = -np.concatenate([np.asarray(prev_acc[-6:], float), w * fl])
    hq.hqp_hs_[0].updateConstraintMatrix(None, None, B, b)
    hq.prepare()
    hq.solvefirst()
    a = hq.hqp_hs_[0].y_ans_[:ncd].copy()
    return a, r.A[vcd:, vcd:] @ a + r.G[vcd:], r.J_I_nc @ a - np.asarray(prev_acc[-6:], float), J_task[:, vcd:] @ a - fl, hq

"""ctypes binding of oracle/liborc.so (the plain-C CPU restatement).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("OMP_STACKSIZE", "32M")  # orc_cycle keeps ~0.5 MB of matrices on the stack
MAXB, MAXN, MAXM, MAXCON, MAXC, MAXL, MAXTL, MAXT, MAXV, MAXR = 48, 54, 48, 4, 24, 4, 2, 12, 30, 160

d = C.c_double
i32 = C.c_int


class Model(C.Structure):
    _fields_ = [
        ("nb", i32), ("ndof", i32), ("parent", i32 * MAXB),
        ("R_T", d * 9 * MAXB), ("p_T", d * 3 * MAXB), ("axis", d * 3 * MAXB),
        ("mass", d * MAXB), ("com", d * 3 * MAXB), ("inertia", d * 9 * MAXB),
    ]


class Setup(C.Structure):
    _fields_ = [
        ("n_contacts", i32), ("c_link", i32 * MAXCON), ("c_point", d * 3 * MAXCON),
        ("c_lx", d * MAXCON), ("c_ly", d * MAXCON), ("c_mu", d * MAXCON), ("c_muz", d * MAXCON),
        ("n_levels", i32), ("t_nlinks", i32 * MAXL), ("t_mode", i32 * MAXTL * MAXL), ("t_link", i32 * MAXTL * MAXL),
        ("t_point", d * 3 * MAXTL * MAXL), ("has_tau_lim", i32), ("tau_lim", d * MAXM),
    ]


class Out(C.Structure):
    _fields_ = [
        ("status", i32), ("st_contact", i32), ("st_task", i32), ("st_redis", i32), ("cdof", i32), ("k", i32),
        ("task_dof", i32 * MAXL), ("qp_iter", i32 * (MAXL + 1)), ("qp_nact", i32 * (MAXL + 1)),
        ("qp_act", i32 * MAXV * (MAXL + 1)),
        ("tau_grav", d * MAXM), ("tau_task", d * MAXM), ("tau_contact", d * MAXM), ("contact_force", d * MAXC),
        ("fstar_qp", d * MAXT * MAXL), ("contact_qp", d * MAXC * MAXL), ("cf_redis", d * MAXC),
        ("G", d * MAXN), ("P_C", d * MAXC), ("com", d * 3),
    ]


class Debug(C.Structure):
    _fields_ = [
        ("A", d * (MAXN * MAXN)), ("A_inv", d * (MAXN * MAXN)),
        ("J_C", d * (MAXC * MAXN)), ("Lambda_c", d * (MAXC * MAXC)), ("J_C_INV_T", d * (MAXC * MAXN)),
        ("N_C", d * (MAXN * MAXN)), ("A_inv_N_C", d * (MAXN * MAXN)),
        ("W", d * (MAXM * MAXM)), ("W_inv", d * (MAXM * MAXM)),
        ("V2", d * (MAXC * MAXM)), ("NwJw", d * (MAXM * MAXC)), ("CMM", d * (6 * MAXN)),
        ("J_task", d * (MAXT * MAXN) * MAXL), ("Lambda_task", d * (MAXT * MAXT) * MAXL),
        ("J_kt", d * (MAXM * MAXT) * MAXL), ("Null_task", d * (MAXM * MAXM) * MAXL),
        ("qpA", d * (MAXR * MAXV) * (MAXL + 1)), ("qpub", d * MAXR * (MAXL + 1)),
        ("qp_rows", i32 * (MAXL + 1)), ("qp_cols", i32 * (MAXL + 1)),
        ("link_R", d * 9 * MAXB), ("link_p", d * 3 * MAXB),
    ]


_lib = None


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    src = [os.path.join(_HERE, f) for f in ("dwbc_oracle.c", "dwbc_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liborc.so"])
    return so


def lib():
    global _lib
    if _lib is None:
        so = build()
        L = C.CDLL(so)
        assert L.orc_sizeof_model() == C.sizeof(Model), (L.orc_sizeof_model(), C.sizeof(Model))
        assert L.orc_sizeof_setup() == C.sizeof(Setup), (L.orc_sizeof_setup(), C.sizeof(Setup))
        assert L.orc_sizeof_out() == C.sizeof(Out), (L.orc_sizeof_out(), C.sizeof(Out))
        assert L.orc_sizeof_debug() == C.sizeof(Debug), (L.orc_sizeof_debug(), C.sizeof(Debug))
        L.orc_cycle.argtypes = [C.POINTER(Model), C.POINTER(Setup), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Out), C.c_void_p]
        L.orc_cycle.restype = None
        L.orc_cycle_batch.argtypes = [C.POINTER(Model), C.POINTER(Setup), i32] + [C.c_void_p] * 3 + [i32] + [C.c_void_p] * 3 + [i32]
        L.orc_cycle_batch.restype = i32
        L.orc_solve_qp.argtypes = [C.c_void_p, C.c_void_p, i32, i32, i32, i32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_solve_qp.restype = i32
        L.orc_pinv_cod.argtypes = [C.c_void_p, i32, i32, d, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_pinv_cod.restype = i32
        _lib = L
    return _lib


def make_model(m):
    M = Model()
    M.nb, M.ndof = int(m["nb"]), int(m["ndof"])
    for i in range(M.nb):
        M.parent[i] = int(m["parent"][i])
        M.mass[i] = float(m["mass"][i])
        for a in range(3):
            M.p_T[i][a] = float(m["p_T"][i][a])
            M.axis[i][a] = float(m["axis"][i][a])
            M.com[i][a] = float(m["com"][i][a])
        for a in range(9):
            M.R_T[i][a] = float(np.asarray(m["R_T"][i]).reshape(-1)[a])
            M.inertia[i][a] = float(np.asarray(m["inertia"][i]).reshape(-1)[a])
    return M


def make_setup(contacts, tasks, tau_lim=None):
    """contacts: list of dict(link, point, lx, ly, mu, muz); tasks: list of levels, each a list of (mode, link, point)."""
    S = Setup()
    S.n_contacts = len(contacts)
    for i, c in enumerate(contacts):
        S.c_link[i] = int(c["link"])
        for a in range(3):
            S.c_point[i][a] = float(c["point"][a])
        S.c_lx[i], S.c_ly[i] = float(c["lx"]), float(c["ly"])
        S.c_mu[i], S.c_muz[i] = float(c.get("mu", 0.2)), float(c.get("muz", 0.2))
    S.n_levels = len(tasks)
    for l, lv in enumerate(tasks):
        S.t_nlinks[l] = len(lv)
        for j, (mode, link, pt) in enumerate(lv):
            S.t_mode[l][j] = int(mode)
            S.t_link[l][j] = int(link)
            for a in range(3):
                S.t_point[l][j][a] = float(pt[a])
    if tau_lim is not None:
        S.has_tau_lim = 1
        for i, v in enumerate(tau_lim):
            S.tau_lim[i] = float(v)
    return S


def _np(arr, shape):
    return np.ctypeslib.as_array(arr).reshape(-1)[: int(np.prod(shape))].reshape(shape).copy()


def cycle(M, S, q, flags, fstar, debug=False):
    L = lib()
    q = np.ascontiguousarray(q, dtype=np.float64)
    flags = np.ascontiguousarray(flags, dtype=np.uint8)
    fstar = np.ascontiguousarray(np.concatenate([np.asarray(f, float).ravel() for f in fstar]), dtype=np.float64)
    out = Out()
    dbg = Debug() if debug else None
    L.orc_cycle(C.byref(M), C.byref(S), q.ctypes.data, flags.ctypes.data, fstar.ctypes.data, C.byref(out),
                C.addressof(dbg) if debug else None)
    return out, dbg


def out_to_dict(M, S, out, dbg=None):
    n, m = M.ndof, M.ndof - 6
    cd, k, L = out.cdof, out.k, S.n_levels
    r = dict(
        status=out.status, st_contact=out.st_contact, st_task=out.st_task, st_redis=out.st_redis, cdof=cd, k=k,
        tau_grav=_np(out.tau_grav, (m,)), tau_task=_np(out.tau_task, (m,)), tau_contact=_np(out.tau_contact, (m,)),
        contact_force=_np(out.contact_force, (cd,)), G=_np(out.G, (n,)), P_C=_np(out.P_C, (cd,)), com=_np(out.com, (3,)),
        cf_redis=_np(out.cf_redis, (k,)),
        fstar_qp=[_np(out.fstar_qp[l], (out.task_dof[l],)) for l in range(L)],
        contact_qp=[_np(out.contact_qp[l], (k,)) for l in range(L)],
        qp_act=[sorted(list(out.qp_act[l][: out.qp_nact[l]])) for l in range(L + 1)],
        qp_iter=[out.qp_iter[l] for l in range(L + 1)],
    )
    if dbg is not None:
        r.update(
            A=_np(dbg.A, (n, n)), A_inv=_np(dbg.A_inv, (n, n)), J_C=_np(dbg.J_C, (cd, n)), Lambda_c=_np(dbg.Lambda_c, (cd, cd)),
            J_C_INV_T=_np(dbg.J_C_INV_T, (cd, n)), N_C=_np(dbg.N_C, (n, n)), A_inv_N_C=_np(dbg.A_inv_N_C, (n, n)),
            W=_np(dbg.W, (m, m)), W_inv=_np(dbg.W_inv, (m, m)), V2=_np(dbg.V2, (k, m)), NwJw=_np(dbg.NwJw, (m, k)),
            CMM=_np(dbg.CMM, (6, n)),
            J_task=[_np(dbg.J_task[l], (out.task_dof[l], n)) for l in range(L)],
            Lambda_task=[_np(dbg.Lambda_task[l], (out.task_dof[l], out.task_dof[l])) for l in range(L)],
            J_kt=[_np(dbg.J_kt[l], (m, out.task_dof[l])) for l in range(L)],
            Null_task=[_np(dbg.Null_task[l], (m, m)) for l in range(L - 1)],
            qpA=[_np(dbg.qpA[l], (dbg.qp_rows[l], dbg.qp_cols[l])) for l in range(L + 1)],
            qpub=[_np(dbg.qpub[l], (dbg.qp_rows[l],)) for l in range(L + 1)],
            link_R=_np(dbg.link_R, (M.nb, 3, 3)), link_p=_np(dbg.link_p, (M.nb, 3)),
        )
    return r


def cycle_batch(M, S, q, flags, fstar, nthreads=0):
    """q: B x (n+1), flags: B x n_contacts (uint8), fstar: B x F.  returns tau (B,3,m), wrench (B, 6*n_contacts), status, threads."""
    L = lib()
    B = q.shape[0]
    n, m = M.ndof, M.ndof - 6
    q = np.ascontiguousarray(q, dtype=np.float64)
    flags = np.ascontiguousarray(flags, dtype=np.uint8)
    fstar = np.ascontiguousarray(fstar, dtype=np.float64)
    tau = np.zeros((B, 3, m))
    wr = np.zeros((B, 6 * S.n_contacts))
    st = np.zeros(B, dtype=np.int32)
    used = L.orc_cycle_batch(C.byref(M), C.byref(S), B, q.ctypes.data, flags.ctypes.data, fstar.ctypes.data, fstar.shape[1],
                             tau.ctypes.data, wr.ctypes.data, st.ctypes.data, nthreads)
    return tau, wr, st, used


def solve_qp(A, ub, t, max_iter=1000):
    L = lib()
    A = np.ascontiguousarray(A, dtype=np.float64)
    ub = np.ascontiguousarray(ub, dtype=np.float64)
    rows, nv = A.shape
    x = np.zeros(nv)
    act = np.zeros(MAXV, dtype=np.int32)
    nact = C.c_int(0)
    it = C.c_int(0)
    st = L.orc_solve_qp(A.ctypes.data, ub.ctypes.data, rows, nv, t, max_iter, x.ctypes.data, act.ctypes.data, C.byref(nact), C.byref(it))
    return st, x, sorted(act[: nact.value].tolist()), it.value


def pinv_cod(Mx, thr=1e-6):
    L = lib()
    Mx = np.ascontiguousarray(Mx, dtype=np.float64)
    r, c = Mx.shape
    P = np.zeros((c, r))
    V2 = np.zeros((r, r))
    rank = C.c_int(0)
    L.orc_pinv_cod(Mx.ctypes.data, r, c, thr, P.ctypes.data, V2.ctypes.data, C.byref(rank))
    return P, V2[: r - rank.value], rank.value

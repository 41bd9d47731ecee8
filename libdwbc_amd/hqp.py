"""Host-side mirror of the reference's generic hierarchical-QP class for a batch (include/dwbc_hqp.h: HQP, HQP_Hierarch) and of
RobotData::ConfigureLQP / CalcControlTorqueLQP (src/dwbc.cpp:4304-4452).

    hqp = HQP(B, acceleration_size, torque_size, contact_size)        # HQP::initialize
    lv = hqp.addHierarchy(ineq, eq)                                   # HQP::addHierarchy
    hqp.updateConstraintMatrix(lv, A, a, B, b)                        # per instance: A (B, ineq, nv) ...
    hqp.updateCostMatrix(lv, H, g); hqp.normalizeConstraintMatrix(lv)
    hqp.prepare(); hqp.solveSequential()                              # levels 1.. (solvefirst(): level 0)
    y = hqp.y_ans(lv)                                                 # hqp_hs_[lv].y_ans_

    hqp = HQP.for_lqp(wbc, contact_dof=12); wbc.solve(); hqp.configure_lqp(wbc); hqp.solveSequential(); tau = hqp.lqp_torque(wbc)

All arithmetic happens in libdwbc_hip.so (dwbc_hqp.h); this is ctypes plumbing.
"""
import numpy as np

from . import _lib
from .batch import DwbcError, _check

Y_ANS, V_ANS, W_ANS, STATUS, ITER, NULL_SIZE, MAT_A, VEC_a, MAT_B, VEC_b = range(10)


class HQP:
    def __init__(self, B, acceleration_size, torque_size, contact_size, device=0):
        self._L = _lib.load()
        self.B = int(B)
        self.nv = acceleration_size + torque_size + contact_size
        self.acceleration_size_, self.torque_size_, self.contact_size_ = acceleration_size, torque_size, contact_size
        self._h = self._L.dwbc_hqp_create(self.B, int(device), int(acceleration_size), int(torque_size), int(contact_size))
        if not self._h:
            raise DwbcError(_lib.last_error())
        self._sizes = []

    @classmethod
    def for_lqp(cls, wbc, contact_dof, device=0):
        """the object RobotData::ConfigureLQP fills: y = [qddot (system dof); f_c (contact dof)]"""
        return cls(wbc.B, wbc.n, 0, contact_dof, device=device)

    def addHierarchy(self, ineq_const_size, eq_const_size):
        lv = self._L.dwbc_hqp_add_hierarchy(self._h, int(ineq_const_size), int(eq_const_size))
        if lv < 0:
            raise DwbcError(_lib.last_error())
        self._sizes.append((int(ineq_const_size), int(eq_const_size)))
        return lv

    def _arr(self, x, shape):
        if x is None:
            return None
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(x, np.float64), shape))
        return a

    def updateConstraintMatrix(self, level, A, a, Bm, b):
        m, e = self._sizes[level]
        A_ = self._arr(A, (self.B, m, self.nv)) if m else None
        a_ = self._arr(a, (self.B, m)) if m else None
        B_ = self._arr(Bm, (self.B, e, self.nv)) if e else None
        b_ = self._arr(b, (self.B, e)) if e else None
        p = lambda x: x.ctypes.data if x is not None else None
        _check(self._L.dwbc_hqp_update_constraint_matrix(self._h, level, p(A_), p(a_), p(B_), p(b_)))

    def updateCostMatrix(self, level, H, g=None):
        H_ = self._arr(H, (self.B, self.nv, self.nv))
        g_ = self._arr(np.zeros(self.nv) if g is None else g, (self.B, self.nv))
        _check(self._L.dwbc_hqp_update_cost_matrix(self._h, level, H_.ctypes.data, g_.ctypes.data))

    def normalizeConstraintMatrix(self, level):
        _check(self._L.dwbc_hqp_normalize_constraint_matrix(self._h, level))

    def updateConstraintWeight(self, level, V=None, W=None):
        """HQP_Hierarch::updateConstraintWeight / updateInequalityCostWeight / updateEqualityCostWeight (reference
        src/dwbc_hqp.cpp:503-553): V (ineq x ineq) and W (eq x eq), full matrices or vectors taken as diagonals, one per instance or
        broadcast; None = identity.  Read by solvefirst() only, as in the reference (:245-254)."""
        m, e = self._sizes[level]

        def full(x, n):
            if x is None or n == 0:
                return None
            x = np.asarray(x, np.float64)
            if x.ndim == 1 or (x.ndim == 2 and x.shape == (self.B, n) and n != self.B):
                x = x[..., :, None] * np.eye(n)
            return self._arr(x, (self.B, n, n))

        V_, W_ = full(V, m), full(W, e)
        p = lambda x: x.ctypes.data if x is not None else None
        _check(self._L.dwbc_hqp_update_constraint_weight(self._h, level, p(V_), p(W_)))

    def set_answer(self, level, y_ans, v_ans=None):
        y_ = self._arr(y_ans, (self.B, self.nv))
        v_ = self._arr(v_ans, (self.B, self._sizes[level][0])) if v_ans is not None else None
        _check(self._L.dwbc_hqp_set_answer(self._h, level, y_.ctypes.data, v_.ctypes.data if v_ is not None else None))

    def prepare(self):
        _check(self._L.dwbc_hqp_prepare(self._h))

    def solvefirst(self, init=True):
        _check(self._L.dwbc_hqp_solve_first(self._h, 1 if init else 0))

    def solveSequential(self, init=True):
        _check(self._L.dwbc_hqp_solve_sequential(self._h, 1 if init else 0))

    # ---- RobotData::ConfigureLQP / CalcControlTorqueLQP
    def configure_lqp(self, wbc):
        _check(self._L.dwbc_batch_configure_lqp(wbc._h, self._h))
        n = self._L.dwbc_hqp_num_levels(self._h)
        m = wbc.m
        self._sizes = [(2 * m, 6), (None, self.contact_size_)] + [(0, wbc.task_dof(i)) for i in range(n - 2)]
        nb = self._L.dwbc_hqp_field_bytes(self._h, 1, V_ANS)
        self._sizes[1] = (nb // (8 * self.B), self.contact_size_)

    def lqp_torque(self, wbc):
        """B x m after configure_lqp; B x (RS - 6) after configure_lqp_r (chain torques, then the wrench on the centroidal coordinates)"""
        tau = np.zeros((self.B, self.acceleration_size_ - 6))
        _check(self._L.dwbc_batch_lqp_torque(wbc._h, self._h, tau.ctypes.data))
        return tau

    # ---- the reduced variants (src/dwbc.cpp:4455-4760, 3946-4302), after wbc.solve(reduced=True) with the dump on
    @classmethod
    def for_lqp_r(cls, wbc, reduced_system_dof, contact_dof, device=0):
        """ConfigureLQP_R's object: y = [qddot_R (reduced_system_dof = vc_dof + 6); f_c]"""
        return cls(wbc.B, reduced_system_dof, 0, contact_dof, device=device)

    @classmethod
    def for_nc(cls, wbc, nc_dof, device=0):
        """ConfigureLQP_R_NC's / JACC_QP_R_NC's object: y = accelerations of the nc_dof non-contact joints"""
        return cls(wbc.B, nc_dof, 0, 0, device=device)

    def configure_lqp_r(self, wbc):
        """RobotData::ConfigureLQP_R(hqp); then solveSequential() = CalcControlTorqueLQP_R"""
        _check(self._L.dwbc_batch_configure_lqp_r(wbc._h, self._h))
        n = self._L.dwbc_hqp_num_levels(self._h)
        m = self.acceleration_size_ - 6
        self._sizes = [(2 * m, 6), (self._L.dwbc_hqp_field_bytes(self._h, 1, V_ANS) // (8 * self.B), self.contact_size_)]
        self._sizes += [(0, self._L.dwbc_hqp_field_bytes(self._h, i, W_ANS) // (8 * self.B)) for i in range(2, n)]

    def configure_lqp_r_nc(self, wbc, hqp_r, level):
        """RobotData::ConfigureLQP_R_NC(hqp_nc, q_acc) with q_acc = the answer of the solved reduced LQP `hqp_r`; `level`: the 6-D
        task level on a non-contact link.  Then solvefirst() + solveSequential() = CalcControlTorqueLQP_R_NC"""
        _check(self._L.dwbc_batch_configure_lqp_r_nc(wbc._h, self._h, hqp_r._h, int(level)))
        nc = self.acceleration_size_
        self._sizes = [(2 * nc, 6), (2 * nc, 6)]

    def solve_jacc_r(self, wbc, level):
        """CalcSingleTaskTorqueWithJACC_QP_R(ts_[level]); `level` counts the contact-chain levels; results: jacc_result(wbc, level)"""
        _check(self._L.dwbc_batch_solve_jacc_r(wbc._h, self._h, int(level)))
        self._sizes = []

    def solve_jacc_r_nc(self, wbc, level, src_level):
        """CalcSingleTaskTorqueWithJACC_QP_R_NC(ts_[level], acc_qp_ of the reduced JACC level src_level)"""
        _check(self._L.dwbc_batch_solve_jacc_r_nc(wbc._h, self._h, int(level), int(src_level)))
        self._sizes = []

    def jacc_nc_result(self, wbc):
        """dict(acc_qp, torque_qp, gacc_qp, f_star_qp, status) after solve_jacc_r_nc"""
        nc = self.acceleration_size_
        out = {}
        for name, fid, width in (("acc_qp", 0, nc), ("torque_qp", 1, nc), ("gacc_qp", 2, 6), ("f_star_qp", 3, 6)):
            a = np.zeros((self.B, width))
            _check(self._L.dwbc_batch_get_jacc_nc(wbc._h, fid, a.ctypes.data, a.nbytes))
            out[name] = a
        st = np.zeros(self.B, np.int32)
        _check(self._L.dwbc_batch_get_jacc_nc(wbc._h, 4, st.ctypes.data, st.nbytes))
        out["status"] = st
        return out

    # ---- RobotData::CalcSingleTaskTorqueWithJACC_QP(ts_[level], init) (src/dwbc.cpp:3772-3945); levels in order 0, 1, ...
    def solve_jacc(self, wbc, level):
        _check(self._L.dwbc_batch_solve_jacc(wbc._h, self._h, int(level)))
        self._sizes = []

    @staticmethod
    def jacc_result(wbc, level, system_dof=None):
        """dict(acc_qp, torque_qp, contact_qp, f_star_qp, status) of ts_[level] after solve_jacc (system_dof = RS after solve_jacc_r)"""
        L = _lib.load()
        B = wbc.B
        out = {}
        n = wbc.n if system_dof is None else int(system_dof)
        for name, fid, width in (("acc_qp", 0, n), ("torque_qp", 1, n - 6), ("contact_qp", 2, 12), ("f_star_qp", 3, 6)):
            a = np.zeros((B, width))
            _check(L.dwbc_batch_get_jacc(wbc._h, int(level), fid, a.ctypes.data, a.nbytes))
            out[name] = a
        st = np.zeros(B, np.int32)
        _check(L.dwbc_batch_get_jacc(wbc._h, int(level), 4, st.ctypes.data, st.nbytes))
        out["status"] = st
        return out

    def num_levels(self):
        return self._L.dwbc_hqp_num_levels(self._h)

    def get(self, level, field):
        nb = self._L.dwbc_hqp_field_bytes(self._h, level, field)
        is_int = field in (STATUS, ITER, NULL_SIZE)
        out = np.zeros(nb // (4 if is_int else 8), dtype=np.int32 if is_int else np.float64)
        _check(self._L.dwbc_hqp_get(self._h, level, field, out.ctypes.data, out.nbytes))
        return out.reshape(self.B, -1) if not is_int else out

    def y_ans(self, level):
        return self.get(level, Y_ANS)

    def v_ans(self, level):
        return self.get(level, V_ANS)

    def w_ans(self, level):
        return self.get(level, W_ANS)

    def close(self):
        if self._h:
            self._L.dwbc_hqp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

"""numpy restatement of libdwbc's REDUCED (centroidal) dynamics path.   TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

PARITY UNPINNED: the reference holds no fixture and no assertion for this path; its own check is
"reduced ~ full" printed side by side by tests/sp_test/redu_dyn_test.cpp:304-317.  This restatement follows the
reference source statement by statement; tests/test_reduced_oracle.py asserts the structural identities the
reference relies on (A_R_inv = J_R A^-1 J_R^T, J_R_INV_T J_R^T = I, tau_grav reduced == full) and records the
reduced-vs-full torque gap.

Reference call sequence (tests/sp_test/redu_dyn_test.cpp:263-298):
  UpdateKinematics, SetContact, ReducedDynamicsCalculate, ReducedCalcContactConstraint, ReducedCalcGravCompensation,
  ReducedCalcTaskSpace, ReducedCalcTaskControlTorque(hqp, init, false), ReducedCalcContactRedistribute(hqp, init)

Reference functions followed (file:line in /root/reference):
  src/dwbc.cpp:2752-2990   ReducedDynamicsCalculate
  src/dwbc.cpp:3077-3142   ReducedCalcContactConstraint
  src/dwbc.cpp:3144-3150   ReducedCalcGravCompensation
  src/dwbc.cpp:3152-3253   ReducedCalcTaskSpace;  src/task.cpp:95-142 CalcJKT_R;  src/wbd.cpp:220-226 CalculateJKT_R
  src/dwbc.cpp:3255-3446   ReducedCalcTaskControlTorque
  src/dwbc.cpp:3448-3599   CalcSingleTaskTorqueWithQP_R
  src/dwbc.cpp:3601-3756   CalcSingleTaskTorqueWithQP_R_NC
  src/dwbc.cpp:3758-3770, 4776-4941   ReducedCalcContactRedistribute -> CalcContactRedistributeR (hqp branch)
  src/math.cpp:304-320     InertiaMatrixSegment / InertiaMatrix

Scope: CONTACT_6D contacts whose kinematic chains occupy the leading joint dofs (the reference's block arithmetic --
`A_inv_.block(0, vc_dof, vc_dof, nc_dof)`, `G_.segment(vc_dof, nc_dof)`, `J_task_.rightCols(nc_dof)` -- silently
assumes vc = first vc_dof system dofs; for TOCABI that is L+R double support or left single support), hqp = true,
no torque limit (the reference's reduced path is inconsistent with SetTorqueLimit, SURVEY App. C-7/C-10, and its
harness disables it, redu_dyn_test.cpp:63).
"""
import numpy as np

from . import dwbc_np as D
from .dwbc_np import Cycle, pinv_cod, skew, solve_qp


def _nc_composite_in_base(model, q, nc_links):
    """model_.Ic[2] after the masked composite pass of dwbc.cpp:2828-2848: spatial inertia of the non-contact bodies
    about the base-body origin in base-body coordinates, returned in the reference's SI_nc_b_ layout [lin; ang]
    (dwbc.cpp:2880-2885)."""
    nb = model["nb"]
    par = model["parent"]
    Ic = [np.zeros((6, 6)) for _ in range(nb)]
    Xl = [None] * nb
    for i in range(1, nb):
        Rj = D.axis_angle_R(model["axis"][i], q[6 + i - 1])
        Xl[i] = D._X((model["R_T"][i] @ Rj).T, model["p_T"][i])
    for i in nc_links:
        Ic[i] = D._spatial_inertia(model["mass"][i], model["com"][i], model["inertia"][i])
    for i in range(nb - 1, 0, -1):
        if i in nc_links:
            Ic[par[i]] = Ic[par[i]] + Xl[i].T @ Ic[i] @ Xl[i]
    I0 = Ic[0]  # [ang; lin]: [[I_o, m c^x], [-m c^x, m 1]]
    SI = np.zeros((6, 6))
    SI[0:3, 0:3] = I0[3:6, 3:6]          # m * 1
    SI[3:6, 3:6] = I0[0:3, 0:3]          # inertia about the base origin
    SI[3:6, 0:3] = I0[0:3, 3:6]          # toMatrix().block(0,3): m c^x
    SI[0:3, 3:6] = SI[3:6, 0:3].T
    return SI


def inertia_matrix_segment(SI):
    """src/math.cpp:304-310"""
    mass = SI[0, 0]
    skm = SI[3:6, 0:3] / mass
    c = np.array([skm[2, 1], skm[0, 2], skm[1, 0]])
    inertia = SI[3:6, 3:6] - mass * skew(c) @ skew(c).T
    return inertia, c, mass


class ReducedCycle(Cycle):
    """RobotData's Reduced* methods on top of the full-model Cycle (which provides UpdateKinematics / SetContact)."""

    # -- ReducedDynamicsCalculate (dwbc.cpp:2752-2990)
    def reduced_dynamics(self):
        mdl, n = self.model, self.n
        par = mdl["parent"]
        nlink = mdl["nb"]
        co_links, co_joints = [0], []
        for cc, f in zip(self.contacts, self.cflags):
            if f:
                li = cc["link"]
                while li != 0:
                    co_links.append(li)
                    co_joints.append(6 + li - 1)  # joint_[link].joint_id_ = q_index (dwbc.cpp:192)
                    li = par[li]
        co_links = sorted(co_links)
        nc_links = [i for i in range(nlink) if i not in co_links]
        co_joints = sorted(co_joints)
        vc = list(range(6)) + co_joints
        nc_joints = []
        for li in nc_links:
            while li != 0:
                jid = 6 + li - 1
                if jid not in nc_joints:
                    nc_joints.append(jid)
                li = par[li]
                if li in nc_links:
                    break
        nc_joints = sorted(nc_joints)
        self.co_links, self.nc_links, self.vc_idx, self.nc_idx = co_links, nc_links, vc, nc_joints
        self.nc_dof, self.co_dof = len(nc_joints), len(co_joints)
        self.vc_dof = self.co_dof + 6
        self.r_model_dof = self.co_dof + 6
        self.r_sys_dof = self.r_model_dof + 6
        vcd, ncd, rs = self.vc_dof, self.nc_dof, self.r_sys_dof
        if vc != list(range(vcd)) or nc_joints != list(range(vcd, n)):
            raise ValueError("reduced path needs the contact chains on the leading joint dofs (see module docstring)")

        # A_NC_O: CRBA restricted to the non-contact bodies (dwbc.cpp:2825-2878).  The composite inertia of a
        # non-contact body only collects non-contact descendants (all of its descendants are), so its rows equal the
        # full CRBA rows of those joints.
        A_NC = np.zeros((ncd + 6, ncd + 6))
        self.SI_nc_b = _nc_composite_in_base(mdl, self.q, set(nc_links))
        self.inertia_nc, self.com_pos_nc, self.mass_nc = inertia_matrix_segment(self.SI_nc_b)
        self.SI_nc_l = np.zeros((6, 6))
        self.SI_nc_l[:3, :3] = self.mass_nc * np.eye(3)
        self.SI_nc_l[3:, 3:] = self.inertia_nc
        A_NC[:6, :6] = self.SI_nc_b
        A_NC[6:, :6] = self.A[np.ix_(nc_joints, range(6))]
        A_NC[:6, 6:] = self.A[np.ix_(range(6), nc_joints)]
        A_NC[6:, 6:] = self.A[np.ix_(nc_joints, nc_joints)]
        A_NC[0:3, 6:] = self.R[0].T @ A_NC[0:3, 6:]  # dwbc.cpp:2906
        self.A_NC = A_NC
        cm_rot6 = np.eye(6)
        cm_rot6[3:, :3] = skew(self.com_pos_nc).T
        self.cmm_nc = cm_rot6 @ A_NC[:6, 6:]
        self.J_I_nc = np.linalg.solve(self.SI_nc_l, self.cmm_nc)  # 6 x nc_dof

        J_R = np.zeros((rs, n))
        for i, j in enumerate(vc):
            J_R[i, j] = 1.0
        for i, j in enumerate(nc_joints):
            J_R[rs - 6 :, j] = self.J_I_nc[:, i]
        self.J_R = J_R

        Ai = self.A_inv
        A_R_inv = np.zeros((rs, rs))
        A_R_inv[:vcd, :vcd] = Ai[np.ix_(vc, vc)]
        A_tr = Ai[vc, :]
        A_R_inv[:vcd, vcd:] = A_tr[:, n - ncd :] @ self.J_I_nc.T
        A_R_inv[vcd:, :vcd] = A_R_inv[:vcd, vcd:].T
        A_R_inv[vcd:, vcd:] = J_R[rs - 6 :] @ Ai @ J_R[rs - 6 :].T
        self.A_R_inv = A_R_inv
        self.A_R = D.llt_inverse(A_R_inv)  # dwbc.cpp:2958
        self.J_I_nc_inv_T = (
            self.A_R[vcd:, :vcd] @ Ai[:vcd, vcd : vcd + ncd] + self.A_R[vcd:, vcd:] @ self.J_I_nc @ Ai[vcd : vcd + ncd, vcd : vcd + ncd]
        )
        self.N_I_nc = np.eye(ncd) - self.J_I_nc.T @ self.J_I_nc_inv_T
        self.J_R_INV_T = np.zeros((rs, n))
        self.J_R_INV_T[:vcd, :vcd] = np.eye(vcd)
        self.J_R_INV_T[vcd:, vcd : vcd + ncd] = self.J_I_nc_inv_T
        self.G_R = np.zeros(rs)
        self.G_R[:vcd] = self.G[:vcd]
        self.G_R[vcd:] = self.J_I_nc_inv_T @ self.G[vcd : vcd + ncd]

    # -- ReducedCalcContactConstraint (dwbc.cpp:3077-3142)
    def reduced_contact_constraint(self):
        n, rs, vcd, ncd = self.n, self.r_sys_dof, self.vc_dof, self.nc_dof
        cd = self.cdof
        J_CR = np.zeros((cd, rs))
        J_CR[:, :vcd] = self.J_C[:, :vcd]
        self.J_CR = J_CR
        self.Lambda_CR = np.linalg.inv(J_CR[:, : cd + 6] @ self.A_R_inv[: cd + 6, : cd + 6] @ J_CR[:, : cd + 6].T)
        self.J_C_INV_T = (self.Lambda_CR @ self.J_C) @ self.A_inv
        self.J_CR_INV_T = np.zeros((cd, rs))
        self.J_CR_INV_T[:, :vcd] = self.J_C_INV_T[:, :vcd]
        self.J_CR_INV_T[:, vcd:] = self.J_C_INV_T[:, vcd : vcd + ncd] @ self.J_I_nc.T
        self.N_C = np.eye(n) - self.J_C.T @ self.J_C_INV_T
        self.N_CR = np.eye(rs) - J_CR.T @ self.J_CR_INV_T
        self.A_R_inv_N_CR = self.A_R_inv @ self.N_CR
        self.W_R = self.A_R_inv_N_CR[6:, 6:]
        if cd > 6:
            # PinvCODWB(W_R, W_R_inv, V2_R, cols - rows): the rank is FORCED to cols - rows (wbd.cpp:40-43)
            self.W_R_inv, V2, _rank = pinv_cod(self.W_R, want_v2=True)
            Q = _cod_q(self.W_R)
            self.V2_R = Q.T[rs - cd :, :]
            self.NwJw_R = self.V2_R.T @ np.linalg.inv(self.J_CR_INV_T[: cd - 6, 6:] @ self.V2_R.T)
            return 1 if self.V2_R.shape[0] == cd - 6 else 0
        self.W_R_inv = pinv_cod(self.W_R)
        self.V2_R = np.zeros((0, rs - 6))
        self.NwJw_R = np.zeros((rs - 6, 0))
        return 1

    # -- ReducedCalcGravCompensation (dwbc.cpp:3144-3150)
    def reduced_grav(self):
        rm, cod, ncd, vcd = self.r_model_dof, self.co_dof, self.nc_dof, self.vc_dof
        self.tau_grav_R = self.W_R_inv @ (self.A_R_inv[6:, :] @ (self.N_CR @ self.G_R))
        tg = np.zeros(self.m)
        tg[:rm] = self.tau_grav_R
        tg[cod : cod + ncd] = self.G[vcd : vcd + ncd]
        self.tau_grav = tg
        self.P_CR = self.J_CR_INV_T @ self.G_R
        return tg

    def _jkt_r(self, J_task_R):
        """CalculateJKT_R (wbd.cpp:220-226)"""
        lam = np.linalg.inv(J_task_R @ self.A_R_inv_N_CR @ J_task_R.T)
        Q = (lam @ J_task_R @ self.A_R_inv_N_CR)[:, 6:]
        Jkt = self.W_R_inv @ Q.T @ pinv_cod(Q @ self.W_R_inv @ Q.T)
        return Jkt, lam

    # -- ReducedCalcTaskSpace (dwbc.cpp:3152-3253) + TaskSpace::CalcJKT_R (task.cpp:95-142)
    def reduced_task_space(self):
        n, rs, vcd, ncd = self.n, self.r_sys_dof, self.vc_dof, self.nc_dof
        rm = self.r_model_dof
        self.A_inv_N_C = self.A_inv @ self.N_C
        L = len(self.tasks)
        self.J_task = [self.task_jacobian(i) for i in range(L)]
        J_base = point_jac_base(self)[:, :rs]  # link_[0].jac_.leftCols(reduced_system_dof_)
        self.J_base_R_kt, self.lambda_base_R = self._jkt_r(J_base)
        nch = 0
        self.kind = []
        for i in range(L):
            red = nc = cmm = False
            nc_h = -1
            for _mode, link, _pt in self.tasks[i]:
                if link == self.model["nb"]:  # the synthetic COM link (link_num_), dwbc.cpp:3190
                    cmm = True
                elif link in self.co_links:
                    red = True
                else:
                    nc = True
                    nc_h = nch
                    nch += 1
            self.kind.append(dict(reduced=red, noncont=nc, cmm=cmm, nc_h=nc_h))
        self.J_task_R, self.Lambda_t, self.Lambda_t_R, self.J_kt_R, self.Null_R, self.J_task_NC = [], [], [], [], [], []
        for i in range(L):
            Jt, kd = self.J_task[i], self.kind[i]
            t = Jt.shape[0]
            JtR = np.zeros((t, rs))
            lamR = np.zeros((t, t))
            JktR = np.zeros((rm, t))
            lam = None
            JtNC = None
            if kd["cmm"]:
                JtR[:, :vcd] = Jt[:, :vcd]
                JtR[:, vcd:] = Jt[:, vcd : vcd + ncd] @ self.J_I_nc_inv_T.T
                JktR, lamR = self._jkt_r(JtR)
                lam = lamR
            elif kd["reduced"] and not kd["noncont"]:
                JtR[:, :vcd] = Jt[:, :vcd]
                JktR, lamR = self._jkt_r(JtR)
                lam = lamR
            elif (not kd["reduced"]) and kd["noncont"]:
                lam = np.linalg.inv(Jt @ self.A_inv_N_C @ Jt.T)
                JtNC = Jt[:, vcd : vcd + ncd]
                JtR[:, :vcd] = Jt[:, :vcd]
                JtR[:, vcd:] = Jt[:, vcd : vcd + ncd] @ self.J_I_nc_inv_T.T
            else:
                raise ValueError("UNDEFINED TASK TYPE (task.cpp:134-141)")
            self.J_task_R.append(JtR)
            self.Lambda_t.append(lam)
            self.Lambda_t_R.append(lamR)
            self.J_kt_R.append(JktR)
            self.J_task_NC.append(JtNC)
            if i != L - 1:
                if not kd["noncont"]:
                    prev = np.eye(rm) if i == 0 else self.Null_R[i - 1]
                    self.Null_R.append(prev @ (np.eye(rm) - JktR @ lamR @ JtR @ self.A_R_inv_N_CR[:, 6:]))
                else:
                    self.Null_R.append(self.Null_R[i - 1])

    def cone_matrix(self):
        return Cycle.cone_matrix(self)

    def _qp_r(self, Ntorque_task, fvec, tau_prev):
        """rows of CalcSingleTaskTorqueWithQP_R / _R_NC without torque limit (dwbc.cpp:3524-3559, 3682-3717)"""
        Ct = self.cone_matrix()
        Atemp = Ct @ self.J_CR_INV_T[:, 6:]
        k = max(self.cdof - 6, 0)
        t = Ntorque_task.shape[1]
        A = np.zeros((Ct.shape[0], t + k))
        A[:, :t] = -Atemp @ Ntorque_task
        if k:
            A[:, t:] = -Atemp @ self.NwJw_R
        bA = Ct @ self.P_CR - Atemp @ (tau_prev + Ntorque_task @ fvec)
        return A, -bA

    # -- ReducedCalcTaskControlTorque(hqp=true) (dwbc.cpp:3255-3446)
    def reduced_task_torque(self, fstars):
        rm, cod, ncd, vcd, m = self.r_model_dof, self.co_dof, self.nc_dof, self.vc_dof, self.m
        L = len(self.tasks)
        R0 = self.R[0]
        tau_task_R = np.zeros(rm)
        tau_task_NC = np.zeros(ncd)
        tau_task_R_qp = np.zeros(rm)
        first_nc = -1
        force_on_nc_r = np.zeros(6)
        self.fstar_qp = [None] * L
        self.contact_qp = [None] * L
        self.qp = []
        th_R = [np.zeros(rm) for _ in range(L)]
        tnull_R = [np.zeros(rm) for _ in range(L)]
        tnull_nc = [np.zeros(ncd) for _ in range(L)]
        t_nc = [np.zeros(ncd) for _ in range(L)]
        for i in range(L):
            kd, Jt, lam, f = self.kind[i], self.J_task[i], self.Lambda_t[i], np.asarray(fstars[i], float)
            if kd["noncont"]:
                temp = Jt.T @ (lam @ f)
                t_nc[i] = temp[vcd : vcd + ncd]
                fon = np.concatenate([temp[0:3], R0 @ temp[3:6]])
                force_on_nc_r = force_on_nc_r + fon
                th_R[i][:cod] = self.J_base_R_kt[:cod] @ fon
                th_R[i][cod : cod + 6] = self.J_I_nc_inv_T @ t_nc[i]
                if kd["nc_h"] == 0:
                    first_nc = i
                    tnull_R[i] = self.Null_R[i - 1] @ th_R[i]
                    tnull_nc[i] = t_nc[i]
                else:
                    Jp, lamp = self.J_task[i - 1], self.Lambda_t[i - 1]
                    null_force = lamp @ (Jp @ (self.A_inv_N_C @ (Jt.T @ (lam @ f))))
                    temp = Jp.T @ null_force
                    tnull_nc[i] = t_nc[i] - temp[vcd : vcd + ncd]
                    temp[3:6] = R0 @ temp[3:6]
                    nthr = np.zeros(rm)
                    nthr[:cod] = th_R[i][:cod] - self.J_base_R_kt[:cod] @ temp[0:6]
                    # ts_[i-1].J_task_NC_ is only assigned for non-contact tasks (task.cpp:128); after a contact-chain
                    # task the reference reads an unsized matrix.  Its definition is used for every previous level.
                    JpNC = Jp[:, vcd : vcd + ncd]
                    nthr[cod : cod + 6] = self.J_I_nc_inv_T @ (t_nc[i] - JpNC.T @ null_force)
                    tnull_R[i] = self.Null_R[i - 1] @ nthr
                    force_on_nc_r = force_on_nc_r - temp[0:6]
            else:
                Nprev = np.eye(rm) if i == 0 else self.Null_R[i - 1]
                Nt = Nprev @ (self.J_kt_R[i] @ lam)
                A, ub = self._qp_r(Nt, f, self.tau_grav_R + tau_task_R)
                t = Nt.shape[1]
                st, x, _act = solve_qp(A, ub, t, 300)  # SolveQPoases(300, ...) dwbc.cpp:3581
                self.qp.append((A, ub))
                if st == 0:
                    return 0
                self.fstar_qp[i] = x[:t]
                self.contact_qp[i] = x[t:]
                th_R[i] = self.J_kt_R[i] @ lam @ (f + x[:t])
                tnull_R[i] = Nprev @ th_R[i]
                tau_task_R = tau_task_R + tnull_R[i]
                tau_task_NC = tau_task_NC + tnull_nc[i]
        self.force_on_nc_r = force_on_nc_r
        if first_nc >= 0:
            Nt = self.Null_R[first_nc - 1] @ self.J_base_R_kt
            A, ub = self._qp_r(Nt, force_on_nc_r, self.tau_grav_R + tau_task_R)
            st, x, _act = solve_qp(A, ub, 6, 300)
            self.qp.append((A, ub))
            if st == 0:
                return 0
            self.force_on_nc_R_qp = x[:6]
            self.nc_qp_contact = x[6:]
            for i in range(first_nc, L):
                if self.kind[i]["noncont"]:
                    tau_task_R = tau_task_R + tnull_R[i]
                    tau_task_NC = tau_task_NC + tnull_nc[i]
            tau_task_R_qp = np.zeros(rm)
            tau_task_R_qp[:cod] = self.J_base_R_kt[:cod] @ self.force_on_nc_R_qp
        self.tau_task_R, self.tau_task_NC = tau_task_R, tau_task_NC
        tt = np.zeros(m)
        tt[:cod] = tau_task_R[:cod] + tau_task_R_qp[:cod]
        tt[cod : cod + ncd] = self.J_I_nc.T @ tau_task_R[cod : cod + 6] + self.N_I_nc @ tau_task_NC
        self.tau_task = tt
        return 1

    # -- ReducedCalcContactRedistribute(hqp=true) -> CalcContactRedistributeR (dwbc.cpp:3758-3770, 4776-4941)
    def reduced_contact_redistribute(self):
        m, rm = self.m, self.r_model_dof
        cd = self.cdof
        k = cd - 6
        self.tau_contact = np.zeros(m)
        if cd <= 6:
            return 0  # `ret = 0` and nothing else happens (dwbc.cpp:3760-3769)
        tau_in = self.tau_grav_R + self.tau_task_R
        nc = len(self.act_contacts)
        crot = np.zeros((cd, cd))
        RotW = np.eye(cd)
        for i in range(nc):
            Rt = self.c_rot[i].T  # cm = I (dwbc.cpp:4829)
            crot[6 * i : 6 * i + 3, 6 * i : 6 * i + 3] = Rt
            crot[6 * i + 3 : 6 * i + 6, 6 * i + 3 : 6 * i + 6] = Rt
            RotW[6 * i + 2, 6 * i + 2] = 0.0
        Jb = self.J_CR_INV_T[:, 6:]
        H_temp = RotW @ crot @ Jb @ self.NwJw_R
        H = H_temp.T @ H_temp
        g = (RotW @ crot @ (Jb @ tau_in - self.P_CR)) @ H_temp
        Ct = self.cone_matrix()
        Atemp = Ct @ Jb
        bA = Ct @ self.P_CR - Atemp @ tau_in
        A = -Atemp @ self.NwJw_R
        ub = -bA
        # strictly convex QP  min 1/2 c^T H c + g^T c  s.t. A c <= ub : unique solution.  c = T y + c0 with H = L L^T,
        # T = L^-T, c0 = -H^-1 g turns it into the least-distance form min 1/2 |y|^2 s.t. (A T) y <= ub - A c0.
        Lc = np.linalg.cholesky(H)
        T = np.linalg.inv(Lc).T
        c0 = -np.linalg.solve(H, g)
        st, y, _act = solve_qp(A @ T, ub - A @ c0, k, 600)  # SolveQPoases(600, ...) dwbc.cpp:4921
        self.redis_qp = (H, g, A, ub)
        if st == 0:
            return 0
        c = T @ y + c0
        self.cf_redis = c
        tcR = self.NwJw_R @ c
        self.tau_contact_R = tcR
        self.tau_contact[:cd] = tcR[:cd]  # torque_contact_.segment(0, contact_dof_) (dwbc.cpp:3766)
        return 1

    def contact_force(self, tau):
        return self.J_C_INV_T[:, 6:] @ tau - self.J_C_INV_T @ self.G  # getContactForce with the full-model P_C

    def run_reduced(self, q, flags, fstars):
        self.update_kinematics(q)
        self.set_contact(flags)
        self.reduced_dynamics()
        ok = self.reduced_contact_constraint()
        self.reduced_grav()
        self.reduced_task_space()
        ok_t = self.reduced_task_torque(fstars)
        ok_c = self.reduced_contact_redistribute() if ok_t else 0
        if self.cdof <= 6:
            ok_c = 1  # single support: the reference returns 0 from a no-op; nothing failed
        self.status = int(bool(ok) and bool(ok_t) and bool(ok_c))
        return self.tau_grav + self.tau_task + self.tau_contact


def point_jac_base(cyc):
    """link_[0].jac_ (src/link.cpp:98-103): Jacobian of the base link origin, rows [lin; ang]"""
    return D.point_jacobian(cyc.model, cyc.R, cyc.p, 0, np.zeros(3))


def _cod_q(M):
    """householderQ() of the column-pivoted QR inside Eigen's COD (wbd.cpp:49)"""
    import scipy.linalg as sla

    Q, _R, _p = sla.qr(M, pivoting=True)
    return Q

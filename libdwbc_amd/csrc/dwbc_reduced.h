// dwbc_reduced.h -- fused control cycle through libdwbc's REDUCED (centroidal) dynamics model, one wavefront per
// robot instance (BASELINE config 5).
//
// The contact chains (floating base + the legs that are in contact: vc_dof = 6 + co_dof coordinates) keep their
// joint coordinates; everything else (the "non-contact" bodies: waist, arms, head and a swing leg) is replaced by the
// 6-D centroidal motion of that body group, J_I_nc (6 x nc_dof).  The reduced system has RS = vc_dof + 6 coordinates
// (24 for TOCABI double support instead of 39).
//
// Reference functions restated (file:line in the reference tree):
//   RobotData::ReducedDynamicsCalculate        src/dwbc.cpp:2752-2990
//   RobotData::ReducedCalcContactConstraint    src/dwbc.cpp:3077-3142
//   RobotData::ReducedCalcGravCompensation     src/dwbc.cpp:3144-3150
//   RobotData::ReducedCalcTaskSpace            src/dwbc.cpp:3152-3253, TaskSpace::CalcJKT_R src/task.cpp:95-142,
//                                              CalculateJKT_R src/wbd.cpp:220-226
//   RobotData::ReducedCalcTaskControlTorque    src/dwbc.cpp:3255-3446 (hqp = true)
//   RobotData::CalcSingleTaskTorqueWithQP_R    src/dwbc.cpp:3448-3599
//   RobotData::CalcSingleTaskTorqueWithQP_R_NC src/dwbc.cpp:3601-3756
//   RobotData::ReducedCalcContactRedistribute  src/dwbc.cpp:3758-3770 -> CalcContactRedistributeR :4776-4941 (hqp)
//
// Algebra used instead of the reference's block copies (same matrices):
//   A_R_inv      = J_R A^-1 J_R^T            (the reference's "simplified calculation", dwbc.cpp:2932-2956)
//   A_R_inv N_CR = J_R (A^-1 N_c) J_R^T      (J_C is zero outside the vc columns, so J_CR A_R_inv = (J_C A^-1) J_R^T)
//   both through reduce_to_R() on the column-per-lane registers of A^-1 / A^-1 N_c;
//   null(W_R) = the internal-wrench basis of the full model restricted to the contact-chain joints (rows of the
//   virtual centroidal coordinates are zero), so NwJw_R / W_R^+ use the same closed forms as dwbc_cycle2.h.
//
// Scope (device sets status 0 otherwise): 1 or 2 active CONTACT_6D contacts whose chains occupy the leading joint
// dofs (TOCABI: L+R or L), every task level made of links that are all on / all off the contact chains, level 0 on
// the chains, no COM ("cmm") task, no torque limit (reference App. C-7/C-10).
#pragma once
#include "dwbc_cycle2.h"

namespace dwbc {

template <int N, int NB, int NLV>
struct LdsR : Lds2<N, NB, NLV> {
    using B2 = Lds2<N, NB, NLV>;
    static constexpr int RSX = 24, RMX = 18, NCX = N - 12, T = kMaxTaskDof, C = 6 * kMaxActiveContacts;
    // regions of the full-model map that the reduced cycle does not use hold reduced data:
    //   [NwJw, Rw) (NwJw, FNl, U, Xl, T1r of dwbc_cycle2.h; free once the staged mass matrix is dead, i.e. after J_I_nc)
    static constexpr int rblk = RSX * RSX + C * RSX + RMX * 6 + C * 6;
    static constexpr int xb = (B2::total + 1) & ~1;
    static constexpr bool reuse = (B2::Rw - ((B2::NwJw + 1) & ~1)) >= rblk;
    static constexpr int AR = reuse ? ((B2::NwJw + 1) & ~1) : xb;  // RSX x RSX A_R_inv, later A_R_inv N_CR
    static constexpr int JbR = AR + RSX * RSX;        // C x RSX   J_CR_INV_T
    static constexpr int NwR = JbR + C * RSX;         // RMX x 6   NwJw_R
    static constexpr int FNR = NwR + RMX * 6;         // C x 6     A_rot J̄_R[:,6:] NwJw_R
    //   c_Y .. (dead after stage 1; the full model's c_Q / c_QW / c_Pi / c_Z slots of the task-space phase)
    static constexpr int JR = (B2::c_Y + 1) & ~1;     // T x RSX
    static constexpr int T1R = JR + T * RSX;          // T x RSX
    static constexpr int QRr = T1R + T * RSX;         // T x RMX
    static_assert(QRr + T * RMX <= B2::c_s1, "jkt scratch must stay below the small-inverse scratch");
    static constexpr int x0 = reuse ? xb : xb + rblk;
    static constexpr int JIt = x0;                    // NCX x 6   J_I_nc transposed
    static constexpr int JIiT = JIt + NCX * 6;        // 6 x NCX   J_I_nc_inv_T
    static constexpr int Bmt = JIiT + 6 * NCX;        // N x 6
    static constexpr int tmpE = Bmt + N * 6;          // N x 6
    static constexpr int GR = tmpE + N * 6;           // RSX
    static constexpr int PCR = GR + RSX;              // C
    static constexpr int Jbk = PCR + C;               // RMX x T   J_base_R_kt_
    static constexpr int UR = Jbk + RMX * T;          // NLV x RMX x T   Null_{l-1} J_kt_R Lambda
    static constexpr int XR = UR + NLV * RMX * T;     // NLV x RMX x T   J_kt_R Lambda
    static constexpr int YR = XR + NLV * RMX * T;     // NLV x T x RMX   (J_task_R A_R_inv N_CR)[:,6:]
    static constexpr int UNC = YR + NLV * T * RMX;    // RMX x T   Null J_base_R_kt_
    static constexpr int MT = UNC + RMX * T;          // (NLV-1) x N x T   J_task^T Lambda of non-contact level l at slot l-1
    // later non-contact levels need >= 3 levels (level 0 is a contact-chain task, one level is the first non-contact one)
    static constexpr int NPN = NLV >= 3 ? NLV : 0;
    static constexpr int PN = MT + (NLV - 1) * N * T;       // NPN x N x T   J_{l-1}^T Lambda_{l-1} J_{l-1} A^-1 N_c J_l^T Lambda_l (dwbc.cpp:3313-3316)
    static constexpr int JttP = PN + NPN * N * T;     // N x T     previous level's J_task^T
    static constexpr int T1P = JttP + (NPN ? N * T : 0);  // T x N     previous level's J_task A^-1 N_c
    static constexpr int QWR = T1P + (NPN ? T * N : 0);   // T x RMX
    static constexpr int sm = QWR + T * RMX;          // 6 small 6x6 blocks
    static constexpr int ARrow = sm;                  // 6 x RSX   rows vc_dof.. of A_R (reduced-dynamics phase only)
    static constexpr int tgR = sm + 6 * 36;           // RMX
    static constexpr int ttR = tgR + RMX;             // RMX
    static constexpr int sumR = ttR + RMX;            // RMX
    static constexpr int thR = sumR + RMX;            // RMX
    static constexpr int tRqp = thR + RMX;            // RMX
    static constexpr int tcR = tRqp + RMX;            // RMX
    static constexpr int tNC = tcR + RMX;             // NCX
    static constexpr int vecR = tNC + NCX;            // RSX
    static constexpr int v6 = vecR + RSX;             // 4 x 6 small vectors
    static constexpr int Jcm = v6 + 24;               // 6 N + 3: Jacobian of the COM link + com_pos (the U block holds A_R here)
    static constexpr int rtotal = Jcm + 6 * N + 4;
    static constexpr int total_bytes = rtotal * (int)sizeof(real_t) + 64;
};

// Out (RS x RS, row stride RSX) = J_R S J_R^T for the symmetric N x N matrix whose column `lane` is s.
// J_R = [I_vc 0; 0 J_I_nc] (reference src/dwbc.cpp:2918-2930); JIt is J_I_nc transposed (ncd x 6).
template <int N, int NT, int RSX>
DWBC_DEV void reduce_to_R(Thr th, PLA_REF(real_t, s, N), const real_t *JIt, int vcd, real_t *Out, real_t *tmpE) {
    DWBC_LANE_DECL;
    const int ncd = N - vcd;
    LANES {
        real_t e[6] = {real_t(0.0), real_t(0.0), real_t(0.0), real_t(0.0), real_t(0.0), real_t(0.0)};
#pragma unroll
        for (int i = 12; i < N; i++) {
            if (i >= vcd) {
#pragma unroll
                for (int r = 0; r < 6; r++) e[r] += LV(s)[i] * JIt[(i - vcd) * 6 + r];
            }
        }
        if (lane < N) {
#pragma unroll
            for (int r = 0; r < 6; r++) tmpE[lane * 6 + r] = e[r];
        }
        if (lane < vcd) {
#pragma unroll
            for (int a = 0; a < 18; a++)
                if (a < vcd) Out[a * RSX + lane] = LV(s)[a];
        }
    }
    DWBC_SYNC();
    for (int idx = th.tid; idx < vcd * 6; idx += NT) {
        const int j = idx / 6, r = idx - j * 6;
        const real_t v = tmpE[j * 6 + r];
        Out[(vcd + r) * RSX + j] = v;
        Out[j * RSX + vcd + r] = v;
    }
    for (int idx = th.tid; idx < 36; idx += NT) {
        const int r = idx / 6, r2 = idx - r * 6;
        real_t acc = real_t(0.0);
        _Pragma("unroll 8")
        for (int i = 0; i < ncd; i++) acc += tmpE[(vcd + i) * 6 + r] * JIt[i * 6 + r2];
        Out[(vcd + r) * RSX + vcd + r2] = acc;
    }
    DWBC_SYNC();
}

// Cholesky of a 6x6 SPD matrix on uniform data: Hinv = H^-1 and Tm = L^-T (so Tm Tm^T = H^-1), both 6x6 row-major
DWBC_WDEV int chol6_hinv_T(const real_t *Ain, real_t *Hinv, real_t *Tm) {
    DWBC_LANE_DECL;
    real_t Lc[6][6], ri[6];
    int ok = 1;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) Lc[i][j] = Ain[i * 6 + j];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        real_t d = Lc[j][j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= Lc[j][k] * Lc[j][k];
        if (!(d > real_t(0.0))) { ok = 0; d = real_t(1.0); }
        ri[j] = fast_rsqrt(d);
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            real_t v = Lc[i][j];
#pragma unroll
            for (int k = 0; k < j; k++) v -= Lc[i][k] * Lc[j][k];
            Lc[i][j] = v * ri[j];
        }
    }
    DWBC_SYNC();
    LANES {
        real_t y[6], z[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            real_t v = (lane == i) ? real_t(1.0) : real_t(0.0);
#pragma unroll
            for (int k = 0; k < i; k++) v -= Lc[i][k] * y[k];
            y[i] = v * ri[i];
        }
#pragma unroll
        for (int i = 5; i >= 0; i--) {
            real_t v = y[i], u = (lane == i) ? real_t(1.0) : real_t(0.0);
#pragma unroll
            for (int k = i + 1; k < 6; k++) { v -= Lc[k][i] * y[k]; u -= Lc[k][i] * z[k]; }
            y[i] = v * ri[i];
            z[i] = u * ri[i];
        }
        if (lane < 6) {
#pragma unroll
            for (int i = 0; i < 6; i++) { Hinv[i * 6 + lane] = y[i]; Tm[i * 6 + lane] = z[i]; }
        }
    }
    DWBC_SYNC();
    return ok;
}

// CalculateJKT_R (reference src/wbd.cpp:220-226) for the task Jacobian JRm (t x RS, row stride RSX) in LDS:
//   Lam = (J A_R_inv N_CR J^T)^-1 (t x t, row stride t), Y = (J A_R_inv N_CR)[:,6:] (t x RM, stride RMX),
//   Jkt = W_R^+ Q^T (Q W_R^+ Q^T)^-1 with Q = Lam Y (RM x t, stride T), X = Jkt Lam (stride T).  Returns 0 if a block is
//   not positive definite.
template <int N, int NB, int NLV, int NT>
DWBC_DEV int jkt_reduced(Thr th, real_t *L, PLA_REF(real_t, w, 18), int t, int RS, int RM, real_t *Lam, real_t *Yo, real_t *Jkt,
                         real_t *Xo) {
    using S = LdsR<N, NB, NLV>;
    constexpr int RSX = S::RSX, RMX = S::RMX, T = S::T;
    DWBC_LANE_DECL;
    const real_t *JRm = L + S::JR, *AR = L + S::AR;
    real_t *T1R = L + S::T1R, *Q = L + S::QRr, *QW = L + S::QWR, *s2 = L + S::sm, *Pi = L + S::sm + 36;
    int ok = 1;
    DWBC_SYNC();
    for (int idx = th.tid; idx < t * RS; idx += NT) {
        const int r = idx / RS, b = idx - r * RS;
        real_t acc = real_t(0.0);
        _Pragma("unroll 8")
        for (int a = 0; a < RS; a++) acc += JRm[r * RSX + a] * AR[a * RSX + b];
        T1R[r * RSX + b] = acc;
    }
    DWBC_SYNC();
    for (int idx = th.tid; idx < t * t; idx += NT) {
        const int i = idx / t, j = idx - i * t;
        real_t acc = real_t(0.0);
        _Pragma("unroll 8")
        for (int a = 0; a < RS; a++) acc += T1R[i * RSX + a] * JRm[j * RSX + a];
        s2[idx] = acc;
    }
    if (!spd_inverse_small(s2, t, t, Lam, t, nullptr)) ok = 0;
    DWBC_SYNC();
    for (int idx = th.tid; idx < T * RM; idx += NT) {
        const int i = idx / RM, c = idx - i * RM;
        real_t acc = real_t(0.0);
        if (i < t)
            _Pragma("unroll 8")
            for (int p = 0; p < t; p++) acc += Lam[i * t + p] * T1R[p * RSX + 6 + c];
        Q[i * RMX + c] = acc;
        if (i < t) Yo[i * RMX + c] = T1R[i * RSX + 6 + c];
    }
    DWBC_SYNC();
    for (int r = 0; r < T; r++) {
        LANES {
            real_t acc = real_t(0.0);
#pragma unroll
            for (int a = 0; a < 18; a++) acc += (a < RM ? Q[r * RMX + a] : real_t(0.0)) * LV(w)[a];
            if (lane < RM) QW[r * RMX + lane] = acc;
        }
    }
    DWBC_SYNC();
    for (int idx = th.tid; idx < t * t; idx += NT) {
        const int i = idx / t, j = idx - i * t;
        real_t acc = real_t(0.0);
        _Pragma("unroll 8")
        for (int a = 0; a < RM; a++) acc += QW[i * RMX + a] * Q[j * RMX + a];
        s2[idx] = acc;
    }
    if (!spd_inverse_small(s2, t, t, Pi, t, nullptr)) ok = 0;
    DWBC_SYNC();
    for (int idx = th.tid; idx < RM * T; idx += NT) {
        const int i = idx / T, r2 = idx - i * T;
        real_t acc = real_t(0.0);
        if (r2 < t)
            _Pragma("unroll 8")
            for (int r = 0; r < t; r++) acc += QW[r * RMX + i] * Pi[r * t + r2];
        Jkt[i * T + r2] = acc;
    }
    DWBC_SYNC();
    for (int idx = th.tid; idx < RM * T; idx += NT) {
        const int i = idx / T, r3 = idx - i * T;
        real_t acc = real_t(0.0);
        if (r3 < t)
            _Pragma("unroll 8")
            for (int r2 = 0; r2 < t; r2++) acc += Jkt[i * T + r2] * Lam[r2 * t + r3];
        Xo[i * T + r3] = acc;
    }
    DWBC_SYNC();
    return ok;
}

template <int N, int NB, int NLV, int NT, class Topo = TopoGeneric>
DWBC_DEV void cycle_instance_reduced(Thr th, const Setup &su, const BatchIO &io, int inst, real_t *L, int *iL) {
    using S = LdsR<N, NB, NLV>;
    constexpr int M = S::M, C = S::C, T = S::T, RSX = S::RSX, RMX = S::RMX, NCX = S::NCX;
    constexpr bool kExtras = true;
    DWBC_LANE_DECL;
    (void)iL;
    constexpr bool kTree = !std::is_same<Topo, TopoGeneric>::value;
    const int nb = kTree ? NB : su.nb;
    const real_t *body = io.body;
    const int *topo = io.topo;
    const io_t *qin = io.q + (size_t)inst * (N + 1);
    const DumpLayout dl = DumpLayout::make(N);
    real_t *dump = io.dump ? io.dump + (size_t)inst * dl.total : nullptr;
    int *diag = io.diag ? io.diag + (size_t)inst * DG_COUNT : nullptr;
    DWBC_STAMP_INIT();

    PLA(real_t, s, N);  // column `lane` of A -> A^-1 -> A^-1 N_c
    PL(real_t, dg);

    // contact chains of this instance: vc coordinates = base + chain joints (dwbc.cpp:2763-2796)
    const unsigned char *fl0 = io.flags + (size_t)inst * su.n_contacts;
    unsigned long long comask = 0x3full;
    int nact = 0;
    for (int i = 0; i < su.n_contacts; i++)
        if (fl0[i] && nact < kMaxActiveContacts) { comask |= su.c_dofmask[i]; nact++; }
    int vcd = 0;
    for (int j = 0; j < N; j++) vcd += (int)((comask >> j) & 1ull);
    int scope_ok = (nact >= 1) && (comask == ((1ull << vcd) - 1ull)) && (vcd == 12 || vcd == 18) && !su.has_tau_lim;
    if (!scope_ok) {  // outside the reduced path's scope: report failure, emit zeros
        io_t *tau = io.tau + (size_t)inst * 3 * M;
        for (int i = th.tid; i < 3 * M; i += NT) tau[i] = real_t(0.0);
        for (int i = th.tid; i < 12; i += NT) io.wrench[(size_t)inst * 12 + i] = real_t(0.0);
        if (th.tid == 0) io.status[inst] = 0;
        return;
    }
    const int RS = vcd + 6, RM = vcd, ncd = N - vcd, cod = vcd - 6;

#include "dwbc_cycle2_stage0.inc"

    // ================= ReducedDynamicsCalculate (dwbc.cpp:2752-2990) =================
    // J_I_nc = SI_nc_l^-1 cmm_nc: centroidal Jacobian of the non-contact bodies in the pelvis frame.  The composite
    // inertia of that body group = sum of the subtree composites (stage 0, world axes about the pelvis origin) of the
    // non-contact bodies whose parent is on a contact chain.
    real_t *JIt = L + S::JIt;
    {
        const real_t *Icm = L + S::k_Ic, *R0 = L + S::Rw, *As = L + S::k_A;
        real_t cmp[10];
        for (int c = 0; c < 10; c++) cmp[c] = real_t(0.0);
        for (int b = 1; b < nb; b++) {
            const int pb = su.parent[b];
            const bool bco = (comask >> (b + 5)) & 1ull;
            const bool pco = pb == 0 ? true : (bool)((comask >> (pb + 5)) & 1ull);
            if (!bco && pco)
                for (int c = 0; c < 10; c++) cmp[c] += Icm[b * 10 + c];
        }
        const real_t mnc = cmp[0], imn = real_t(1.0) / mnc;
        const real_t cw[3] = {cmp[1] * imn, cmp[2] * imn, cmp[3] * imn};
        const real_t Io[9] = {cmp[4], cmp[5], cmp[6], cmp[5], cmp[7], cmp[8], cmp[6], cmp[8], cmp[9]};
        const real_t cc = cw[0] * cw[0] + cw[1] * cw[1] + cw[2] * cw[2];
        real_t Icw[9], Il[9], cl[3];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) Icw[a * 3 + b] = Io[a * 3 + b] - mnc * ((a == b ? cc : real_t(0.0)) - cw[a] * cw[b]);
        for (int a = 0; a < 3; a++) cl[a] = R0[0 * 3 + a] * cw[0] + R0[1 * 3 + a] * cw[1] + R0[2 * 3 + a] * cw[2];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                real_t acc = real_t(0.0);
                for (int u = 0; u < 3; u++)
                    for (int v = 0; v < 3; v++) acc += R0[u * 3 + a] * Icw[u * 3 + v] * R0[v * 3 + b];
                Il[a * 3 + b] = acc;  // inertia_nc_ (math.cpp:309), pelvis axes about the group's COM
            }
        const real_t det = Il[0] * (Il[4] * Il[8] - Il[5] * Il[7]) - Il[1] * (Il[3] * Il[8] - Il[5] * Il[6]) + Il[2] * (Il[3] * Il[7] - Il[4] * Il[6]);
        const real_t id = real_t(1.0) / det;
        const real_t Ii[9] = {(Il[4] * Il[8] - Il[5] * Il[7]) * id, (Il[2] * Il[7] - Il[1] * Il[8]) * id, (Il[1] * Il[5] - Il[2] * Il[4]) * id,
                              (Il[5] * Il[6] - Il[3] * Il[8]) * id, (Il[0] * Il[8] - Il[2] * Il[6]) * id, (Il[2] * Il[3] - Il[0] * Il[5]) * id,
                              (Il[3] * Il[7] - Il[4] * Il[6]) * id, (Il[1] * Il[6] - Il[0] * Il[7]) * id, (Il[0] * Il[4] - Il[1] * Il[3]) * id};
        for (int i = th.tid; i < ncd; i += NT) {
            const int j = vcd + i;
            const real_t al[3] = {As[0 * N + j], As[1 * N + j], As[2 * N + j]};
            real_t top[3], bot[3];
            for (int a = 0; a < 3; a++) top[a] = R0[0 * 3 + a] * al[0] + R0[1 * 3 + a] * al[1] + R0[2 * 3 + a] * al[2];  // dwbc.cpp:2906
            bot[0] = As[3 * N + j] - (cl[1] * top[2] - cl[2] * top[1]);  // + skew(com_pos_nc_)^T top  (dwbc.cpp:2912-2913)
            bot[1] = As[4 * N + j] - (cl[2] * top[0] - cl[0] * top[2]);
            bot[2] = As[5 * N + j] - (cl[0] * top[1] - cl[1] * top[0]);
            for (int a = 0; a < 3; a++) {
                JIt[i * 6 + a] = top[a] * imn;
                JIt[i * 6 + 3 + a] = Ii[a * 3] * bot[0] + Ii[a * 3 + 1] * bot[1] + Ii[a * 3 + 2] * bot[2];
            }
        }
        DWBC_SYNC();
    }
    // A_R_inv = J_R A^-1 J_R^T (dwbc.cpp:2937-2956), A_R = its inverse (dwbc.cpp:2958)
    real_t *AR = L + S::AR;
    reduce_to_R<N, NT, RSX>(th, s, JIt, vcd, AR, L + S::tmpE);
    if (dump) {  // A_R_inv, J_I_nc for the reduced LQP / JACC configurators (dwbc.cpp:4504-4760) and the facade
        for (int idx = th.tid; idx < RSX * RSX; idx += NT) dump[dl.A_R_inv + idx] = (idx / RSX < RS && idx % RSX < RS) ? AR[idx] : real_t(0.0);
        for (int idx = th.tid; idx < 6 * NCX; idx += NT) {
            const int r6 = idx / NCX, i = idx - r6 * NCX;
            dump[dl.J_I_nc + idx] = i < ncd ? JIt[i * 6 + r6] : real_t(0.0);
        }
    }
    {
        PLA(real_t, r, RSX);
        PL(real_t, dr);
        LANES {
            const int col = lane < RS ? lane : 0;
#pragma unroll
            for (int a = 0; a < RSX; a++) LV(r)[a] = (lane < RS && a < RS) ? AR[a * RSX + col] : real_t(0.0);
            LV(dr) = (lane < RS) ? AR[col * RSX + col] : real_t(1.0);
        }
        if (!sweep_inverse_rl<RSX>(r, dr, RS)) st_contact = 0;
        if (dump) {
            LANES {
                if (lane < RSX) {
#pragma unroll
                    for (int a = 0; a < RSX; a++) dump[dl.A_R + a * RSX + lane] = (lane < RS && a < RS) ? LV(r)[a] : real_t(0.0);
                }
            }
        }
        real_t *ARrow = L + S::ARrow;
        LANES {
            if (lane >= vcd && lane < vcd + 6) {
#pragma unroll
                for (int a = 0; a < RSX; a++) ARrow[(lane - vcd) * RSX + a] = LV(r)[a];
            }
        }
        DWBC_SYNC();
        // J_I_nc_inv_T = A_R[vc:, :vc] A^-1[:vc, nc] + A_R[vc:, vc:] J_I_nc A^-1[nc, nc]  (dwbc.cpp:2971) = Bm A^-1[:, nc]
        real_t *Bmt = L + S::Bmt;
        for (int idx = th.tid; idx < N * 6; idx += NT) {
            const int i = idx / 6, r6 = idx - i * 6;
            real_t acc;
            if (i < vcd) acc = ARrow[r6 * RSX + i];
            else {
                acc = real_t(0.0);
                for (int r2 = 0; r2 < 6; r2++) acc += ARrow[r6 * RSX + vcd + r2] * JIt[(i - vcd) * 6 + r2];
            }
            Bmt[i * 6 + r6] = acc;
        }
        DWBC_SYNC();
        LANES {
            real_t acc[6] = {real_t(0.0), real_t(0.0), real_t(0.0), real_t(0.0), real_t(0.0), real_t(0.0)};
#pragma unroll
            for (int i = 0; i < N; i++)
#pragma unroll
                for (int r6 = 0; r6 < 6; r6++) acc[r6] += Bmt[i * 6 + r6] * LV(s)[i];
            if (lane >= vcd && lane < N) {
#pragma unroll
                for (int r6 = 0; r6 < 6; r6++) L[S::JIiT + r6 * NCX + (lane - vcd)] = acc[r6];
            }
        }
        DWBC_SYNC();
    }
    DWBC_STAMP(3);  // reduced dynamics: J_I_nc, A_R_inv, A_R, J_I_nc_inv_T
    const real_t *JIiT = L + S::JIiT;
    if (dump)
        for (int idx = th.tid; idx < 6 * NCX; idx += NT) dump[dl.J_I_nc_inv_T + idx] = (idx % NCX) < ncd ? JIiT[idx] : real_t(0.0);

#include "dwbc_cycle2_stage1.inc"

    // ================= ReducedCalcContactConstraint (dwbc.cpp:3077-3142) =================
    // A_R_inv N_CR = J_R (A^-1 N_c) J_R^T ; J_CR_INV_T = [J̄[:, :vc], J̄[:, nc] J_I_nc^T] ; G_R ; P_CR
    reduce_to_R<N, NT, RSX>(th, s, JIt, vcd, AR, L + S::tmpE);
    real_t *JbR = L + S::JbR, *GR = L + S::GR, *PCR = L + S::PCR;
    for (int idx = th.tid; idx < C * RS; idx += NT) {
        const int p = idx / RS, a = idx - p * RS;
        real_t acc = real_t(0.0);
        if (p < cd) {
            if (a < vcd) acc = JbT[p * N + a];
            else
                _Pragma("unroll 8")
                for (int i = 0; i < ncd; i++) acc += JbT[p * N + vcd + i] * JIt[i * 6 + (a - vcd)];
        }
        JbR[p * RSX + a] = acc;
    }
    for (int a = th.tid; a < RS; a += NT) {
        real_t acc;
        if (a < vcd) acc = L[S::G + a];
        else {
            acc = real_t(0.0);
            _Pragma("unroll 8")
            for (int i = 0; i < ncd; i++) acc += JIiT[(a - vcd) * NCX + i] * L[S::G + vcd + i];  // dwbc.cpp:2984
        }
        GR[a] = acc;
        if (dump) dump[dl.G_R + a] = acc;
    }
    DWBC_SYNC();
    for (int p = th.tid; p < C; p += NT) {
        real_t acc = real_t(0.0);
        if (p < cd)
            _Pragma("unroll 8")
            for (int a = 0; a < RS; a++) acc += JbR[p * RSX + a] * GR[a];
        PCR[p] = acc;  // P_CR (dwbc.cpp:3149)
    }
    for (int a = th.tid; a < RM; a += NT) {
        real_t acc = real_t(0.0);
        _Pragma("unroll 8")
        for (int b = 0; b < RS; b++) acc += AR[(6 + a) * RSX + b] * GR[b];
        L[S::vecR + a] = acc;  // (A_R_inv N_CR)[6:, :] G_R  (dwbc.cpp:3146)
    }
    DWBC_SYNC();
    DWBC_STAMP(4);  // A_R_inv N_CR, J_CR_INV_T, G_R, P_CR
    // ---- NwJw_R and the projector on null(W_R): internal-wrench basis restricted to the chain joints
    real_t *Vb = L + S::c_Vb, *VG = L + S::c_VG, *NwR = L + S::NwR, *FNR = L + S::FNR;
    if (k > 0) {
        const real_t *Pc = L + S::Pc;
        for (int idx = th.tid; idx < RM * k; idx += NT) {
            const int r = idx / 6, a = idx - r * 6;  // k == 6 here
            const int ci = 1 + a / 6, e = a % 6;
            real_t f2[3] = {0, 0, 0}, m2[3] = {0, 0, 0};
            if (e < 3) f2[e] = real_t(1.0); else m2[e - 3] = real_t(1.0);
            const real_t d0 = Pc[ci * 3] - Pc[0], d1 = Pc[ci * 3 + 1] - Pc[1], d2 = Pc[ci * 3 + 2] - Pc[2];
            const real_t m1x = -m2[0] - (d1 * f2[2] - d2 * f2[1]);
            const real_t m1y = -m2[1] - (d2 * f2[0] - d0 * f2[2]);
            const real_t m1z = -m2[2] - (d0 * f2[1] - d1 * f2[0]);
            real_t acc = real_t(0.0);
            if (r < cod) {  // the six virtual centroidal coordinates carry no contact Jacobian
                const real_t *Jc = JCt + (6 + r) * C;
                const int o1 = 6 * ci;
                acc = -f2[0] * Jc[0] - f2[1] * Jc[1] - f2[2] * Jc[2];
                acc += m1x * Jc[3] + m1y * Jc[4] + m1z * Jc[5];
                acc += f2[0] * Jc[o1 + 0] + f2[1] * Jc[o1 + 1] + f2[2] * Jc[o1 + 2];
                acc += m2[0] * Jc[o1 + 3] + m2[1] * Jc[o1 + 4] + m2[2] * Jc[o1 + 5];
            }
            Vb[idx] = acc;
        }
        DWBC_SYNC();
        real_t *JV = L + S::sm, *Gi = L + S::sm + 36, *Bm = L + S::sm + 72, *Sm6 = L + S::sm + 108;
        for (int idx = th.tid; idx < k * k; idx += NT) {
            const int i = idx / 6, j = idx - i * 6;
            real_t acc = real_t(0.0);
            _Pragma("unroll 8")
            for (int c = 0; c < RM; c++) acc += JbR[i * RSX + 6 + c] * Vb[c * k + j];
            JV[idx] = acc;
        }
        mm_tn<NT>(th, Gi, k, Vb, k, Vb, k, k, RM, k);
        DWBC_SYNC();
        spd_inverse_small(Gi, k, k, Gi, k, nullptr);
        mm_nn<NT>(th, VG, k, Vb, k, Gi, k, RM, k, k);
        mm_nn<NT>(th, Bm, k, JV, k, Gi, k, k, k, k);
        DWBC_SYNC();
        mm_nt<NT>(th, Sm6, k, Bm, k, JV, k, k, k, k);
        DWBC_SYNC();
        if (!spd_inverse_small(Sm6, k, k, Sm6, k, nullptr)) st_contact = 0;
        mm_nn<NT>(th, Bm, k, Sm6, k, JV, k, k, k, k);
        DWBC_SYNC();
        mm_nt<NT>(th, NwR, k, VG, k, Bm, k, RM, k, k);  // NwJw_R (dwbc.cpp:3123)
        DWBC_SYNC();
        real_t *s1 = L + S::tmpE;  // C x k scratch (the QP scratch would overlay Vb / VG, still needed for W_R^+)
        for (int idx = th.tid; idx < cd * k; idx += NT) {
            const int i = idx / 6, j = idx - i * 6;
            real_t acc = real_t(0.0);
            _Pragma("unroll 8")
            for (int c = 0; c < RM; c++) acc += JbR[i * RSX + 6 + c] * NwR[c * k + j];
            s1[idx] = acc;
        }
        DWBC_SYNC();
        for (int idx = th.tid; idx < cd * k; idx += NT) {
            const int i = idx / 6, j = idx - i * 6;
            const int a = i / 6, h = (i % 6) / 3, x = i % 3;
            const real_t *R = L + S::Rc + a * 9;
            const real_t *src = s1 + (6 * a + 3 * h) * k + j;
            FNR[idx] = R[0 * 3 + x] * src[0] + R[1 * 3 + x] * src[k] + R[2 * 3 + x] * src[2 * k];
        }
        DWBC_SYNC();
    }
    DWBC_STAMP(5);  // NwJw_R
    // ---- W_R^+ = (W_R + alpha P)^-1 - P / alpha, column per lane; torque_grav_R_ (dwbc.cpp:3146)
    PLA(real_t, w, 18);
    PL(real_t, dw);
    static_assert(RMX == 18, "w is sized for 18 reduced joints");
    LANES {
        const int col = lane < RM ? lane : 0;
#pragma unroll
        for (int a = 0; a < RMX; a++) LV(w)[a] = (lane < RM && a < RM) ? AR[(6 + a) * RSX + 6 + col] : real_t(0.0);
        LV(dw) = (lane < RM) ? AR[(6 + col) * RSX + 6 + col] : real_t(1.0);
    }
    real_t alpha = real_t(0.0);
    for (int i = 0; i < RM; i++) alpha += AR[(6 + i) * RSX + 6 + i];
    alpha /= RM;
    const real_t ialpha = alpha != real_t(0.0) ? real_t(1.0) / alpha : real_t(0.0);
    PLA(real_t, vbr, 6);
    LANES {
#pragma unroll
        for (int a = 0; a < 6; a++) LV(vbr)[a] = (k > 0 && lane < RM) ? Vb[lane * k + a] : real_t(0.0);
        if (k > 0 && lane < RM) {
            real_t dp = real_t(0.0);
#pragma unroll
            for (int i = 0; i < RMX; i++) {
                real_t pij = real_t(0.0);
                if (i < RM) {
#pragma unroll
                    for (int a = 0; a < 6; a++) pij += VG[i * k + a] * LV(vbr)[a];
                }
                LV(w)[i] += alpha * pij;
                dp = (i == lane) ? pij : dp;
            }
            LV(dw) += alpha * dp;
        }
    }
    if (!sweep_inverse_rl<RMX>(w, dw, RM)) st_contact = 0;
    LANES {
        if (k > 0 && lane < RM) {
#pragma unroll
            for (int i = 0; i < RMX; i++) {
                real_t pij = real_t(0.0);
                if (i < RM) {
#pragma unroll
                    for (int a = 0; a < 6; a++) pij += VG[i * k + a] * LV(vbr)[a];
                }
                LV(w)[i] -= ialpha * pij;
            }
        }
        real_t acc = real_t(0.0);
#pragma unroll
        for (int i = 0; i < RMX; i++) acc += LV(w)[i] * (i < RM ? L[S::vecR + i] : real_t(0.0));
        if (lane < RM) L[S::tgR + lane] = acc;
    }
    DWBC_SYNC();

    DWBC_STAMP(6);  // W_R^+ and gravity torque
    // ================= ReducedCalcTaskSpace (dwbc.cpp:3152-3253) =================
    int st_task = 1;
    real_t *JRm = L + S::JR, *Jbk = L + S::Jbk;
    {
        // J_base_R_ = link_[0].jac_.leftCols(RS): [I 0; 0 R_pelvis] on the base coordinates (dwbc.cpp:3159-3160)
        for (int idx = th.tid; idx < T * RSX; idx += NT) {
            const int r = idx / RSX, a = idx - r * RSX;
            real_t v = real_t(0.0);
            if (r < 3) v = (a == r) ? real_t(1.0) : real_t(0.0);
            else if (a >= 3 && a < 6) v = L[S::Rw + (r - 3) * 3 + (a - 3)];
            JRm[idx] = v;
        }
        if (!jkt_reduced<N, NB, NLV, NT>(th, L, w, 6, RS, RM, L + S::sm + 72, L + S::QWR /*Y scratch*/, Jbk, L + S::UNC /*X scratch*/))
            st_task = 0;
    }
    DWBC_STAMP(7);  // J_base_R_kt_
    int kind[NLV];      // 1 = contact-chain ("reduced") task, 2 = non-contact task, 0 = unsupported mix
    int first_nc = -1;
    for (int lv = 0; lv < su.n_levels; lv++) {
        int nco = 0, nnc = 0, ncm = 0;
        for (int li = 0; li < su.t_nlinks[lv]; li++) {
            const int link = su.t_link[lv][li];
            if (link == nb) { ncm++; continue; }  // the COM link: "cmm_task" (dwbc.cpp:3190-3195)
            const bool co = link == 0 || ((comask >> (link + 5)) & 1ull);
            if (co) nco++; else nnc++;
        }
        // 1 = contact-chain task, 2 = non-contact task, 3 = centroidal (COM) task: like 1 with the non-contact columns of the
        // Jacobian folded through J_I_nc_inv_T (task.cpp:106-114); mixed levels are undefined in the reference (task.cpp:134-141)
        // (TASK_CUSTOM levels are classified by the norms of their Jacobian blocks in the reference, dwbc.cpp:3170-3185: not built)
        kind[lv] = (ncm && !nco && !nnc) ? 3 : ((nco && !nnc && !ncm) ? 1 : ((nnc && !nco && !ncm) ? 2 : 0));
        if (kind[lv] == 2 && first_nc < 0) first_nc = lv;
        if (kind[lv] == 0 || (lv == 0 && kind[lv] == 2)) st_task = 0;
    }
    for (int lv = 0; lv < su.n_levels && st_task; lv++) {
        const int t = su.t_dof[lv];
        real_t *Jtt = L + S::c_Jt, *T1 = L + S::c_T1, *Lt = L + S::c_Lt + lv * T * T;
        DWBC_SYNC();
        for (int idx = th.tid; idx < T * N; idx += NT) Jtt[idx] = real_t(0.0);
        DWBC_SYNC();
        int row = 0;
        if (su.t_custom_slot[lv] >= 0 && io.custom_J) {  // TASK_CUSTOM: J_task handed over by SetTaskSpace(h, f*, J) (dwbc.cpp:664-681)
            const io_t *cj = io.custom_J + ((size_t)inst * su.n_custom + su.t_custom_slot[lv]) * (T * N);
            for (int idx = th.tid; idx < t * N; idx += NT) Jtt[(idx % N) * T + idx / N] = (real_t)cj[idx];
        }
        for (int li = 0; li < su.t_nlinks[lv]; li++) {
            const int mode = su.t_mode[lv][li], link = su.t_link[lv][li];
            real_t pl[3] = {0, 0, 0};
            if ((mode == TASK_LINK_6D_COM_FRAME || mode == TASK_LINK_POSITION_COM_FRAME) && link < nb)
                for (int a = 0; a < 3; a++) pl[a] = body[link * kBodyStride + BF_COM + a];
            else if (mode == TASK_LINK_6D_CUSTOM_FRAME || mode == TASK_LINK_POSITION_CUSTOM_FRAME)
                for (int a = 0; a < 3; a++) pl[a] = su.t_point[lv][li][a];
            const int rsel = mode <= TASK_LINK_6D_CUSTOM_FRAME ? 0 : (mode <= TASK_LINK_POSITION_CUSTOM_FRAME ? 1 : 2);
            if (link == nb) {  // the COM link: jac_ = jac_com_ (dwbc.cpp:352-353)
                com_task_rows<N, NT>(th, L + S::Jcm, Jtt, row, rsel, T);
            } else {
                const real_t *R = L + S::Rw + link * 9;
                real_t P[3];
                for (int a = 0; a < 3; a++) P[a] = L[S::pw + link * 3 + a] + R[a * 3] * pl[0] + R[a * 3 + 1] * pl[1] + R[a * 3 + 2] * pl[2];
                point_jacobian<N, NB, NT>(th, L + S::Rw, L + S::pw, L + S::aw, topo, nb, link, P, Jtt, 1, row, rsel == 0 ? 6 : 3, rsel, T);
            }
            row += rsel == 0 ? 6 : 3;
        }
        DWBC_SYNC();
        if (dump)
            for (int idx = th.tid; idx < t * N; idx += NT) dump[dl.J_task + lv * T * N + idx] = Jtt[(idx % N) * T + idx / N];
        if (kind[lv] != 2) {
            // J_task_R_ = [J_task[:, :vc], 0] (task.cpp:119) or, for a COM task, [J_task[:, :vc], J_task[:, nc] J_I_nc_inv_T^T]
            // (task.cpp:109-110); CalculateJKT_R, Null_task_R_ chain (dwbc.cpp:3236-3246)
            for (int idx = th.tid; idx < T * RSX; idx += NT) {
                const int r = idx / RSX, a = idx - r * RSX;
                real_t v = real_t(0.0);
                if (r < t && a < vcd) v = Jtt[a * T + r];
                else if (r < t && a < RS && kind[lv] == 3)
                    for (int i = 0; i < ncd; i++) v += Jtt[(vcd + i) * T + r] * JIiT[(a - vcd) * NCX + i];
                JRm[idx] = v;
            }
            real_t *Xs = L + S::XR + lv * RMX * T, *Ys = L + S::YR + lv * T * RMX, *Ul = L + S::UR + lv * RMX * T;
            if (!jkt_reduced<N, NB, NLV, NT>(th, L, w, t, RS, RM, Lt, Ys, L + S::Bmt /*J_kt scratch*/, Xs)) st_task = 0;
            for (int idx = th.tid; idx < RM * T; idx += NT) Ul[idx] = Xs[idx];
            DWBC_SYNC();
            for (int pl = lv - 1; pl >= 0; pl--) {  // U <- (I - X_pl Y_pl) U over the earlier contact-chain levels
                if (kind[pl] == 2) continue;
                const int tp = su.t_dof[pl];
                const real_t *Xp = L + S::XR + pl * RMX * T, *Yp = L + S::YR + pl * T * RMX;
                real_t *Z = L + S::sm;
                for (int idx = th.tid; idx < tp * t; idx += NT) {
                    const int i = idx / t, j = idx - i * t;
                    real_t acc = real_t(0.0);
                    _Pragma("unroll 8")
                    for (int c = 0; c < RM; c++) acc += Yp[i * RMX + c] * Ul[c * T + j];
                    Z[idx] = acc;
                }
                DWBC_SYNC();
                for (int idx = th.tid; idx < RM * t; idx += NT) {
                    const int i = idx / t, j = idx - i * t;
                    real_t acc = Ul[i * T + j];
                    _Pragma("unroll 8")
                    for (int p = 0; p < tp; p++) acc -= Xp[i * T + p] * Z[p * t + j];
                    Ul[i * T + j] = acc;
                }
                DWBC_SYNC();
            }
        } else {
            // non-contact task: Lambda_task_ = (J A^-1 N_c J^T)^-1 on the FULL model (task.cpp:126); MT = J^T Lambda
            LANES {
                real_t tc_[T];
#pragma unroll
                for (int r = 0; r < T; r++) tc_[r] = real_t(0.0);
#pragma unroll
                for (int i = 0; i < N; i++)
#pragma unroll
                    for (int r = 0; r < T; r++) tc_[r] += Jtt[i * T + r] * LV(s)[i];
#pragma unroll
                for (int r = 0; r < T; r++)
                    if (lane < N) T1[r * N + lane] = tc_[r];
            }
            DWBC_SYNC();
            for (int idx = th.tid; idx < t * t; idx += NT) {
                const int i = idx / t, j = idx - i * t;
                real_t acc = real_t(0.0);
                _Pragma("unroll 8")
                for (int c = 0; c < N; c++) acc += T1[i * N + c] * Jtt[c * T + j];
                L[S::sm + idx] = acc;
            }
            if (!spd_inverse_small(L + S::sm, t, t, Lt, t, nullptr)) st_task = 0;
            DWBC_SYNC();
            for (int idx = th.tid; idx < N * T; idx += NT) {
                const int j = idx / T, r2 = idx - j * T;
                real_t acc = real_t(0.0);
                if (r2 < t)
                    _Pragma("unroll 8")
                    for (int r = 0; r < t; r++) acc += Jtt[j * T + r] * Lt[r * t + r2];
                L[S::MT + (lv - 1) * N * T + idx] = acc;
            }
            DWBC_SYNC();
            if (lv != first_nc) {
                // later non-contact levels (nc_heirarchy_ >= 1): PN = J_p^T Lambda_p (J_p A^-1 N_c J^T) Lambda with p = lv - 1,
                // so that J_p^T null_force_ = PN f* (dwbc.cpp:3313-3316)
                const int tp = su.t_dof[lv - 1];
                const real_t *Ltp = L + S::c_Lt + (lv - 1) * T * T, *T1P = L + S::T1P, *JttP = L + S::JttP;
                real_t *B1 = L + S::sm, *B2 = L + S::sm + 36, *B3 = L + S::sm + 72;
                for (int idx = th.tid; idx < tp * t; idx += NT) {
                    const int i = idx / t, j = idx - i * t;
                    real_t acc = real_t(0.0);
                    _Pragma("unroll 8")
                    for (int c = 0; c < N; c++) acc += T1P[i * N + c] * Jtt[c * T + j];
                    B1[idx] = acc;
                }
                DWBC_SYNC();
                for (int idx = th.tid; idx < tp * t; idx += NT) {
                    const int i = idx / t, j = idx - i * t;
                    real_t acc = real_t(0.0);
                    _Pragma("unroll 8")
                    for (int a = 0; a < tp; a++) acc += Ltp[i * tp + a] * B1[a * t + j];
                    B2[idx] = acc;
                }
                DWBC_SYNC();
                for (int idx = th.tid; idx < tp * t; idx += NT) {
                    const int i = idx / t, j = idx - i * t;
                    real_t acc = real_t(0.0);
                    _Pragma("unroll 8")
                    for (int a = 0; a < t; a++) acc += B2[i * t + a] * Lt[a * t + j];
                    B3[idx] = acc;
                }
                DWBC_SYNC();
                for (int idx = th.tid; idx < N * T; idx += NT) {
                    const int j = idx / T, r = idx - j * T;
                    real_t acc = real_t(0.0);
                    if (r < t)
                        _Pragma("unroll 8")
                        for (int a = 0; a < tp; a++) acc += JttP[j * T + a] * B3[a * t + r];
                    L[S::PN + lv * N * T + idx] = acc;
                }
                DWBC_SYNC();
            }
        }
        // keep J^T and J A^-1 N_c of this level if the next one is a later non-contact level
        if (lv + 1 < su.n_levels && kind[lv + 1] == 2 && lv + 1 != first_nc) {
            if (kind[lv] != 2) {
                LANES {
                    real_t tc_[T];
#pragma unroll
                    for (int r = 0; r < T; r++) tc_[r] = real_t(0.0);
#pragma unroll
                    for (int i = 0; i < N; i++)
#pragma unroll
                        for (int r = 0; r < T; r++) tc_[r] += Jtt[i * T + r] * LV(s)[i];
#pragma unroll
                    for (int r = 0; r < T; r++)
                        if (lane < N) T1[r * N + lane] = tc_[r];
                }
                DWBC_SYNC();
            }
            for (int idx = th.tid; idx < N * T; idx += NT) { L[S::JttP + idx] = Jtt[idx]; L[S::T1P + idx] = T1[idx]; }
            DWBC_SYNC();
        }
    }
    DWBC_SYNC();

    DWBC_STAMP(8);  // task-space dynamics of every level
    // ================= ReducedCalcTaskControlTorque(hqp = true) (dwbc.cpp:3255-3446) =================
    const int ncone = 10 * nc;
    int st_redis = 1;
    const real_t *fs_in = L + S::fs;  // filled by task_reference() after stage 0
    real_t *base = L + S::t_base, *F = L + S::t_F, *fv = L + S::t_fv;
    real_t *tgR = L + S::tgR, *ttR = L + S::ttR, *sumR = L + S::sumR, *tNC = L + S::tNC, *tRqp = L + S::tRqp;
    real_t *fon = L + S::v6;  // force_on_nc_r_
    for (int i = th.tid; i < RMX; i += NT) { ttR[i] = real_t(0.0); sumR[i] = real_t(0.0); tRqp[i] = real_t(0.0); L[S::tcR + i] = real_t(0.0); }
    for (int i = th.tid; i < NCX; i += NT) tNC[i] = real_t(0.0);
    for (int i = th.tid; i < 6; i += NT) fon[i] = real_t(0.0);
    DWBC_SYNC();
    // passes 0..n_levels-1: task levels; pass n_levels: the force-on-non-contact QP (CalcSingleTaskTorqueWithQP_R_NC)
    for (int qi = 0; qi <= su.n_levels && st_task; qi++) {
        const bool is_level = qi < su.n_levels;
        if (!is_level && first_nc < 0) break;
        if (is_level && kind[qi] == 2) {
            // temp = J^T Lambda f* ; torque_nc_, force_on_nc_, torque_h_R_, torque_null_h_R_ (dwbc.cpp:3292-3310)
            const int t = su.t_dof[qi];
            const real_t *fs = fs_in + su.fstar_off[qi];
            real_t *tmpv = L + S::Bmt;  // N
            real_t *tmp2 = L + S::tmpE;  // N: J_p^T null_force_ for the later non-contact levels, else 0
            const bool later = qi != first_nc;
            for (int j = th.tid; j < N; j += NT) {
                real_t acc = real_t(0.0), acc2 = real_t(0.0);
                for (int r = 0; r < t; r++) {
                    acc += L[S::MT + (qi - 1) * N * T + j * T + r] * fs[r];
                    if (later) acc2 += L[S::PN + qi * N * T + j * T + r] * fs[r];
                }
                tmpv[j] = acc;
                tmp2[j] = acc2;
            }
            DWBC_SYNC();
            real_t *fo = L + S::v6 + 6;  // this task's force_on_nc_
            for (int a = th.tid; a < 6; a += NT) {
                real_t v;
                if (a < 3) v = tmpv[a];
                else v = L[S::Rw + (a - 3) * 3] * tmpv[3] + L[S::Rw + (a - 3) * 3 + 1] * tmpv[4] + L[S::Rw + (a - 3) * 3 + 2] * tmpv[5];
                // later levels: temp_torque_ = J_p^T null_force_, angular part rotated (dwbc.cpp:3316-3318); both the level's
                // force on the non-contact group and force_on_nc_r_ lose it (dwbc.cpp:3320,3324)
                real_t v2;
                if (a < 3) v2 = tmp2[a];
                else v2 = L[S::Rw + (a - 3) * 3] * tmp2[3] + L[S::Rw + (a - 3) * 3 + 1] * tmp2[4] + L[S::Rw + (a - 3) * 3 + 2] * tmp2[5];
                fo[a] = v - v2;
                fon[a] += v - v2;
            }
            DWBC_SYNC();
            real_t *thR = L + S::thR;
            for (int i = th.tid; i < RM; i += NT) {
                real_t acc = real_t(0.0);
                if (i < cod) { for (int a = 0; a < 6; a++) acc += Jbk[i * T + a] * fo[a]; }
                else { for (int c = 0; c < ncd; c++) acc += JIiT[(i - cod) * NCX + c] * (tmpv[vcd + c] - tmp2[vcd + c]); }
                thR[i] = acc;  // torque_h_R_ (first level) / null_torque_h_r (later levels, dwbc.cpp:3320-3321)
            }
            for (int i = th.tid; i < ncd; i += NT) tNC[i] += tmpv[vcd + i] - tmp2[vcd + i];  // torque_null_h_nc_ (dwbc.cpp:3309,3317)
            DWBC_SYNC();
            for (int pl = qi - 1; pl >= 0; pl--) {  // torque_null_h_R_ = Null_task_R_{qi-1} torque_h_R_
                if (kind[pl] == 2 || pl == su.n_levels - 1) continue;
                const int tp = su.t_dof[pl];
                const real_t *Xp = L + S::XR + pl * RMX * T, *Yp = L + S::YR + pl * T * RMX;
                real_t *Z = L + S::sm;
                for (int i = th.tid; i < tp; i += NT) {
                    real_t acc = real_t(0.0);
                    _Pragma("unroll 8")
                    for (int c = 0; c < RM; c++) acc += Yp[i * RMX + c] * thR[c];
                    Z[i] = acc;
                }
                DWBC_SYNC();
                for (int i = th.tid; i < RM; i += NT) {
                    real_t acc = thR[i];
                    _Pragma("unroll 8")
                    for (int p = 0; p < tp; p++) acc -= Xp[i * T + p] * Z[p];
                    thR[i] = acc;
                }
                DWBC_SYNC();
            }
            for (int i = th.tid; i < RM; i += NT) sumR[i] += thR[i];
            DWBC_SYNC();
            continue;
        }
        // QP over a torque map Ul (RM x t): a contact-chain task level, or Null J_base_R_kt_ for the non-contact force
        const int t = is_level ? su.t_dof[qi] : 6;
        const FastDiv fdt1(t + 1);
        real_t *Ul = is_level ? L + S::UR + qi * RMX * T : L + S::UNC;
        const real_t *fs = is_level ? fs_in + su.fstar_off[qi] : fon;
        if (!is_level) {
            // Ntorque_task = Null_task_R_{first_nc-1} J_base_R_kt_ (dwbc.cpp:3663)
            for (int idx = th.tid; idx < RM * T; idx += NT) Ul[idx] = Jbk[idx];
            DWBC_SYNC();
            for (int pl = first_nc - 1; pl >= 0; pl--) {
                if (kind[pl] == 2) continue;
                const int tp = su.t_dof[pl];
                const real_t *Xp = L + S::XR + pl * RMX * T, *Yp = L + S::YR + pl * T * RMX;
                real_t *Z = L + S::sm;
                for (int idx = th.tid; idx < tp * 6; idx += NT) {
                    const int i = idx / 6, j = idx - i * 6;
                    real_t acc = real_t(0.0);
                    _Pragma("unroll 8")
                    for (int c = 0; c < RM; c++) acc += Yp[i * RMX + c] * Ul[c * T + j];
                    Z[idx] = acc;
                }
                DWBC_SYNC();
                for (int idx = th.tid; idx < RM * 6; idx += NT) {
                    const int i = idx / 6, j = idx - i * 6;
                    real_t acc = Ul[i * T + j];
                    _Pragma("unroll 8")
                    for (int p = 0; p < tp; p++) acc -= Xp[i * T + p] * Z[p * 6 + j];
                    Ul[i * T + j] = acc;
                }
                DWBC_SYNC();
            }
        }
        DWBC_SYNC();
        for (int i = th.tid; i < RM; i += NT) {
            real_t acc = tgR[i] + ttR[i];
            _Pragma("unroll 8")
            for (int j = 0; j < t; j++) acc += Ul[i * T + j] * fs[j];
            base[i] = acc;
        }
        DWBC_SYNC();
        for (int idx = th.tid; idx < cd * (t + 1); idx += NT) {
            const int i = fdt1.div(idx), j = idx - i * (t + 1);
            real_t acc = real_t(0.0);
            if (j < t) {
                _Pragma("unroll 8")
                for (int c = 0; c < RM; c++) acc += JbR[i * RSX + 6 + c] * Ul[c * T + j];
            } else {
                _Pragma("unroll 8")
                for (int c = 0; c < RM; c++) acc += JbR[i * RSX + 6 + c] * base[c];
                acc -= PCR[i];
            }
            L[S::t_s1 + i * (T + 1) + j] = acc;
        }
        DWBC_SYNC();
        for (int idx = th.tid; idx < cd * (t + 1); idx += NT) {
            const int i = fdt1.div(idx), j = idx - i * (t + 1);
            const int a = i / 6, h = (i % 6) / 3, x = i % 3;
            const real_t *R = L + S::Rc + a * 9;
            const real_t *src = L + S::t_s1 + (6 * a + 3 * h) * (T + 1) + j;
            const real_t v = R[0 * 3 + x] * src[0] + R[1 * 3 + x] * src[T + 1] + R[2 * 3 + x] * src[2 * (T + 1)];
            if (j < t) F[i * kQpLd + j] = v; else fv[i] = v;
        }
        DWBC_SYNC();
        QpResult qres;
        qp_rows_and_solve<N, NB>(su, L, 0, ncone, act_c[0], act_c[1], Ul, T, t, NwR, k, k, kQpScaleGI, F, kQpLd, FNR, k, fv, base, t,
                                 300 /* SolveQPoases(300, ..) dwbc.cpp:3581,3739 */, qres, L + S::qp_V, L + S::qp_x);
        if (diag && th.tid == 0) {
            const int slot = is_level ? qi : kMaxLevels - 1;
            diag[DG_QP_ITER + slot] = qres.iters;
            diag[DG_QP_NACT + slot] = qres.nact;
            for (int a = 0; a < kQpLd; a++) diag[DG_QP_ACT + slot * kQpLd + a] = qres.act[a];
        }
        if (!qres.status) { st_task = 0; break; }
        const real_t *x = L + S::qp_x;
        if (is_level) {
            for (int i = th.tid; i < RM; i += NT) {
                real_t acc = real_t(0.0);
                _Pragma("unroll 8")
                for (int j = 0; j < t; j++) acc += Ul[i * T + j] * (fs[j] + x[j]);
                ttR[i] += acc;  // torque_task_R_ += Null J_kt_R Lambda (f* + f*_qp)  (dwbc.cpp:3346-3362)
            }
            if (dump)
                for (int j = th.tid; j < t; j += NT) dump[dl.fstar_qp + qi * T + j] = x[j];
        } else {
            for (int i = th.tid; i < RM; i += NT) {
                ttR[i] += sumR[i];  // the non-contact tasks' torque_null_h_R_ (dwbc.cpp:3430-3437)
                real_t acc = real_t(0.0);
                if (i < cod)
                    for (int a = 0; a < 6; a++) acc += Jbk[i * T + a] * x[a];
                tRqp[i] = acc;      // torque_task_R_qp = J_base_R_kt_.topRows(co_dof) force_on_nc_R_qp_ (dwbc.cpp:3439)
            }
            if (dump)
                for (int j = th.tid; j < 6; j += NT) dump[dl.fstar_qp + (kMaxLevels - 1) * T + j] = x[j];
        }
        DWBC_SYNC();
    }
    DWBC_SYNC();

    DWBC_STAMP(9);  // task cascade + non-contact force QP
    // ================= ReducedCalcContactRedistribute(hqp = true) (dwbc.cpp:3758-3770, 4776-4941) =================
    if (k > 0 && st_task) {
        // torque_input = torque_grav_R_ + torque_task_R_ ; wrench (contact frame) fvr = A_rot (J̄_R[:,6:] tau_in - P_CR)
        for (int i = th.tid; i < RM; i += NT) base[i] = tgR[i] + ttR[i];
        DWBC_SYNC();
        real_t *s1 = L + S::t_s1;
        for (int i = th.tid; i < cd; i += NT) {
            real_t acc = -PCR[i];
            _Pragma("unroll 8")
            for (int c = 0; c < RM; c++) acc += JbR[i * RSX + 6 + c] * base[c];
            s1[i] = acc;
        }
        DWBC_SYNC();
        for (int i = th.tid; i < cd; i += NT) {
            const int a = i / 6, h = (i % 6) / 3, x = i % 3;
            const real_t *R = L + S::Rc + a * 9;
            const real_t *src = s1 + 6 * a + 3 * h;
            fv[i] = R[0 * 3 + x] * src[0] + R[1 * 3 + x] * src[1] + R[2 * 3 + x] * src[2];
        }
        DWBC_SYNC();
        // H = H_temp^T H_temp, g = H_temp^T RotW fvr with H_temp = RotW FNR (RotW drops the normal-force rows)
        real_t *Hm = L + S::sm, *Hi = L + S::sm + 36, *Tm = L + S::sm + 72, *gv = L + S::v6 + 12, *c0 = L + S::v6 + 18;
        for (int idx = th.tid; idx < 36 + 6; idx += NT) {
            if (idx < 36) {
                const int i = idx / 6, j = idx - i * 6;
                real_t acc = real_t(0.0);
                for (int p = 0; p < cd; p++)
                    if (p % 6 != 2) acc += FNR[p * k + i] * FNR[p * k + j];
                Hm[idx] = acc;
            } else {
                const int i = idx - 36;
                real_t acc = real_t(0.0);
                for (int p = 0; p < cd; p++)
                    if (p % 6 != 2) acc += FNR[p * k + i] * fv[p];
                gv[i] = acc;
            }
        }
        DWBC_SYNC();
        const int hok = chol6_hinv_T(Hm, Hi, Tm);
        for (int i = th.tid; i < 6; i += NT) {
            real_t acc = real_t(0.0);
            for (int j = 0; j < 6; j++) acc -= Hi[i * 6 + j] * gv[j];
            c0[i] = acc;  // unconstrained minimiser -H^-1 g
        }
        DWBC_SYNC();
        // least-distance form in y (c = Tm y + c0): wrench map FNR Tm, wrench at y = 0: fv + FNR c0
        for (int idx = th.tid; idx < cd * 7; idx += NT) {
            const int i = idx / 7, j = idx - i * 7;
            real_t acc = real_t(0.0);
            if (j < 6) { for (int a = 0; a < 6; a++) acc += FNR[i * k + a] * Tm[a * 6 + j]; F[i * kQpLd + j] = acc; }
            else { acc = fv[i]; for (int a = 0; a < 6; a++) acc += FNR[i * k + a] * c0[a]; s1[i] = acc; }
        }
        DWBC_SYNC();
        for (int i = th.tid; i < cd; i += NT) fv[i] = s1[i];
        DWBC_SYNC();
        QpResult qres;
        qp_rows_and_solve<N, NB>(su, L, 0, ncone, act_c[0], act_c[1], NwR, k, k, NwR, k, 0, real_t(1.0), F, kQpLd, FNR, k, fv, base, k,
                                 600 /* SolveQPoases(600, ..) dwbc.cpp:4921 */, qres, L + S::qp_V, L + S::qp_x);
        if (diag && th.tid == 0) {
            diag[DG_QP_ITER + kMaxLevels] = qres.iters;
            diag[DG_QP_NACT + kMaxLevels] = qres.nact;
            for (int a = 0; a < kQpLd; a++) diag[DG_QP_ACT + kMaxLevels * kQpLd + a] = qres.act[a];
        }
        if (qres.status && hok) {
            const real_t *y = L + S::qp_x;
            real_t *cv = L + S::v6 + 6;
            for (int i = th.tid; i < 6; i += NT) {
                real_t acc = c0[i];
                for (int j = 0; j < 6; j++) acc += Tm[i * 6 + j] * y[j];
                cv[i] = acc;
                if (dump) dump[dl.cf_redis + i] = acc;
            }
            DWBC_SYNC();
            for (int i = th.tid; i < RM; i += NT) {
                real_t acc = real_t(0.0);
                _Pragma("unroll 8")
                for (int j = 0; j < k; j++) acc += NwR[i * k + j] * cv[j];
                L[S::tcR + i] = acc;  // torque_contact_R_ = NwJw_R qpres (dwbc.cpp:4923)
            }
        } else {
            st_redis = 0;
        }
        DWBC_SYNC();
    }

    DWBC_STAMP(10);  // redistribution QP
    // ================= outputs: torque_grav_ (dwbc.cpp:3147-3148), torque_task_ (:3442-3443), torque_contact_ (:3765-3766)
    {
        real_t *z6 = L + S::v6 + 18;
        for (int r = th.tid; r < 6; r += NT) {
            real_t acc = real_t(0.0);
            _Pragma("unroll 8")
            for (int i = 0; i < ncd; i++) acc += JIiT[r * NCX + i] * tNC[i];
            z6[r] = acc;
        }
        DWBC_SYNC();
        io_t *tau = io.tau + (size_t)inst * 3 * M;
        for (int i = th.tid; i < M; i += NT) {
            real_t g_, t_, c_;
            if (i < cod) {
                g_ = tgR[i];
                t_ = ttR[i] + tRqp[i];
            } else {
                const int c = i - cod;
                g_ = L[S::G + vcd + c];
                real_t acc = tNC[c];  // J_I_nc^T torque_task_R_[co:] + N_I_nc torque_task_NC_, N_I_nc = I - J_I_nc^T J_I_nc_inv_T
                for (int r = 0; r < 6; r++) acc += JIt[c * 6 + r] * (ttR[cod + r] - z6[r]);
                t_ = acc;
            }
            c_ = (i < cd && i < RM && k > 0) ? L[S::tcR + i] : real_t(0.0);
            if (!st_task) { t_ = real_t(0.0); c_ = real_t(0.0); }
            if (too_many) { g_ = real_t(0.0); t_ = real_t(0.0); c_ = real_t(0.0); }
            tau[i] = g_;
            tau[M + i] = t_;
            tau[2 * M + i] = c_;
            L[S::tg + i] = g_ + t_ + c_;
        }
        DWBC_SYNC();
        io_t *wr = io.wrench + (size_t)inst * 12;
        for (int i = th.tid; i < 12; i += NT) {
            real_t acc = real_t(0.0);
            if (i < cd && !too_many) {
                acc = -L[S::PC + i];
                for (int c = 0; c < M; c++) acc += JbT[i * N + 6 + c] * L[S::tg + c];
            }
            wr[i] = acc;  // getContactForce(tau_total) on the full-model J_C_INV_T (dwbc.cpp:3104) and P_C = J̄^T G
        }
        if (dump) { DWBC_SYNC(); dump_contacts_zmp(th, L + S::Pc, L + S::Rc, wr, nc, dump, dl); }
        if (th.tid == 0) {
            io.status[inst] = (st_contact && st_task && st_redis) ? 1 : 0;
            if (diag) {
                diag[DG_ST_CONTACT] = st_contact;
                diag[DG_ST_TASK] = st_task;
                diag[DG_ST_REDIS] = st_redis;
                diag[DG_FAIL_LEVEL] = -1;
            }
        }
    }
}

}  // namespace dwbc

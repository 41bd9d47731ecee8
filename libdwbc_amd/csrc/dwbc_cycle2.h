// dwbc_cycle2.h -- fused OSF/HQP control cycle, register-resident version (one wavefront = one robot instance).
//
// The three big symmetric matrices never live in LDS: lane j holds column j of A -> A^-1 -> A^-1 N_c (39 doubles) and later
// column j of W^+ (33 doubles) in registers.  Every large product is then "own column . uniform operand": one FMA per
// term, operands fetched with broadcast LDS reads (no bank conflicts, no index arithmetic), and the inverses are
// symmetric sweeps whose pivot column is broadcast with v_readlane (A^-1, tree-sparse) or through LDS (W^+).  LDS holds only
// the thin matrices (J_C, J̄_c^T, task blocks, QP inputs): 31.6 KB per instance for two task levels => 5 instances per CU.
//
// Reference functions restated: see the table at the top of dwbc_cycle.h (shared helpers, QP row assembly).
#pragma once
#include <type_traits>

#include "dwbc_cycle.h"
#include "dwbc_topo.h"
#include "dwbc_velocity.h"
#include "dwbc_fstar.h"
#include "dwbc_nohqp.h"

namespace dwbc {

template <int N, int NB, int NLV>
struct Lds2 {
    static constexpr int M = N - 6;
    static constexpr int C = 6 * kMaxActiveContacts;
    static constexpr int K = C - 6;
    static constexpr int T = kMaxTaskDof;
    static constexpr int max2(int a, int b) { return a > b ? a : b; }
    static constexpr int ev(int a) { return (a + 1) & ~1; }  // 16-byte alignment: lets row reads become ds_read_b128
    // ---- persistent
    static constexpr int q = 0;                        // N+1
    static constexpr int G = q + N + 1;
    static constexpr int tg = G + N;
    static constexpr int tt = tg + M;
    static constexpr int tc = tt + M;
    static constexpr int PC = tc + M;
    static constexpr int Rc = PC + C;
    static constexpr int Pc = Rc + kMaxActiveContacts * 9;
    static constexpr int fs = Pc + kMaxActiveContacts * 3;    // f* of every level (SetTaskSpace values or the on-device task reference)
    static constexpr int comp = fs + kMaxLevels * kMaxTaskDof; // com_pos (3 + pad), written for hqp = false only
    static constexpr int JbT = comp + 4;  // C x N            (written from stage 1 on)
    static constexpr int NwJw = JbT + C * N;                   // M x K
    static constexpr int FNl = NwJw + M * K;                   // C x K
    static constexpr int U = FNl + C * K;                      // levels x (M x T)
    static constexpr int Xl = U + NLV * M * T;                 // (levels-1) x (M x T)
    static constexpr int T1r = Xl + (NLV - 1) * M * T;         // (levels-1) x (T x M)
    static constexpr int Rw = T1r + (NLV - 1) * T * M;
    static constexpr int pw = Rw + NB * 9;
    static constexpr int aw = pw + NB * 3;
    static constexpr int tmp = aw + NB * 3;
    // ---- kinematics scratch.  The staged N x N mass matrix overlays JbT..T1r (not written before stage 1) if it fits.
    static constexpr int k_Rl = tmp;
    static constexpr int k_Iw = k_Rl + NB * 9;
    static constexpr int k_Ic = k_Iw + NB * 10;
    static constexpr int k_S = k_Ic + NB * 10;
    static constexpr int k_F = k_S + N * 6;
    static constexpr int k_col = k_F + N * 6;                  // pivot column of the sweep
    static constexpr bool a_overlay = (Rw - JbT) >= N * N;
    // larger models (N >= 42, kernel packs): the square no longer fits the hole, the lower triangle (row-packed) does -- without it
    // the map of a 43-dof model is 48 KB and only three workgroups share a CU
    static constexpr bool a_packed = !a_overlay && (Rw - JbT) >= N * (N + 1) / 2;
    static constexpr int k_A = (a_overlay || a_packed) ? JbT : k_col + N;
    static constexpr int k_end = (a_overlay || a_packed) ? k_col + N : k_col + N + N * N;
    // ---- contact / task-space scratch
    static constexpr int c_Vb = tmp;                           // M x K
    static constexpr int c_VG = c_Vb + M * K;                  // M x K
    static constexpr int c_vec = c_VG + M * K;                 // N
    static constexpr int c_Lt = ev(c_vec + N);                 // levels x T x T
    static constexpr int c_ov = c_Lt + NLV * T * T;            // two overlaid groups:
    //   (1) contact algebra
    static constexpr int c_JC = c_ov;
    static constexpr int c_Y = c_JC + C * N;
    static constexpr int c_s1 = c_Y + C * N;                   // C x K  (wrench-map scratch; later the diagonal of W)
    static constexpr int c_col = c_s1;                         // M  (W stage only)
    static constexpr int c_s2 = c_s1 + max2(C * K, M);         // C x C  (input of the small SPD inverses)
    static constexpr int c_Lam = c_s2;                         // Lambda_c is inverted in place
    static constexpr int c_end1 = c_s2 + C * C;
    //   (2) task-space dynamics (after NwJw / projector are done; keeps c_s1/c_s2 for its small inverses)
    static constexpr int c_Jt = c_ov;                          // T x N
    static constexpr int c_T1 = c_Jt + T * N;                  // T x N
    static constexpr int c_Q = c_T1 + T * N;                   // T x M
    static constexpr int c_QW = c_Q + T * M;                   // T x M
    static constexpr int c_Pi = c_QW + T * M;                  // T x T
    static constexpr int c_Z = c_Pi + T * T;                   // T x T
    static constexpr int c_end2 = c_Z + T * T;
    static_assert(c_end2 <= c_s1, "task-space scratch must not reach the small-inverse scratch it shares");
    static constexpr int c_end = max2(c_end1, c_end2);
    // ---- QP scratch (after the task-space phase)
    static constexpr int t_base = tmp;
    static constexpr int t_F = ev(t_base + M);                 // C x kQpLd
    static constexpr int t_fv = t_F + C * kQpLd;
    static constexpr int t_s1 = t_fv + C;                      // C x (T+1)
    static constexpr int qp_V = t_s1 + C * (T + 1);
    static constexpr int qp_x = qp_V + kQpLd * kQpLd + 32;
    static constexpr int WLD = 32;                             // row stride of the wrench maps: 1 + sum t_l + k <= 31 columns
    static constexpr int wm = ev(qp_x + kQpLd);                // C x WLD: A_rot Jbar[:,6:] [tg | U_0 .. | NwJw]  (wrench_maps below)
    static constexpr int t_wacc = wm + C * WLD;                // C: wrench of the torque committed so far, minus A_rot P_C
    static constexpr int t_cl = t_wacc + C;                    // K: contact_qp_ of the last task level that was solved
    static constexpr int t_end = ev(t_cl + K);
    static constexpr int total0 = max2(max2(k_end, c_end), t_end);
    // Jacobian of the COM link + com_pos (6 N + 3), for COM task levels: written at the end of stage 0 and read while the task
    // Jacobians are built, i.e. before stage 3a first writes U -- so it borrows the U block when that is large enough
    static constexpr bool jcm_in_U = NLV * M * T >= 6 * N + 4;
    static constexpr int Jcm = jcm_in_U ? U : ev(total0);
    static constexpr int total = jcm_in_U ? total0 : ev(total0) + 6 * N + 4;
    static constexpr int total_bytes = total * (int)sizeof(real_t) + 64;
    // names the compact map (Lds3 below) gives a place of their own
    static constexpr bool compact = false;
    static constexpr int k_anc = k_Ic;                         // ancestor indices of the pointer-jumping rounds (2 x NB)
    static constexpr int Rw0 = Rw;                             // pelvis rotation for the base columns of the point Jacobians
    static constexpr int c_QWp = c_Jt;                         // Q W^+ of stage 3a (c_Jt / c_T1 are free again by then)
    static constexpr int c_s2b = c_s2;                         // input of the small SPD inverses of stage 3a
    static constexpr int cod_Q = c_s1, cod_v = c_s1 + T * T, cod_G = c_s2 + T * T, cod_T = c_s2 + 2 * T * T;  // scratch of pinv_cod_small
    static constexpr int Pt = 0, c_Gi = 0;                     // (compact map only)
    static constexpr int xl(int lv) { return Xl + lv * M * T; }  // X = J_kt Lambda of level lv < NLV - 1
    static constexpr int MS = M;                               // row stride of T1r / c_Q (the compact maps pad it to an even number)
    static constexpr bool batch_reads = true;                  // row products issue their LDS reads in hand-made batches (lds_rows_dot)
};

// The COMPACT map of the lean, register-capped kernel (`_v2`, EXTRAS = false; batches beyond four instances per CU): the same
// blocks under the same names, laid out by life time so that two task levels fit 20 KB -- EIGHT workgroups per CU (two waves per
// SIMD) where Lds2's 31.6 KB allows five (profiles/r01_final_lds_coresidency.txt: <= 20 480 B for 8).  What changes against Lds2:
//   * the staged mass matrix is the row-packed lower triangle (as Lds2 does beyond 41 dof) and overlays the kinematics scratch
//     that is dead by then; the local rotations of the FK rounds overlay the inertia buffers that are written after them;
//   * J_C lives where Jbar^T will: the internal-wrench basis Vb, the only later reader of J_C, is formed right behind J_C;
//   * the link rotations die with stage 1's prologue: the task points are taken there (Pt), the point Jacobians keep pw, aw and
//     the pelvis rotation (Rw0);
//   * T1 = J_t A^-1 N_c goes straight to its M-wide home (T1r of the level, c_Q for the last) plus its six base columns (c_T1);
//   * VG = Vb G^-1 is not stored (G^-1, 36 doubles, is: c_Gi); X of level 0 IS U of level 0 (nothing projects it).
// Life times (region : stage 0 | stage 1 | NwJw | task Jacobians | W^+ | stage 3a | QP):
//   JbT  : k_S k_F | J_C, then Jbar^T ...........................................................
//   U    : Rw      | Rw (frames, task points) | . | J_t^T, T1 base columns | . | U ..............
//   RE   : pw aw ......................................................... | . | Q    | QP scratch
//   RF   : k_Iw k_Ic (k_Rl, k_A over them) | Vb ...... c_Gi ................... | Q W^+ | QP scratch
//   RG/RH: (k_Ic, k_A)   | Y            | . | Lambda_t, T1r, c_Q .......................... | .
//   RX   : (k_A)         | small-inverse scratch ...... | sweep column | Pi, Z, small inverses | .
template <int N, int NB, int NLV>
struct Lds3 {
    static constexpr bool compact = true;
    static constexpr int M = N - 6;
    static constexpr int C = 6 * kMaxActiveContacts;
    static constexpr int K = C - 6;
    static constexpr int T = kMaxTaskDof;
    static constexpr int max2(int a, int b) { return a > b ? a : b; }
    static constexpr int ev(int a) { return (a + 1) & ~1; }
    static constexpr int NL2 = NLV < 2 ? 2 : NLV;              // the U / T1r regions are sized for at least two levels (they host stage-0/1 data)
    // ---- head: alive for the whole cycle
    static constexpr int tg = 0;
    static constexpr int tt = tg + M;
    static constexpr int tc = tt + M;
    static constexpr int PC = tc + M;
    static constexpr int Rc = PC + C;
    static constexpr int Pc = Rc + kMaxActiveContacts * 9;
    static constexpr int fs = Pc + kMaxActiveContacts * 3;
    static constexpr int comp = fs + kMaxLevels * kMaxTaskDof;
    static constexpr int q = ev(comp + 4);                     // N+1, stage 0 only; then the gravity pre-vector (c_vec)
    static constexpr int G = q + ev(N + 1);                    // N, stages 0-1; then the diagonal of W (c_col)
    static constexpr int hend = G + ev(N);
    static constexpr int c_vec = q;
    static constexpr int c_col = G;
    // ---- long-lived blocks
    static constexpr int JbT = hend;                           // C x N
    static constexpr int c_JC = JbT;                           // N x C, until Jbar^T is written over it
    static constexpr int NwJw = JbT + C * N;                   // M x K
    static constexpr int FNl = NwJw + M * K;                   // C x K
    static constexpr int U = FNl + C * K;                      // levels x (M x T)
    static constexpr int RE = U + NL2 * M * T;
    static constexpr int pw = RE;
    static constexpr int aw = pw + NB * 3;
    static constexpr int Rw0 = aw + NB * 3;                    // 9 (+1)
    static constexpr int Pt = Rw0 + 10;                        // task points, kMaxLevels x kMaxTaskLinks x 3
    static constexpr int RF = ev(Pt + kMaxLevels * kMaxTaskLinks * 3);
    static constexpr int c_Vb = RF;                            // M x K
    static constexpr int c_Gi = c_Vb + M * K;                  // K x K
    static constexpr int c_VG = c_Vb;                          // (not stored in this map)
    static constexpr int RG = c_Gi + K * K;
    static constexpr int c_Lt = RG;                            // levels x T x T
    static constexpr int MS = ev(M);                           // row stride of T1r / c_Q: even, so that every row is 16-byte aligned (lds_rows_dot)
    // the hand-batched LDS reads pin 34 .. 68 VGPRs at once: inside the 256-register cap of this kernel the allocator then sends the
    // whole W^+ column to scratch around stage 3a (1 076 bytes, 19.4 -> 15.3 M cycles/s at B = 8192, profiles/r03i).  With two waves per
    // SIMD the other wave covers most of the LDS latency anyway: the compact kernel keeps the compiler-scheduled loops.
    static constexpr bool batch_reads = false;
    static constexpr int T1r = c_Lt + NL2 * T * T;             // (levels-1) x (T x MS)
    static constexpr int c_Q = T1r + (NL2 - 1) * T * MS;       // T x MS
    static constexpr int RX = c_Q + T * MS;                    // phase-local scratch
    static constexpr int rx_size = max2(max2(C * K, 64) + C * C, 64 + 3 * T * T + 8);
    static constexpr int RXend = RX + ev(rx_size);
    static constexpr int Xl = RXend;                           // X of levels 1 .. NLV-2 (level 0: U)
    static constexpr int total = Xl + (NLV > 2 ? (NLV - 2) * M * T : 0);
    static constexpr int xl(int lv) { return lv == 0 ? U : Xl + (lv - 1) * M * T; }
    // ---- stage 0
    static constexpr int Rw = U;                               // NB x 9, until the contact frames and task points are taken
    static_assert(NL2 * M * T >= NB * 9, "the link rotations borrow the U region");
    static constexpr int k_S = JbT;                            // N x 6
    static constexpr int k_F = k_S + N * 6;                    // N x 6
    static_assert(2 * N * 6 <= C * N, "S and F borrow the Jbar^T region");
    static constexpr int k_Iw = RF;                            // NB x 10 (first the ping-pong partner of pw)
    static constexpr int k_Ic = k_Iw + NB * 10;                // NB x 10
    static constexpr int k_Rl = k_Iw + NB * 3;                 // NB x 9: local rotations / ping-pong partner of Rw, dead before Iw and Ic are written
    static constexpr int k_anc = k_Ic + NB * 10 - 2 * NB;      // ancestor indices at the tail of k_Ic
    static_assert(k_Rl + NB * 9 <= k_anc, "local rotations must not reach the ancestor indices");
    static constexpr bool a_overlay = false, a_packed = true;
    static constexpr int k_A = RF;                             // N (N + 1) / 2, written after Ic has been consumed
    static_assert(k_Ic + NB * 10 <= total && k_A + N * (N + 1) / 2 <= total, "stage-0 scratch");
    // ---- stage 1
    static constexpr int c_Y = RG;                             // N x C, dead before Lambda_t / T1r / c_Q are written
    static_assert(c_Y + C * N <= RX, "Y borrows the Lambda_t / T1r / c_Q region");
    static constexpr int c_s1 = RX;                            // max(C x K, 64): wrench-map scratch, sweep column (one double per lane)
    static constexpr int c_s2 = c_s1 + max2(C * K, 64);        // C x C
    static constexpr int c_Lam = c_s2;
    // ---- task Jacobians
    static constexpr int c_Jt = U;                             // N x T (J_task transposed)
    static constexpr int c_T1 = c_Jt + N * T;                  // T x 6: the base columns of T1
    static_assert(c_T1 + T * 6 <= RE, "J_t^T and the base columns of T1 borrow the U region");
    // ---- stage 3a
    static constexpr int c_QW = RE;                            // Q, T x M
    static constexpr int c_QWp = RF;                           // Q W^+, T x M
    static_assert(T * M <= RF - RE && T * M <= RG - RF, "Q / Q W^+");
    static constexpr int c_Pi = c_s1 + 64;                     // T x T
    static constexpr int c_Z = c_Pi + T * T;                   // T x T
    static constexpr int c_s2b = c_Z + T * T;                  // T x T: input of the small SPD inverses of stage 3a
    static constexpr int cod_Q = c_s1, cod_v = c_s1 + T * T, cod_G = c_Z, cod_T = c_s2b + T * T;  // scratch of pinv_cod_small (c_Z is idle then)
    static_assert(cod_v + 3 * T <= c_Pi && cod_T + T * T <= RXend, "rank-revealing route scratch");
    // ---- QP scratch (RE .. RG)
    static constexpr int WLD = 32;
    static constexpr int t_base = RE;                          // M
    static constexpr int wm = ev(t_base + M);                  // C x WLD
    static constexpr int t_fv = wm + C * WLD;                  // C
    static constexpr int t_wacc = t_fv + C;                    // C
    static constexpr int t_cl = t_wacc + C;                    // K
    static constexpr int qp_V = t_cl + K;                      // kQpLd (the row a drop step passes through LDS)
    static constexpr int qp_x = qp_V + kQpLd;
    static constexpr int t_F = wm, t_s1 = wm;                  // (names of the Lds2 map the lean cascade does not use)
    static_assert(qp_x + kQpLd <= RG, "QP scratch");
    static constexpr int Jcm = 0;                              // (COM task levels: full build only)
    static constexpr int total_bytes = total * (int)sizeof(real_t) + 64;
};

// Symmetric Gauss-Jordan sweep on a column-per-lane register matrix.  Per pivot k every lane j applies
//     S[i][j] -= (c_i - delta_ik) * h_j,   c = column k,   h_j = c_j / d  (h_k = 1 - 1/d)
// to its own column: one FMA per element, the row-k and column-k special cases of the sweep fall out of the modified multiplier;
// on exit -S is the inverse.  Same arithmetic role as Eigen's llt().solve(I) (reference src/dwbc.cpp:307).
// The sweep without LDS.  The pivot column is broadcast with v_readlane (lane select = the uniform pivot index k).  Each
// lane's own element of the pivot ROW, S[k][lane] = S[lane][k], would be a dynamic register index; the pivot loop is
// therefore unrolled by 8 with k = 8*kb + kr: kr is static, and the element is picked from the five candidates
// s[kr], s[8+kr], ... by the uniform kb.  The "- delta_ik" of the multiplier is absorbed by storing the diagonal
// shifted by one (s[j][j] = S[j][j] - 1), so that readlane(s[k], k) is already c_k - 1; the true diagonal lives in dg.
template <int NN>
DWBC_WDEV int sweep_inverse_rl(PLA_REF(real_t, s, NN), PL_REF(real_t, dg), int npiv = NN) {
    DWBC_LANE_DECL;
    static_assert(NN <= 56, "seven candidate registers per kr");
    int ok = 1;
    LANES {
        DWBC_LANE_OPAQUE(lp);
#pragma unroll
        for (int i = 0; i < NN; i++) LV(s)[i] = (i == lp) ? LV(dg) - real_t(1.0) : LV(s)[i];
    }
    for (int kb = 0; kb < (NN + 7) / 8; kb++) {
#pragma unroll
        for (int kr = 0; kr < 8; kr++) {
            const int k = 8 * kb + kr;
            if (k < NN && k < npiv) {
                real_t d = BCAST(dg, k);
                if (!(d > real_t(0.0))) { ok = 0; d = real_t(1.0); }
                const real_t rp = fast_rcp(d);
#ifdef DWBC_HOST_EMU
                real_t snap_[NN];
                for (int i_ = 0; i_ < NN; i_++) snap_[i_] = s[k][i_];
#endif
                LANES {
                    // small NN: the pivot loop unrolls completely, k folds to a constant and the masks would be shared by every inlined
                    // copy (spills).  Large NN keeps the plain lane: an asm in the loop body sends the register matrix to scratch.
                    DWBC_LANE_OPAQUE_IF(lk, NN <= 16);
                    // own element k = 8*kb + kr: static candidates, uniform selector
                    real_t cj = LV(s)[kr];
                    if (8 + kr < NN) cj = (kb == 1) ? LV(s)[(8 + kr) < NN ? 8 + kr : 0] : cj;
                    if (16 + kr < NN) cj = (kb == 2) ? LV(s)[(16 + kr) < NN ? 16 + kr : 0] : cj;
                    if (24 + kr < NN) cj = (kb == 3) ? LV(s)[(24 + kr) < NN ? 24 + kr : 0] : cj;
                    if (32 + kr < NN) cj = (kb == 4) ? LV(s)[(32 + kr) < NN ? 32 + kr : 0] : cj;
                    if (40 + kr < NN) cj = (kb == 5) ? LV(s)[(40 + kr) < NN ? 40 + kr : 0] : cj;  // (models beyond TOCABI's 39 dof: kernel packs)
                    if (48 + kr < NN) cj = (kb == 6) ? LV(s)[(48 + kr) < NN ? 48 + kr : 0] : cj;
                    const bool piv = lk == k;
                    const real_t h = piv ? (real_t(1.0) - rp) : cj * rp;
#pragma unroll
                    for (int i = 0; i < NN; i++) {
#ifdef DWBC_HOST_EMU
                        const real_t ci = snap_[i];
#else
                        const real_t ci = readlane_f64(LV(s)[i], k);
#endif
                        LV(s)[i] -= ci * h;
                    }
                    LV(dg) = piv ? -rp : LV(dg) - cj * h;
                }
            }
        }
    }
    LANES {
        DWBC_LANE_OPAQUE(le);  // own compares: the prologue's masks are not kept alive (spilled) across the sweep
#pragma unroll
        for (int i = 0; i < NN; i++) LV(s)[i] = (i == le) ? -LV(dg) : -LV(s)[i];
        LV(dg) = -LV(dg);
    }
    return ok;
}

// The same sweep for a joint-space mass matrix, which is sparse by the kinematic tree: A[i][j] != 0 only if dof i is an
// ancestor or a descendant of dof j.  Pivoting leaves first (k = NN-1 .. 0; a parent's index is below its children's) keeps
// that pattern without fill-in: when k is the pivot all its descendants have been swept and none of its ancestors, the swept
// set is a forest of complete subtrees, so column k is non-zero exactly on the relatives of k (Featherstone's branch-induced
// sparsity).  Rows outside the relatives would receive c_i * h with c_i == 0.0 exactly, so leaving them out changes no bit
// of the result of this pivot order.  The pattern is a compile-time constant (Topo, dwbc_topo.h): the pivots are unrolled by
// template recursion and each keeps only its relatives' rows -- for TOCABI 753 of the 1521 FMA + broadcast pairs.
// LO: the matrix columns sit in lanes LO .. LO + NN - 1 (column c in lane LO + c); the other lanes may carry right-hand sides
template <class Topo, int NN, int K, int LO = 0>
DWBC_WDEV void tree_pivot(PLA_REF(real_t, s, NN), PL_REF(real_t, dg), int &ok) {
    DWBC_LANE_DECL;
    constexpr unsigned long long rel = Topo::relatives(K);
    real_t d = BCAST(dg, K + LO);
    int pos = d > real_t(0.0) ? 1 : 0;
    DWBC_FLAG_VGPR(pos);
    ok &= pos;
    if (!(d > real_t(0.0))) d = real_t(1.0);
    const real_t rp = fast_rcp(d);
#ifdef DWBC_HOST_EMU
    real_t snap_[NN];
    for (int i_ = 0; i_ < NN; i_++) snap_[i_] = s[K + LO][i_];
#endif
    LANES {
        DWBC_LANE_OPAQUE(lk);
        const bool piv = lk == K + LO;
        const real_t cj = LV(s)[K];
        const real_t h = piv ? (real_t(1.0) - rp) : cj * rp;
#pragma unroll
        for (int i = 0; i < NN; i++) {
            if ((rel >> i) & 1ull) {
#ifdef DWBC_HOST_EMU
                const real_t ci = snap_[i];
#else
                const real_t ci = readlane_f64(LV(s)[i], K + LO);
#endif
                LV(s)[i] -= ci * h;
            }
        }
        LV(dg) = piv ? -rp : LV(dg) - cj * h;
    }
    if constexpr (K > 0) tree_pivot<Topo, NN, K - 1, LO>(s, dg, ok);
}

template <class Topo, int NN, int LO = 0>
DWBC_WDEV int sweep_inverse_tree(PLA_REF(real_t, s, NN), PL_REF(real_t, dg)) {
    DWBC_LANE_DECL;
    static_assert(NN == Topo::ndof, "topology / kernel size mismatch");
    int ok = 1;
    LANES {
        DWBC_LANE_OPAQUE(lp);
#pragma unroll
        for (int i = 0; i < NN; i++) LV(s)[i] = (i == lp - LO) ? LV(dg) - real_t(1.0) : LV(s)[i];
    }
    tree_pivot<Topo, NN, NN - 1, LO>(s, dg, ok);
    ok = DWBC_FLAG_UNIFORM(ok);
    LANES {
        DWBC_LANE_OPAQUE(le);  // own compares: the prologue's masks are not kept alive (spilled) across the sweep
#pragma unroll
        for (int i = 0; i < NN; i++) LV(s)[i] = (i == le - LO) ? -LV(dg) : -LV(s)[i];
        LV(dg) = -LV(dg);
    }
    return ok;
}

// The dense sweep with the pivot column fed through LDS instead of v_readlane (fp64 device build; W^+, 33 x 33).  The swept
// matrix stays symmetric, so column K is row K across the lanes: ONE ds_write_b64 per pivot publishes it and broadcast
// ds_read_b128 bring it back, two rows per instruction.  Issue cost per row update (tools/ubench/ubench3): 2 v_readlane + s_nop
// + v_fma = 21 cycles against 8 (half a ds_read_b128) + 5.75; the compiler keeps only 2-3 such reads in flight, so all reads of
// a column are issued back to back by hand and waited for once (profiles/r02_sweep_prototypes.txt: 24.9 k -> 16.3 k cycles).
// The diagonal is carried shifted (dg - 2, next to the s[j][j] - 1 of the register sweep): with it the multiplier and the
// diagonal update of the pivot lane are the same FMAs as every other lane's -- no per-pivot selects.
#if !defined(DWBC_HOST_EMU)
typedef double dwbc_d2v __attribute__((ext_vector_type(2)));
template <int P, int NP>
__device__ __forceinline__ void lds_col_issue(dwbc_d2v (&c)[NP], unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c[P]) : "v"(addr), "n"(16 * P));
    if constexpr (P + 1 < NP) lds_col_issue<P + 1, NP>(c, addr);
}
template <int P, int NP>
__device__ __forceinline__ void lds_col_pin(dwbc_d2v (&c)[NP]) {  // (no instruction: the value of c[P] is defined after the wait)
    asm volatile("" : "+v"(c[P]));
    if constexpr (P + 1 < NP) lds_col_pin<P + 1, NP>(c);
}
template <int NP>
__device__ __forceinline__ void lds_col_wait(dwbc_d2v (&c)[NP]) {
    if constexpr (NP == 17) {  // TOCABI's 33-wide sweep: one statement
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]),
                     "+v"(c[9]), "+v"(c[10]), "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), "+v"(c[14]), "+v"(c[15]), "+v"(c[16]));
    } else {  // any width (the kernel packs): the wait, then every register re-defined behind it (volatile asms keep their order)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c[0]));
        if constexpr (NP > 1) lds_col_pin<1, NP>(c);
    }
}
#endif

// out[r] (MODE 0: =, MODE 1: -=)  sum_i A[r][i] x[i]  for NR rows of an LDS-resident matrix A (row stride RS, uniform address) against the
// lane's own register column x -- the "uniform operand . own column" product of this design.  Left to the compiler, every
// ds_read_b128 of such a loop is followed by its own s_waitcnt (the disassembly of the J_kt rows: 81 reads, 81 waits, one LDS round
// trip each: 9.8 k cycles for 198 FMAs); here the reads of CH rows are issued back to back and waited for once, as the W^+ sweep does
// for its pivot column.  Needs 16-byte-aligned rows (even RS, aligned base); anything else takes the plain loop.
// OFF / NO: the results go to out[OFF .. OFF + NR - 1] of an array of NO entries (a slice of a register column, addressed statically --
// a reference to `&column[OFF]` would make the whole column addressable and send it to scratch)
template <int NR, int NC, int RS, int CH, int MODE, bool ALIGNED = true, int OFF = 0, int NO = NR>
DWBC_WDEV void lds_rows_dot(const real_t *A, const real_t (&x)[NC], real_t (&out)[NO]) {
    static_assert(OFF + NR <= NO, "output slice");
#if !defined(DWBC_HOST_EMU)
    if constexpr (sizeof(real_t) == 8 && RS % 2 == 0 && ALIGNED) {
        constexpr int NP = (NC + 1) / 2;  // reads per row (the last one may fetch one element beyond NC: it stays inside the row's stride
                                          // when NC < RS, or belongs to the next row and is ignored)
        const unsigned a0 = (unsigned)(size_t)A;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r0 = 0; r0 < NR; r0 += CH) {
            dwbc_d2v c[CH][NP];
#pragma unroll
            for (int q = 0; q < CH; q++)
                if (r0 + q < NR) lds_col_issue<0, NP>(c[q], a0 + (unsigned)((r0 + q) * RS * 8));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < CH; q++)
                if (r0 + q < NR) lds_col_pin<0, NP>(c[q]);
#pragma unroll
            for (int q = 0; q < CH; q++) {
                if (r0 + q < NR) {
                    real_t a4[4] = {real_t(0.0), real_t(0.0), real_t(0.0), real_t(0.0)};
#pragma unroll
                    for (int i = 0; i < NC; i++) a4[i & 3] += c[q][i / 2][i & 1] * x[i];
                    const real_t d = (a4[0] + a4[1]) + (a4[2] + a4[3]);
                    if (MODE == 0) out[OFF + r0 + q] = d; else out[OFF + r0 + q] -= d;
                    asm volatile("" : "+v"(out[OFF + r0 + q]));  // (keeps the products here: sunk into a later conditional user they would hold every piece alive)
                }
            }
        }
        return;
    }
#endif
#pragma unroll
    for (int r = 0; r < NR; r++) {
        real_t a4[4] = {real_t(0.0), real_t(0.0), real_t(0.0), real_t(0.0)};
#pragma unroll
        for (int i = 0; i < NC; i++) a4[i & 3] += A[r * RS + i] * x[i];
        const real_t d = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        if (MODE == 0) out[OFF + r] = d; else out[OFF + r] -= d;
    }
}

// acc[p] += sum_{r < NR} A[r][p] * x[r]  for NR consecutive rows (NC entries each, row stride RS) of an LDS-resident matrix: the
// "own coefficients . uniform rows" product with the rows as the reduction index (the inverse-dynamics form of W^+ in dwbc_cycle2p.h:
// A = the vectors stored transposed, x = the lane's entries of its row of the mass matrix).  The reads of CH rows are issued back to
// back and waited for once (see lds_rows_dot); the accumulators are pinned behind the last batch so that the products stay here.
template <int NR, int NC, int RS, int CH, bool ALIGNED = true>
DWBC_WDEV void lds_rows_axpy(const real_t *A, const real_t (&x)[NR], real_t (&acc)[NC]) {
#if !defined(DWBC_HOST_EMU)
    if constexpr (sizeof(real_t) == 8 && RS % 2 == 0 && NC % 2 == 0 && ALIGNED) {
        constexpr int NP = NC / 2;
        const unsigned a0 = (unsigned)(size_t)A;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r0 = 0; r0 < NR; r0 += CH) {
            dwbc_d2v c[CH][NP];
#pragma unroll
            for (int q = 0; q < CH; q++)
                if (r0 + q < NR) lds_col_issue<0, NP>(c[q], a0 + (unsigned)((r0 + q) * RS * 8));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < CH; q++)
                if (r0 + q < NR) lds_col_pin<0, NP>(c[q]);
#pragma unroll
            for (int q = 0; q < CH; q++) {
                if (r0 + q < NR) {
#pragma unroll
                    for (int p = 0; p < NC; p++) acc[p] += c[q][p / 2][p & 1] * x[r0 + q];
                }
            }
        }
#pragma unroll
        for (int p = 0; p < NC; p++) asm volatile("" : "+v"(acc[p]));
        return;
    }
#endif
#pragma unroll
    for (int r = 0; r < NR; r++)
#pragma unroll
        for (int p = 0; p < NC; p++) acc[p] += A[r * RS + p] * x[r];
}

#if !defined(DWBC_HOST_EMU)
template <int NN, int K, int LO = 0>
__device__ __forceinline__ void lds_pivot(double (&s)[NN], double &dg2, int &ok, unsigned cb, int lane) {
    constexpr int NP = (NN + 1) / 2;
    dwbc_d2v col[NP];
    asm volatile("ds_write_b64 %0, %1" ::"v"(cb + 8 * ((lane - LO) & 63)), "v"(s[K]) : "memory");  // (64 doubles: every lane has a slot)
    lds_col_issue<0, NP>(col, cb);
    double d = readlane_f64(dg2, K + LO) + 2.0;
    int pos = d > 0.0 ? 1 : 0;
    DWBC_FLAG_VGPR(pos);
    ok &= pos;
    if (!(d > 0.0)) d = 1.0;
    const double rp = fast_rcp(d);
    const double cj = s[K];     // pivot lane: d - 1
    const double h = cj * rp;   // pivot lane: 1 - 1/d, the multiplier of the pivot column itself
    lds_col_wait<NP>(col);
#pragma unroll
    for (int i = 0; i < NN; i++) s[i] -= col[i / 2][i & 1] * h;
    dg2 -= cj * h;              // pivot lane: (d - 2) - (d - 2 + 1/d) = -1/d
    if constexpr (K > 0) lds_pivot<NN, K - 1, LO>(s, dg2, ok, cb, lane);
}
#endif

// The tree-sparse A^-1 sweep (sweep_inverse_tree) with the pivot column fed through LDS like the W^+ sweep: per pivot one ds_write_b64
// publishes it (the swept matrix stays symmetric), and only the 16-byte pieces that hold a relative of the pivot are read back -- the rows
// outside the relatives are exact zeros in this pivot order.  What it buys is VALU issue slots: a v_readlane-fed row costs two v_readlane
// and a hazard s_nop next to its FMA (753 rows: 2.3 k of the kernel's ~18 k VALU / SALU instructions), an LDS-fed row half a ds_read_b128
// on the LDS port.  Used by the two-wave kernel (dwbc_cycle2p.h: 71.2 -> 69.0 us per launch at B = 1024).  In the register-capped compact
// kernel the same sweep -- hand-batched or with compiler-scheduled plain loads alike -- makes the allocator spill inside the sweep
// (830 spilled registers, measured in round 4), so that kernel keeps the v_readlane feed.  colbuf: 64 doubles (every lane writes its
// slot; lanes beyond NN may carry right-hand sides, see dwbc_cycle2p.h).
#if !defined(DWBC_HOST_EMU)
// pieces P0 .. P1-1 of the pivot column (16 bytes = rows 2P, 2P + 1), those that hold a relative of pivot K
template <class Topo, int K, int P, int P1, int NP>
__device__ __forceinline__ void lds_rel_issue(dwbc_d2v (&c)[NP], unsigned addr) {
    if constexpr (P < P1) {
        if constexpr (((Topo::relatives(K) >> (2 * P)) & 3ull) != 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c[P]) : "v"(addr), "n"(16 * P));
        lds_rel_issue<Topo, K, P + 1, P1, NP>(c, addr);
    }
}
template <class Topo, int K, int P, int P1, int NP>
__device__ __forceinline__ void lds_rel_pin(dwbc_d2v (&c)[NP]) {
    if constexpr (P < P1) {
        if constexpr (((Topo::relatives(K) >> (2 * P)) & 3ull) != 0) asm volatile("" : "+v"(c[P]));
        lds_rel_pin<Topo, K, P + 1, P1, NP>(c);
    }
}
// CHP: pieces per batch of reads (a batch pins up to 4 CHP registers next to the 2 NN of the column)
template <class Topo, int NN, int K, int CHP>
__device__ __forceinline__ void lds_pivot_tree(double (&s)[NN], double &dg2, int &ok, unsigned cb, int lane) {
    constexpr unsigned long long rel = Topo::relatives(K);
    constexpr int NP = (NN + 1) / 2;
    dwbc_d2v col[NP];
    asm volatile("ds_write_b64 %0, %1" ::"v"(cb + 8 * lane), "v"(s[K]) : "memory");
    lds_rel_issue<Topo, K, 0, (CHP < NP ? CHP : NP), NP>(col, cb);
    double d = readlane_f64(dg2, K) + 2.0;
    int pos = d > 0.0 ? 1 : 0;
    DWBC_FLAG_VGPR(pos);
    ok &= pos;
    if (!(d > 0.0)) d = 1.0;
    const double rp = fast_rcp(d);
    const double cj = s[K];     // pivot lane: d - 1 (the diagonal is carried shifted, see lds_pivot)
    const double h = cj * rp;
#pragma unroll
    for (int p0 = 0; p0 < NP; p0 += CHP) {
        if (p0 > 0) {
            if (p0 == CHP) lds_rel_issue<Topo, K, (CHP < NP ? CHP : NP), (2 * CHP < NP ? 2 * CHP : NP), NP>(col, cb);
            else lds_rel_issue<Topo, K, (2 * CHP < NP ? 2 * CHP : NP), NP, NP>(col, cb);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (p0 == 0) lds_rel_pin<Topo, K, 0, (CHP < NP ? CHP : NP), NP>(col);
        else if (p0 == CHP) lds_rel_pin<Topo, K, (CHP < NP ? CHP : NP), (2 * CHP < NP ? 2 * CHP : NP), NP>(col);
        else lds_rel_pin<Topo, K, (2 * CHP < NP ? 2 * CHP : NP), NP, NP>(col);
#pragma unroll
        for (int i = 2 * p0; i < NN && i < 2 * (p0 + CHP); i++)
            if ((rel >> i) & 1ull) s[i] -= col[i / 2][i & 1] * h;
    }
    static_assert(3 * CHP >= NP, "at most three batches");
    dg2 -= cj * h;
    if constexpr (K > 0) lds_pivot_tree<Topo, NN, K - 1, CHP>(s, dg2, ok, cb, lane);
}
#endif
// f(std::integral_constant<int, I>) for I = I0 .. I1 - 1: a loop whose index is a constant expression inside the body
template <int I0, int I1, class F>
DWBC_WDEV void static_for(F &&f) {
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        static_for<I0 + 1, I1>(f);
    }
}

// ---- several pivots per step.  Two dofs that are not relatives (different branches of the tree) do not see each other's pivot:
// the sweep of k touches the entries (i, j) with i, j both relatives of k, and neither column k' nor its diagonal is among them
// when k' is not -- fill between two branches arises only when a COMMON ancestor is swept, and ancestors come last.  So the pivots
// of up to W branches are taken in one step: their columns are published and read back together, their multipliers formed
// together, and one LDS round trip and one reciprocal chain are paid per step instead of per pivot -- TOCABI's 39 pivots take 17
// steps of up to four (legs, arms and head side by side; the waist and the base one by one).  Any order that sweeps a dof after
// all of its descendants gives the same inverse (rounding aside); the schedule below takes, per step, the highest-numbered ready
// dofs that are pairwise unrelated.
template <int NN, int W>
struct TreeSchedule {
    int nsteps;
    int cnt[NN];
    int piv[NN][W];
};
template <class Topo, int NN, int W>
constexpr TreeSchedule<NN, W> make_tree_schedule() {
    TreeSchedule<NN, W> sch{};
    bool done[NN] = {};
    int left = NN, st = 0;
    while (left > 0) {
        int c = 0;
        for (int k = NN - 1; k >= 0 && c < W; k--) {
            if (done[k]) continue;
            const unsigned long long rel = Topo::relatives(k);
            bool ready = true;
            for (int j = k + 1; j < NN; j++)
                if (((rel >> j) & 1ull) && !done[j]) ready = false;
            for (int q = 0; q < c; q++)
                if ((rel >> sch.piv[st][q]) & 1ull) ready = false;
            if (ready) sch.piv[st][c++] = k;
        }
        for (int q = 0; q < c; q++) done[sch.piv[st][q]] = true;
        left -= c;
        sch.cnt[st] = c;
        st++;
    }
    sch.nsteps = st;
    return sch;
}
template <class Topo, int NN, int W>
struct TreeScheduleOf {
    static constexpr TreeSchedule<NN, W> value = make_tree_schedule<Topo, NN, W>();
};
#if !defined(DWBC_HOST_EMU)
template <int OFF>
__device__ __forceinline__ void lds_write_b64(unsigned addr, double v) {
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int CNT>
__device__ __forceinline__ void lds_wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(CNT) : "memory");
}
// 16-byte pieces of the pivot columns of step STEP that are asked for after those of its pivot number q
template <class Topo, int NN, int W, int STEP>
constexpr int lds_pieces_after(int q) {
    constexpr TreeSchedule<NN, W> sch = TreeScheduleOf<Topo, NN, W>::value;
    int n = 0;
    for (int r = q + 1; r < sch.cnt[STEP]; r++) {
        const unsigned long long rel = Topo::relatives(sch.piv[STEP][r]);
        for (int P = 0; P < (NN + 1) / 2; P++)
            if ((rel >> (2 * P)) & 3ull) n++;
    }
    return n;
}
template <class Topo, int NN, int W, int STEP>
__device__ __forceinline__ void lds_step_tree(double (&s)[NN], double &dg2, int &ok, unsigned cb, int lane) {
    using Sch = TreeScheduleOf<Topo, NN, W>;
    constexpr int cnt = Sch::value.cnt[STEP];
    constexpr int NP = (NN + 1) / 2, CS = 2 * NP;  // column stride in LDS: the rows of the matrix only (lanes beyond carry right-hand sides)
    dwbc_d2v col[W][NP];
    double cj[W], h[W];
    const unsigned slot = cb + 8u * (unsigned)(lane < CS - 1 ? lane : CS - 1);  // (row CS - 1 is padding when NN is odd; lanes beyond write there)
    static_assert(NN % 2 == 1 || W == 1, "an even NN has no padding row for the lanes beyond the matrix");
    static_for<0, cnt>([&](auto qc) {
        constexpr int q = decltype(qc)::value, K = Sch::value.piv[STEP][q];
        lds_write_b64<8 * q * CS>(slot, s[K]);
    });
    static_for<0, cnt>([&](auto qc) {
        constexpr int q = decltype(qc)::value, K = Sch::value.piv[STEP][q];
        lds_rel_issue<Topo, K, 0, NP, NP>(col[q], cb + (unsigned)(8 * q * CS));
    });
    static_for<0, cnt>([&](auto qc) {
        constexpr int q = decltype(qc)::value, K = Sch::value.piv[STEP][q];
        double d = readlane_f64(dg2, K) + 2.0;
        int pos = d > 0.0 ? 1 : 0;
        DWBC_FLAG_VGPR(pos);
        ok &= pos;
        if (!(d > 0.0)) d = 1.0;
        const double rp = fast_rcp(d);
        cj[q] = s[K];
        h[q] = cj[q] * rp;
    });
    // the columns come back in the order they were asked for: the rows of pivot q are updated as soon as ITS pieces are in, while
    // the later columns are still on their way (the counter saturates at 15: a wait for more than that waits for 15)
    static_for<0, cnt>([&](auto qc) {
        constexpr int q = decltype(qc)::value, K = Sch::value.piv[STEP][q];
        constexpr unsigned long long rel = Topo::relatives(K);
        constexpr int later = lds_pieces_after<Topo, NN, W, STEP>(q);
        lds_wait_lgkm<(later < 15 ? later : 15)>();
        lds_rel_pin<Topo, K, 0, NP, NP>(col[q]);
#pragma unroll
        for (int i = 0; i < NN; i++)
            if ((rel >> i) & 1ull) s[i] -= col[q][i / 2][i & 1] * h[q];
        dg2 -= cj[q] * h[q];
    });
    if constexpr (STEP + 1 < Sch::value.nsteps) lds_step_tree<Topo, NN, W, STEP + 1>(s, dg2, ok, cb, lane);
}
#endif
// colbuf: W * 2 * ((NN + 1) / 2) doubles
template <class Topo, int NN, int W>
DWBC_WDEV int sweep_inverse_tree_lds_multi(PLA_REF(real_t, s, NN), PL_REF(real_t, dg), real_t *colbuf) {
#if defined(DWBC_HOST_EMU)
    (void)colbuf;
    return sweep_inverse_tree<Topo, NN>(s, dg);
#else
    if constexpr (sizeof(real_t) != 8) {
        return sweep_inverse_tree<Topo, NN>(s, dg);
    } else {
        static_assert(NN == Topo::ndof, "topology / kernel size mismatch");
        const int lane = (int)(threadIdx.x & 63u);
        int ok = 1;
        {
            DWBC_LANE_OPAQUE(lp);
#pragma unroll
            for (int i = 0; i < NN; i++) s[i] = (i == lp) ? dg - 1.0 : s[i];
        }
        double dg2 = dg - 2.0;
        lds_step_tree<Topo, NN, W, 0>(s, dg2, ok, (unsigned)(size_t)colbuf, lane);
        ok = DWBC_FLAG_UNIFORM(ok);
        {
            DWBC_LANE_OPAQUE(le);
#pragma unroll
            for (int i = 0; i < NN; i++) s[i] = (i == le) ? -dg2 : -s[i];
            dg = -dg2;
        }
#pragma unroll
        for (int i = 0; i < NN; i++) asm volatile("" : "+v"(s[i]));  // (see sweep_inverse_tree_lds)
        return ok;
    }
#endif
}

template <class Topo, int NN, int CHP = (NN + 1) / 2>
DWBC_WDEV int sweep_inverse_tree_lds(PLA_REF(real_t, s, NN), PL_REF(real_t, dg), real_t *colbuf) {
#if defined(DWBC_HOST_EMU)
    (void)colbuf;
    return sweep_inverse_tree<Topo, NN>(s, dg);
#else
    if constexpr (sizeof(real_t) != 8) {
        return sweep_inverse_tree<Topo, NN>(s, dg);
    } else {
        static_assert(NN == Topo::ndof, "topology / kernel size mismatch");
        const int lane = (int)(threadIdx.x & 63u);
        int ok = 1;
        {
            DWBC_LANE_OPAQUE(lp);
#pragma unroll
            for (int i = 0; i < NN; i++) s[i] = (i == lp) ? dg - 1.0 : s[i];
        }
        double dg2 = dg - 2.0;
        lds_pivot_tree<Topo, NN, NN - 1, CHP>(s, dg2, ok, (unsigned)(size_t)colbuf, lane);
        ok = DWBC_FLAG_UNIFORM(ok);
        {
            DWBC_LANE_OPAQUE(le);
#pragma unroll
            for (int i = 0; i < NN; i++) s[i] = (i == le) ? -dg2 : -s[i];
            dg = -dg2;
        }
        // (the swept columns are pinned here: when their only readers sit in a conditional block -- the stores of the riding lanes in
        // dwbc_cycle2p.h -- the compiler sinks the row updates of every pivot into that block and keeps all the pivot columns alive
        // for it: 900 spilled registers inside the sweep)
#pragma unroll
        for (int i = 0; i < NN; i++) asm volatile("" : "+v"(s[i]));
        return ok;
    }
#endif
}

// LO: lane of matrix column 0 (the columns sit in lanes LO .. LO + NN - 1; colbuf: 64 doubles)
template <int NN, int LO = 0>
DWBC_WDEV int sweep_inverse_lds(PLA_REF(real_t, s, NN), PL_REF(real_t, dg), real_t *colbuf) {
#if defined(DWBC_HOST_EMU)
    (void)colbuf;
    return sweep_inverse_tree<TopoDense<NN>, NN, LO>(s, dg);
#else
    if constexpr (sizeof(real_t) != 8 || NN < 16) {  // fp32 build; small matrices: the register sweep
        return sweep_inverse_tree<TopoDense<NN>, NN, LO>(s, dg);
    } else {
        const int lane = (int)(threadIdx.x & 63u);
        int ok = 1;
        {
            DWBC_LANE_OPAQUE(lp);
#pragma unroll
            for (int i = 0; i < NN; i++) s[i] = (i == lp - LO) ? dg - 1.0 : s[i];
        }
        double dg2 = dg - 2.0;
        lds_pivot<NN, NN - 1, LO>(s, dg2, ok, (unsigned)(size_t)colbuf, lane);
        ok = DWBC_FLAG_UNIFORM(ok);
        {
            DWBC_LANE_OPAQUE(le);
#pragma unroll
            for (int i = 0; i < NN; i++) s[i] = (i == le - LO) ? -dg2 : -s[i];
            dg = -dg2;
        }
        return ok;
    }
#endif
}

// inverse of a small SPD matrix (n <= 12) held in LDS: column per lane in registers + the sweep above.  Used for
// Lambda_c^-1 = J A^-1 J^T, the null-space Gram matrix, Lambda_task^-1 and Q W^+ Q^T (all symmetric positive definite),
// where the reference calls Eigen's general inverse / COD pseudo-inverse (src/wbd.cpp:115,210,212).  Returns 0 when a
// pivot is not positive (rank deficient block => status 0, where the reference would return a pseudo-inverse).
// n <= 6: the Cholesky factor is computed redundantly by every lane on uniform (LDS-broadcast) data -- 56 FMAs, no
// cross-lane traffic -- and lane c < n then solves L L^T x = e_c for column c of the inverse with 30 FMAs.  About half
// the instructions of the 12-wide register sweep and none of its v_readlane broadcasts (a lone wave issues one instruction of
// any kind per ~5 cycles, a readlane-fed FMA costs ~21: profiles/r02_ubench3_instruction_costs.txt).
DWBC_WDEV int spd_inverse_chol6(const real_t *Ain, int lda, int n, real_t *Out, int ldo, real_t *pivratio = nullptr) {
    DWBC_LANE_DECL;
    real_t Lc[6][6], ri[6];
    int ok = 1;
    real_t dmin = kF32 ? real_t(1e30) : real_t(1e300), dmax = real_t(0.0);  // smallest / largest pivot of the block (rows < n)
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) Lc[i][j] = (i < n) ? Ain[i * lda + j] : (i == j ? real_t(1.0) : real_t(0.0));
#pragma unroll
    for (int j = 0; j < 6; j++) {
        real_t d = Lc[j][j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= Lc[j][k] * Lc[j][k];
        if (pivratio && j < n) { dmin = d < dmin ? d : dmin; dmax = d > dmax ? d : dmax; }
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
    if (pivratio) *pivratio = (ok && dmax > real_t(0.0)) ? dmin / dmax : real_t(0.0);
    DWBC_SYNC();  // Out may alias Ain
    LANES {
        DWBC_LANE_OPAQUE(lc);  // see DWBC_LANE_OPAQUE: the unit-vector masks of every inlined copy would otherwise be shared kernel-wide
        real_t y[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            real_t v = (lc == i) ? real_t(1.0) : real_t(0.0);
#pragma unroll
            for (int k = 0; k < i; k++) v -= Lc[i][k] * y[k];
            y[i] = v * ri[i];
        }
#pragma unroll
        for (int i = 5; i >= 0; i--) {
            real_t v = y[i];
#pragma unroll
            for (int k = i + 1; k < 6; k++) v -= Lc[k][i] * y[k];
            y[i] = v * ri[i];
        }
        if (lane < n) {
#pragma unroll
            for (int i = 0; i < 6; i++)
                if (i < n) Out[i * ldo + lane] = y[i];
        }
    }
    DWBC_SYNC();
    return ok;
}

// Three such inverses at once (densely stored: row stride n_q, in place allowed): the factorisation above is per-lane work on
// broadcast data, so three groups of eight lanes can each take a matrix of their own for the price of one -- the helper wave of
// dwbc_cycle2p.h inverts Lambda_task^-1 of two levels and the null-space Gram matrix this way.  n_q = 0: no such matrix.  ok[q] = 0
// when a pivot of matrix q is not positive.
DWBC_WDEV void spd_inverse_chol6_x3(const real_t *A0, int n0, real_t *O0, const real_t *A1, int n1, real_t *O1, const real_t *A2, int n2, real_t *O2, int (&ok)[3]) {
#if defined(DWBC_HOST_EMU)
    ok[0] = n0 > 0 ? spd_inverse_chol6(A0, n0, n0, O0, n0) : 1;
    ok[1] = n1 > 0 ? spd_inverse_chol6(A1, n1, n1, O1, n1) : 1;
    ok[2] = n2 > 0 ? spd_inverse_chol6(A2, n2, n2, O2, n2) : 1;
#else
    const int lane = (int)(threadIdx.x & 63u), g = lane >> 3, c = lane & 7;
    const real_t *Ag = g == 0 ? A0 : (g == 1 ? A1 : A2);
    real_t *Og = g == 0 ? O0 : (g == 1 ? O1 : O2);
    const int n = g == 0 ? n0 : (g == 1 ? n1 : (g == 2 ? n2 : 0));
    real_t Lc[6][6], ri[6];
    int okv = 1;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) {
            const real_t v_ = Ag[(i < n ? i * n + j : 0)];
            Lc[i][j] = (i < n) ? v_ : (i == j ? real_t(1.0) : real_t(0.0));
        }
#pragma unroll
    for (int j = 0; j < 6; j++) {
        real_t d = Lc[j][j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= Lc[j][k] * Lc[j][k];
        if (!(d > real_t(0.0))) { okv = 0; d = real_t(1.0); }
        ri[j] = fast_rsqrt(d);
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            real_t v = Lc[i][j];
#pragma unroll
            for (int k = 0; k < j; k++) v -= Lc[i][k] * Lc[j][k];
            Lc[i][j] = v * ri[j];
        }
    }
    DWBC_SYNC();  // in place allowed
    {
        real_t y[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            real_t v = (c == i) ? real_t(1.0) : real_t(0.0);
#pragma unroll
            for (int k = 0; k < i; k++) v -= Lc[i][k] * y[k];
            y[i] = v * ri[i];
        }
#pragma unroll
        for (int i = 5; i >= 0; i--) {
            real_t v = y[i];
#pragma unroll
            for (int k = i + 1; k < 6; k++) v -= Lc[k][i] * y[k];
            y[i] = v * ri[i];
        }
        if (c < n) {
#pragma unroll
            for (int i = 0; i < 6; i++)
                if (i < n) Og[i * n + c] = y[i];
        }
    }
    ok[0] = __builtin_amdgcn_readlane(okv, 0);
    ok[1] = __builtin_amdgcn_readlane(okv, 8);
    ok[2] = __builtin_amdgcn_readlane(okv, 16);
    DWBC_SYNC();
#endif
}

DWBC_DEVN int spd_inverse_small(const real_t *Ain, int lda, int n, real_t *Out, int ldo, real_t *colbuf, real_t *pivratio = nullptr) {
    DWBC_LANE_DECL;
    DWBC_SYNC();
    if (n <= 6) return spd_inverse_chol6(Ain, lda, n, Out, ldo, pivratio);
    // Jacobi scaling: the sweep runs on D^-1 A D^-1, D ~ sqrt(diag A) rounded to a power of two, and the inverse is scaled back.  The unpivoted sweep's error goes
    // with the condition number it sees, and J A^-1 J^T of a foot and a HAND mixes rows of 1e-1 with rows of 1e3 (light links): unscaled,
    // A^-1 N_c came out 4e-11 (relative) off and W^+ -- which amplifies it by 1 / lambda_min(W)^2 -- 2.5e-7, the torques 1e-5 Nm
    // against the restatement on every foot + hand pair (feet only: 1e-10).  The scaled matrix has its diagonal in [0.5, 2).
    PLA(real_t, s, 12);
    PL(real_t, dg);
    PL(real_t, dsc);
    // (the scale factors go through colbuf: twelve uniform doubles held in registers across the sweep cost the 256-register kernels spills)
    LANES {
        const int col = lane < n ? lane : 0;
        const real_t a = Ain[col * lda + col];
        // a power of two next to 1 / sqrt(a): the scaling is then exact (no rounding of its own) and costs two instructions
        int e2 = 0;
        (void)frexp(a > real_t(0.0) ? a : real_t(1.0), &e2);
        LV(dsc) = (lane < n && a > real_t(0.0)) ? ldexp(real_t(1.0), -(e2 >> 1)) : real_t(1.0);
        if (lane < 12) colbuf[lane] = LV(dsc);
    }
    DWBC_SYNC();
    LANES {
        const int col = lane < n ? lane : 0;
        const real_t dc = LV(dsc);
#pragma unroll
        for (int i = 0; i < 12; i++) LV(s)[i] = (lane < n && i < n) ? Ain[i * lda + col] * colbuf[i] * dc : real_t(0.0);
        LV(dg) = (lane < n) ? Ain[col * lda + col] * dc * dc : real_t(1.0);
    }
    const int ok = sweep_inverse_rl<12>(s, dg, n);
    DWBC_SYNC();
    LANES {
        if (lane < n) {
#pragma unroll
            for (int i = 0; i < 12; i++)
                if (i < n) Out[i * ldo + lane] = LV(s)[i] * colbuf[i] * LV(dsc);
        }
    }
    DWBC_SYNC();
    return ok;
}

// Householder QR (optionally with column pivoting) of a small matrix held in LDS: A (m x n, row stride n) becomes R, Q (m x m) is
// formed explicitly, piv[j] (stored as reals) = original column now in position j.  m, n <= 6.  Same steps as the restatement's
// qrcp() (oracle/dwbc_oracle.c) -- largest remaining column norm first, reflector sign against the pivot -- so that the rank
// decision of pinv_cod_small below falls on the same pivots as Eigen's CompleteOrthogonalDecomposition does in the reference.
template <int NT>
DWBC_DEVN void qr_small(Thr th, real_t *A, int m, int n, real_t *Q, real_t *piv, bool pivoting) {
    for (int idx = th.tid; idx < m * m; idx += NT) Q[idx] = (idx / m == idx % m) ? real_t(1.0) : real_t(0.0);
    for (int j = th.tid; j < n; j += NT) piv[j] = (real_t)j;
    const int steps = m < n ? m : n;
    for (int s = 0; s < steps; s++) {
        DWBC_SYNC();
        int best = s;
        if (pivoting) {
            real_t bn = -real_t(1.0);
            for (int j = s; j < n; j++) {
                real_t t = real_t(0.0);
                for (int i = s; i < m; i++) t += A[i * n + j] * A[i * n + j];
                if (t > bn) { bn = t; best = j; }
            }
        }
        DWBC_SYNC();
        if (best != s) {
            for (int i = th.tid; i <= m; i += NT) {
                if (i < m) { const real_t t = A[i * n + s]; A[i * n + s] = A[i * n + best]; A[i * n + best] = t; }
                else { const real_t t = piv[s]; piv[s] = piv[best]; piv[best] = t; }
            }
        }
        DWBC_SYNC();
        real_t v[kMaxTaskDofWide];  // (m <= 6 in the product kernels, <= 12 in the general-contact kernel's wide-task build)
#pragma unroll
        for (int i = 0; i < kMaxTaskDofWide; i++) v[i] = real_t(0.0);
        real_t nrm = real_t(0.0);
        for (int i = s; i < m; i++) { v[i] = A[i * n + s]; nrm += v[i] * v[i]; }
        nrm = sqrt(nrm);
        DWBC_SYNC();
        if (nrm == real_t(0.0)) continue;
        const real_t alpha = v[s] > real_t(0.0) ? -nrm : nrm;
        v[s] -= alpha;
        real_t vn = real_t(0.0);
        for (int i = s; i < m; i++) vn += v[i] * v[i];
        if (vn == real_t(0.0)) continue;
        const real_t beta = real_t(2.0) / vn;
        for (int idx = th.tid; idx < n + m; idx += NT) {
            if (idx < n) {  // column idx of A (>= s): reflector from the left
                const int j = idx;
                if (j >= s) {
                    real_t d = real_t(0.0);
                    for (int i = s; i < m; i++) d += v[i] * A[i * n + j];
                    d *= beta;
                    for (int i = s; i < m; i++) A[i * n + j] = (j == s && i > s) ? real_t(0.0) : A[i * n + j] - d * v[i];
                }
            } else {        // row of Q: Q = Q H
                const int i = idx - n;
                real_t d = real_t(0.0);
                for (int k = s; k < m; k++) d += Q[i * m + k] * v[k];
                d *= beta;
                for (int k = s; k < m; k++) Q[i * m + k] -= d * v[k];
            }
        }
    }
    DWBC_SYNC();
}

// Moore-Penrose pseudo-inverse of a small square block the way the reference takes it: complete orthogonal decomposition with
// Eigen's threshold rule, rank = #{ |R_ii| > thr max |R_ii| } on the column-pivoted QR (PinvCODWB, reference src/wbd.cpp:5-30 with
// COD_THRESHOLD = 1e-6, include/dwbc_wbd.h:10; restated in oracle/dwbc_oracle.c orc_pinv_cod).  Called only for a block the SPD
// factorisation found ill-conditioned (a nearly straight knee makes Q W^+ Q^T of a pelvis level lose a direction): a full-rank
// verdict leaves the caller's SPD inverse in place (returns t), a truncated one writes pinv(M_r) = P R1^+ Q1^T into Out.
//   Mr: the block (t x t, row stride t), destroyed.  Qm, Gm, Tm: t x t scratch each, vec: 3 t reals.
template <int NT>
DWBC_DEVN int pinv_cod_small(Thr th, real_t *Mr, int t, real_t thr, real_t *Out, real_t *Qm, real_t *Gm, real_t *Tm, real_t *vec) {
    real_t *piv = vec, *pz = vec + 6;
    qr_small<NT>(th, Mr, t, t, Qm, piv, true);
    real_t maxp = real_t(0.0);
    for (int i = 0; i < t; i++) { const real_t a = fabs(Mr[i * t + i]); maxp = a > maxp ? a : maxp; }
    int rank = 0;
    for (int i = 0; i < t; i++) rank += (fabs(Mr[i * t + i]) > thr * maxp) ? 1 : 0;
    if (rank == t) return t;
    DWBC_SYNC();
    if (rank == 0) {
        for (int idx = th.tid; idx < t * t; idx += NT) Out[idx] = real_t(0.0);
        DWBC_SYNC();
        return 0;
    }
    // R1 = first `rank` rows of R (rank x t).  QR of R1^T (t x rank) = Qz Tz; R1^+ = Qz[:, :rank] Tz^-T
    for (int idx = th.tid; idx < t * rank; idx += NT) Gm[idx] = Mr[(idx % rank) * t + idx / rank];
    DWBC_SYNC();
    qr_small<NT>(th, Gm, t, rank, Tm, pz, false);
    // X = Tz^-T (rank x rank) into Mr: column c by forward substitution with the lower-triangular Tz^T
    for (int c = th.tid; c < rank; c += NT) {
        for (int i = 0; i < rank; i++) {
            real_t sacc = (i == c) ? real_t(1.0) : real_t(0.0);
            for (int k = 0; k < i; k++) sacc -= Gm[k * rank + i] * Mr[k * rank + c];
            Mr[i * rank + c] = sacc / Gm[i * rank + i];
        }
    }
    DWBC_SYNC();
    // R1p = Qz[:, :rank] X (t x rank) into Gm (Tz is dead)
    for (int idx = th.tid; idx < t * rank; idx += NT) {
        const int i = idx / rank, j = idx - i * rank;
        real_t sacc = real_t(0.0);
        for (int k = 0; k < rank; k++) sacc += Tm[i * t + k] * Mr[k * rank + j];
        Out[idx] = sacc;            // staged in Out (t x rank, row stride rank)
    }
    DWBC_SYNC();
    for (int idx = th.tid; idx < t * rank; idx += NT) Gm[idx] = Out[idx];
    DWBC_SYNC();
    // pinv[piv[i]][j] = sum_k R1p[i][k] Q[j][k]
    for (int idx = th.tid; idx < t * t; idx += NT) {
        const int i = idx / t, j = idx - i * t;
        real_t sacc = real_t(0.0);
        for (int k = 0; k < rank; k++) sacc += Gm[i * rank + k] * Qm[j * t + k];
        Out[(int)piv[i] * t + j] = sacc;
    }
    DWBC_SYNC();
    return rank;
}

// Internal-wrench basis mapped to joint torques: Vb[r][a] = (J_C[:, 6 + r])^T lambda_a, where lambda_a (a < 6) is the wrench pair
// "unit wrench e_a on the second contact, the balancing wrench on the first" -- null(W) = { J_C[:,6:]^T lambda : J_C[:,:6]^T lambda = 0 }
// (closed form instead of the complete orthogonal decomposition of src/wbd.cpp:32-53).  JCt is J_C transposed (N x C), Pc the contact points.
template <int N, int NT>
DWBC_DEV void internal_wrench_basis(Thr th, const real_t *Pc, const real_t *JCt, real_t *Vb) {
    static_assert(kMaxActiveContacts == 2, "k in {0, 6}");
    constexpr int M = N - 6, C = 6 * kMaxActiveContacts, K6 = 6;
    // lambda_a = (unit wrench e_a on contact 1, the balancing wrench on contact 0: force -e_a, moment -m - d x f with d = P_1 - P_0),
    // so for column c = 6 + r of J_C (rows [f0 m0 f1 m1]):   a < 3 (unit force):  J[6+a] - J[a] - (d x e_a) . J[3:6]
    //                                                          a >= 3 (unit moment): J[6+a] - J[a]
    // one row of Vb per thread: the twelve entries of its column of J_C are two contiguous 48-byte runs
    const real_t d0 = Pc[3] - Pc[0], d1 = Pc[4] - Pc[1], d2 = Pc[5] - Pc[2];
    for (int r = th.tid; r < M; r += NT) {
        const real_t *Jc = JCt + (6 + r) * C;
        real_t j[12];
#pragma unroll
        for (int i = 0; i < 12; i++) j[i] = Jc[i];
        real_t *o = Vb + r * K6;
        o[0] = (j[6] - j[0]) - (d2 * j[4] - d1 * j[5]);
        o[1] = (j[7] - j[1]) - (d0 * j[5] - d2 * j[3]);
        o[2] = (j[8] - j[2]) - (d1 * j[3] - d0 * j[4]);
        o[3] = j[9] - j[3];
        o[4] = j[10] - j[4];
        o[5] = j[11] - j[5];
    }
}

// The contact wrench maps of EVERY QP of the cascade in one product, taken once after stage 3a:
//     WM = A_rot Jbar[:, 6:] [ tg | U_0 | U_1 ... | NwJw ]        (C x (1 + sum t_l + k), row stride WLD, contact-local frames)
// Column 0 is the wrench of the gravity torque, the U_l blocks are the maps F of CalcSingleTaskTorqueWithQP (dwbc.cpp:1018-1040),
// the last block is the map of the contact-null variables (dwbc.cpp:1041-1053 and CalcContactRedistribute :1458-1517).  Every
// right-hand side the cascade needs is a combination of these columns with the f* / QP answers of the levels (see the cascade), so
// the per-QP products `Jbar U`, `Jbar base`, `Jbar NwJw` of rounds 1-2 (three passes per QP, ~25 k cycles per cycle) are gone.
// fp64 device build: one 16 x 16 accumulator tile per 16 columns, v_mfma_f64_16x16x4_f64 over K = 33 (9 steps); the A operand is the
// ROTATED row of Jbar, formed on the fly from three LDS reads (operand layout verified on gfx950 by tools/ubench/mfma_stage.hip:
// A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], D[i = (l >> 4) + 4 r][j = l & 15]).  BASELINE configs[1] has exactly 16 columns.
template <int N, int NT, class S>
DWBC_DEV void wrench_maps(Thr th, const Setup &su, real_t *L, const real_t *JbT, int cd, int k, real_t *WM) {
    constexpr int M = N - 6, T = S::T, WLD = S::WLD;
    DWBC_LANE_DECL;
    const int ftot = su.fstar_total, ncols = 1 + ftot + k;
    // element (c, col) of the right-hand operand sits at L[bb + bs * c]
    auto col_src = [&](int col, int &bb, int &bs) {
        bb = S::tg; bs = 1;
        if (col >= 1 && col - 1 < ftot) {
            int l = 0;
            for (int q = 1; q < su.n_levels; q++) l = (col - 1 >= su.fstar_off[q]) ? q : l;
            bb = S::U + l * M * T + (col - 1 - su.fstar_off[l]);
            bs = T;
        } else if (col - 1 - ftot >= 0 && col - 1 - ftot < k) {
            bb = S::NwJw + (col - 1 - ftot);
            bs = 6;
        }
    };
#if !defined(DWBC_HOST_EMU)
    if constexpr (sizeof(real_t) == 8) {
        typedef double wm_d4 __attribute__((ext_vector_type(4)));
        const int li = lane & 15, lk = lane >> 4;
        const int a_ = li >= 6 ? 1 : 0, h = (li % 6) / 3, x = li % 3;   // (rows 12..15 of the tile are padding: any finite data)
        const real_t *R = L + S::Rc + a_ * 9;
        const real_t r0 = R[x], r1 = R[3 + x], r2 = R[6 + x];          // column x of R_a: row of A_rot = blockdiag(R^T, R^T)
        const real_t *J0 = JbT + (6 * a_ + 3 * h) * N + 6;
        const int ntile = (ncols + 15) >> 4;
        for (int tile = 0; tile < ntile; tile++) {
            const int col = 16 * tile + li;
            int bb, bs;
            col_src(col, bb, bs);
            wm_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < (M + 3) / 4; s++) {
                const int c = 4 * s + lk;
                const bool in = c < M;
                const int cc = in ? c : M - 1;
                real_t av = r0 * J0[cc] + r1 * J0[N + cc] + r2 * J0[2 * N + cc];
                av = in ? av : 0.0;
                const real_t bv = L[bb + bs * cc];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 3; r++) WM[(lk + 4 * r) * WLD + col] = acc[r];  // rows lk + 4 r < 12
        }
        DWBC_SYNC();
        return;
    }
#endif
    for (int idx = th.tid; idx < cd * ncols; idx += NT) {
        const int i = idx / ncols, col = idx - i * ncols;
        const int a_ = i / 6, h = (i % 6) / 3, x = i % 3;
        const real_t *R = L + S::Rc + a_ * 9;
        const real_t *J0 = JbT + (6 * a_ + 3 * h) * N + 6;
        int bb, bs;
        col_src(col, bb, bs);
        real_t acc = real_t(0.0);
        for (int c = 0; c < M; c++) acc += (R[x] * J0[c] + R[3 + x] * J0[N + c] + R[6 + x] * J0[2 * N + c]) * L[bb + bs * c];
        WM[i * WLD + col] = acc;
    }
    DWBC_SYNC();
}

// EXTRAS = false compiles the optional paths out (task-link trajectories, qdot outputs, COM / TASK_CUSTOM levels, hqp = false):
// merely having them inlined costs the common cycle 2 % (register allocation around the 39-column register matrix), so the
// launcher picks the lean instantiation whenever none of them is in use.
// COMPACT = true: the 20 KB LDS map Lds3 (lean build only; same arithmetic, blocks laid out by life time)
template <int N, int NB, int NLV, int NT, bool EXTRAS = true, class Topo = TopoGeneric, bool COMPACT = false>
DWBC_DEV void cycle_instance_v2(Thr th, const Setup &su, const BatchIO &io, int inst, real_t *L, int *iL) {
    static_assert(!(COMPACT && EXTRAS), "the compact LDS map has no room for the optional paths");
    using S = typename std::conditional<COMPACT, Lds3<N, NB, NLV>, Lds2<N, NB, NLV>>::type;
    constexpr bool kExtras = EXTRAS;
    constexpr int M = S::M, C = S::C, T = S::T;
    DWBC_LANE_DECL;
    (void)iL;
    // a constant tree fixes the body count and the depth as well (the launcher checked the model against it)
    constexpr bool kTree = !std::is_same<Topo, TopoGeneric>::value;
    const int nb = kTree ? NB : su.nb;
    const real_t *body = io.body;
    const int *topo = io.topo;  // parent[nb] depth[nb] subtree[nb]
    const io_t *qin = io.q + (size_t)inst * (N + 1);
    const DumpLayout dl = DumpLayout::make(N);
    real_t *dump = (EXTRAS && io.dump) ? io.dump + (size_t)inst * dl.total : nullptr;  // the lean build has no dump record
    int *diag = io.diag ? io.diag + (size_t)inst * DG_COUNT : nullptr;
    DWBC_STAMP_INIT();

    PLA(real_t, s, N);  // column `lane` of A -> A^-1 -> A^-1 N_c
    PL(real_t, dg);     // its diagonal element

#include "dwbc_cycle2_stage0.inc"

#include "dwbc_cycle2_stage1.inc"
    // ---- NwJw and the projector on null(W) from the closed-form internal-wrench basis (see dwbc_cycle.h stage 1)
    real_t *Vb = L + S::c_Vb, *VG = L + S::c_VG;
    if (k > 0) {
        // k is 0 or 6 (one or two 6D contacts): inside this block it is the constant K6, so that the small products below
        // unroll with compile-time strides after inlining
        static_assert(kMaxActiveContacts == 2, "k in {0, 6}");
        constexpr int K6 = 6;
        if constexpr (!S::compact) internal_wrench_basis<N, NT>(th, L + S::Pc, JCt, Vb);  // (compact map: formed in stage 1, before Jbar^T overwrites J_C)
        DWBC_SYNC();
        DWBC_FSTAMP(7);  // Vb
        for (int idx = th.tid; idx < K6 * K6; idx += NT) {
            const int i = idx / 6, j = idx - i * 6;  // k == 6 here
            real_t acc = real_t(0.0);
            _Pragma("unroll 8")
            for (int c = 0; c < M; c++) acc += JbT[i * N + 6 + c] * Vb[c * K6 + j];
            L[S::c_s2 + idx] = acc;
        }
        DWBC_FSTAMP(8);  // JV
        // NwJw = Vb (J̄1 Vb)^-1 (wbd.cpp:128) written with SPD inverses only: with G = Vb^T Vb, VG = Vb G^-1 and
        // JV = J̄[0:k,6:] Vb:  NwJw = VG JV^T (JV G^-1 JV^T)^-1  (same matrix: both satisfy J̄1 NwJw = I on span(Vb))
        DWBC_FSTAMP(8);
        {
            real_t *JV = L + S::c_s2, *Gi = L + S::c_s2 + 36, *Bm = L + S::c_s2 + 72, *Sm6 = L + S::c_s2 + 108;  // 4 x (6x6) in C*C = 144
            mm_tn<NT>(th, Gi, K6, Vb, K6, Vb, K6, K6, M, K6);                      // G
            DWBC_SYNC();
            spd_inverse_small(Gi, K6, K6, Gi, K6, L + S::c_s1);                   // G^-1
            if constexpr (!S::compact) mm_nn<NT>(th, VG, K6, Vb, K6, Gi, K6, M, K6, K6);  // VG = Vb G^-1   (projector P = VG Vb^T)
            else for (int idx = th.tid; idx < K6 * K6; idx += NT) L[S::c_Gi + idx] = Gi[idx];  // (compact map: G^-1 is kept instead of VG)
            mm_nn<NT>(th, Bm, K6, JV, K6, Gi, K6, K6, K6, K6);                       // B = JV G^-1
            DWBC_SYNC();
            mm_nt<NT>(th, Sm6, K6, Bm, K6, JV, K6, K6, K6, K6);                      // S = B JV^T  (SPD)
            DWBC_SYNC();
            if (!spd_inverse_small(Sm6, K6, K6, Sm6, K6, L + S::c_s1)) st_contact = 0;
            mm_nn<NT>(th, Bm, K6, Sm6, K6, JV, K6, K6, K6, K6);                      // X = S^-1 JV
            DWBC_SYNC();
            if constexpr (!S::compact) {
                mm_nt<NT>(th, L + S::NwJw, K6, VG, K6, Bm, K6, M, K6, K6);          // NwJw = VG X^T
            } else {
                mm_nt<NT>(th, JV, K6, Gi, K6, Bm, K6, K6, K6, K6);                   // G^-1 X^T (JV is dead)
                DWBC_SYNC();
                mm_nn<NT>(th, L + S::NwJw, K6, Vb, K6, JV, K6, M, K6, K6);          // NwJw = Vb (G^-1 X^T)
            }
            DWBC_SYNC();
        }
        DWBC_FSTAMP(12);  // VG
        // FNl = A_rot * (J̄[:,6:] NwJw)   (cd x k), contact-local frame: the QP cascade takes it from the wrench maps (wrench_maps(),
        // below); only the closed-form redistribution of hqp = false reads this block
        if (kExtras && !io.hqp)
        for (int idx = th.tid; idx < cd * K6; idx += NT) {
            const int i = idx / 6, j = idx - i * 6;  // k == 6 here
            real_t acc = real_t(0.0);
            _Pragma("unroll 8")
            for (int c = 0; c < M; c++) acc += JbT[i * N + 6 + c] * L[S::NwJw + c * K6 + j];
            L[S::c_s1 + idx] = acc;
        }
        DWBC_SYNC();
        if (kExtras && !io.hqp)
        for (int idx = th.tid; idx < cd * K6; idx += NT) {
            const int i = idx / 6, j = idx - i * 6;  // k == 6 here
            const int a = i / 6, h = (i % 6) / 3, x = i % 3;
            const real_t *R = L + S::Rc + a * 9;
            const real_t *src = L + S::c_s1 + (6 * a + 3 * h) * K6 + j;
            L[S::FNl + idx] = R[0 * 3 + x] * src[0] + R[1 * 3 + x] * src[K6] + R[2 * 3 + x] * src[2 * K6];
        }
        DWBC_SYNC();
    }
    DWBC_STAMP(12);  // (diagnostic) NwJw / projector block done
    DWBC_FSTAMP(13);  // FNl
    // ---- task Jacobians, T1 = J_t A^-1 N_c and Lambda_task for every level (dwbc.cpp:685-793, wbd.cpp:210)
    int fastmask = 0;  // levels whose J_kt needs no second inverse (see below)
    for (int lv = 0; lv < su.n_levels; lv++) {
        // the level body is instantiated for 3 and 6 task rows; in the lean build (no TASK_CUSTOM level) the row count of a
        // level is exactly one of the two, so `t` is a compile-time constant there and every t-dependent loop unrolls
        auto level_body = [&](auto ttl) {
        constexpr int TTL = decltype(ttl)::value;
        const int t = kExtras ? su.t_dof[lv] : TTL;
        real_t *Jtt = L + S::c_Jt, *T1 = L + S::c_T1, *Lt = L + S::c_Lt + lv * T * T;  // Jtt: N x T (J_task transposed)
        // compact map: T1 holds the six base columns only (T x 6); the joint columns go straight to T1r of the level / c_Q of the last
        real_t *T1x = (lv < NLV - 1) ? L + S::T1r + lv * T * S::MS : L + S::c_Q;
        const unsigned long long tm = su.t_dofmask[lv];
        DWBC_SYNC();
        for (int idx = th.tid; idx < T * N; idx += NT) Jtt[idx] = real_t(0.0);
        DWBC_SYNC();
        int row = 0;
        if (kExtras && su.t_custom_slot[lv] >= 0 && io.custom_J) {  // TASK_CUSTOM: J_task handed over by SetTaskSpace(h, f*, J) (dwbc.cpp:664-681)
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
            if (kExtras && link == nb) {  // the COM link: jac_ = jac_com_ (dwbc.cpp:352-353)
                com_task_rows<N, NT>(th, L + S::Jcm, Jtt, row, rsel, T);
            } else {
                real_t P[3];
                if constexpr (S::compact) {  // taken in stage 1, while the link rotations were alive
                    for (int a = 0; a < 3; a++) P[a] = L[S::Pt + (lv * kMaxTaskLinks + li) * 3 + a];
                } else {
                    const real_t *R = L + S::Rw + link * 9;
                    for (int a = 0; a < 3; a++) P[a] = L[S::pw + link * 3 + a] + R[a * 3] * pl[0] + R[a * 3 + 1] * pl[1] + R[a * 3 + 2] * pl[2];
                }
                point_jacobian<N, NB, NT>(th, L + S::Rw0, L + S::pw, L + S::aw, topo, nb, link, P, Jtt, 1, row, rsel == 0 ? 6 : 3, rsel, T);
            }
            row += rsel == 0 ? 6 : 3;
        }
        DWBC_SYNC();
        // T1[:, lane] = J_t * (lane's column of A^-1 N_c): all task rows in one pass, zero columns of J_t skipped
        // (instantiated for 3 and 6 task rows: a 3-dof level does not pay for the padding of the 6-wide blocks)
        auto t1_rows = [&](auto ttc) {
            constexpr int TT = decltype(ttc)::value;
            LANES {
                real_t tc_[TT];
#pragma unroll
                for (int r = 0; r < TT; r++) tc_[r] = real_t(0.0);
#pragma unroll
                for (int ib = 0; ib < N; ib += 3) {
                    if ((tm >> ib) & 7) {
#pragma unroll
                        for (int i = ib; i < ib + 3 && i < N; i++)
#pragma unroll
                            for (int r = 0; r < TT; r++) tc_[r] += Jtt[i * T + r] * LV(s)[i];
                    }
                }
#pragma unroll
                for (int r = 0; r < TT; r++) {
                    if constexpr (S::compact) {
                        if (lane < 6) T1[r * 6 + lane] = tc_[r];
                        else if (lane < N) T1x[r * S::MS + (lane - 6)] = tc_[r];
                    } else {
                        if (lane < N) T1[r * N + lane] = tc_[r];
                        if (lv < NLV - 1 && lane >= 6 && lane < N) L[S::T1r + lv * T * S::MS + r * S::MS + (lane - 6)] = tc_[r];
                    }
                }
            }
        };
        if (t <= 3) t1_rows(std::integral_constant<int, 3>{}); else t1_rows(std::integral_constant<int, T>{});
        DWBC_SYNC();
        if (lv == 0) DWBC_FSTAMP(14);  // level 0: Jt + T1
        for (int idx = th.tid; idx < t * t; idx += NT) {  // J_t A^-1 N_c J_t^T
            const int i = idx / t, j = idx - i * t;
            real_t acc = real_t(0.0);
#pragma unroll
            for (int c = 0; c < N; c++)
                if ((tm >> c) & 1) acc += (S::compact ? (c < 6 ? T1[i * 6 + c] : T1x[i * S::MS + (c < 6 ? 0 : c - 6)]) : T1[i * N + c]) * Jtt[c * T + j];
            L[S::c_s2 + idx] = acc;
        }
        if (lv == 0) DWBC_STAMP(13);  // (diagnostic) level-0 J_t and T1 done
        if (lv == 0) DWBC_FSTAMP(15);  // level 0: J A J^T
        const int ok_lt = spd_inverse_small(L + S::c_s2, t, t, Lt, t, L + S::c_s1);  // Lambda_task (wbd.cpp:210)
        {
            // With a 6D contact, A^-1 N_c has the rank of its joint block W, hence J A^-1 N_c J^T = T1r W^+ T1r^T and the block
            // Q W^+ Q^T of CalculateJKT (wbd.cpp:212) IS Lambda_task: J_kt = W^+ Q^T pinv(Lambda) = (T1r W^+)^T whenever the reference's
            // rank-revealing pseudo-inverse does not truncate.  A level whose Lambda_task^-1 is well conditioned takes that route in
            // stage 3a (no Q, no second t x t inverse); the estimate max diag(M) * max diag(M^-1) bounds the condition number from
            // below within a factor t^2, and the reference truncates at a pivot ratio of 1e-6.
            real_t da = real_t(0.0), dl = real_t(0.0);
            for (int i = 0; i < t; i++) {
                const real_t a_ = L[S::c_s2 + i * t + i], l_ = Lt[i * t + i];
                da = a_ > da ? a_ : da;
                dl = l_ > dl ? l_ : dl;
            }
            if (ok_lt && nc > 0 && da * dl < kCodCondFast) fastmask |= 1 << lv;
        }
        if (lv == 0) DWBC_FSTAMP(16);  // level 0: Lambda_task
        // Q = (Lambda J A^-1 N_c)[:,6:] is formed later from T1r; the last level keeps its T1r in the Q slot
        if (!S::compact && lv == NLV - 1)
            for (int idx = th.tid; idx < t * M; idx += NT) L[S::c_Q + (idx / M) * S::MS + idx % M] = T1[(idx / M) * N + 6 + idx % M];
        if (dump) {
            for (int idx = th.tid; idx < t * N; idx += NT) dump[dl.J_task + lv * T * N + idx] = Jtt[(idx % N) * T + idx / N];
            for (int idx = th.tid; idx < t * t; idx += NT) dump[dl.Lambda_task + lv * T * T + idx] = Lt[idx];
        }
            };
        if (su.t_dof[lv] <= 3) level_body(std::integral_constant<int, 3>{}); else level_body(std::integral_constant<int, T>{});
    }
    DWBC_SYNC();
    DWBC_STAMP(3);  // task Jacobians / Lambda_task / NwJw / projector done
    DWBC_FSTAMP(17);  // all task levels
    // ---- W^+ = (W + alpha P)^-1 - P / alpha, column c of W held by lane c (moved down from lane 6 + c)
    PLA(real_t, w, M);
    PL(real_t, dw);
    LANES {
        const int src = lane < M ? lane + 6 : lane;
#pragma unroll
        for (int i = 0; i < M; i++) LV(w)[i] = SHFLA(s, 6 + i, src);
        LV(dw) = SHFL(dg, src);
    }
    LANES {
        if (lane < M) L[S::c_col + lane] = LV(dw);
    }
    DWBC_SYNC();
    real_t alpha = real_t(0.0);
    for (int i = 0; i < M; i++) alpha += L[S::c_col + i];
    alpha /= M;
    const real_t ialpha = alpha != real_t(0.0) ? real_t(1.0) / alpha : real_t(0.0);
    DWBC_SYNC();
    PLA(real_t, vbr, 6);  // row `lane` of Vb
    PLA(real_t, pc, M);   // column `lane` of the projector P = VG Vb^T: kept in registers across the sweep for the correction
    LANES {
#pragma unroll
        for (int a = 0; a < 6; a++) LV(vbr)[a] = (k > 0 && lane < M) ? Vb[lane * 6 + a] : real_t(0.0);
#pragma unroll
        for (int i = 0; i < M; i++) LV(pc)[i] = real_t(0.0);
        if (k > 0) {
            DWBC_LANE_OPAQUE(lw);
            real_t dp = real_t(0.0);
            real_t gv[6] = {real_t(0.0), real_t(0.0), real_t(0.0), real_t(0.0), real_t(0.0), real_t(0.0)};
            if constexpr (S::compact) {
#pragma unroll
                for (int b = 0; b < 6; b++)
#pragma unroll
                    for (int a = 0; a < 6; a++) gv[b] += L[S::c_Gi + b * 6 + a] * LV(vbr)[a];
            }
            // column `lane` of the projector: P = Vb G^-1 Vb^T with gv = G^-1 (row `lane` of Vb) on the compact map, VG Vb^T otherwise
            if constexpr (S::compact) lds_rows_dot<M, 6, 6, 3, 0, S::batch_reads && S::c_Vb % 2 == 0>(Vb, gv, LV(pc));
            else lds_rows_dot<M, 6, 6, 3, 0, S::batch_reads && S::c_VG % 2 == 0>(VG, LV(vbr), LV(pc));
#pragma unroll
            for (int i = 0; i < M; i++) {
                const real_t pij = LV(pc)[i];
                LV(w)[i] += alpha * pij;
                dp = (i == lw) ? pij : dp;
            }
            LV(dw) += alpha * dp;
        }
        if (lane >= M) {
#pragma unroll
            for (int i = 0; i < M; i++) LV(w)[i] = real_t(0.0);
            LV(dw) = real_t(1.0);
        }
    }
    DWBC_FSTAMP(18);  // W + alpha P assembled
    // dense, but with compile-time pivots (TopoDense): no selector chain for the lane's own pivot-row element, 126.3 -> 125.3 us
#ifdef DWBC_W_SWEEP_READLANE
    if (!sweep_inverse_tree<TopoDense<M>, M>(w, dw)) st_contact = 0;
#else
    DWBC_SYNC();
    if (!sweep_inverse_lds<M>(w, dw, L + S::c_s1)) st_contact = 0;  // c_s1: 72 doubles, free between alpha and the small inverses
    DWBC_SYNC();
#endif
    DWBC_FSTAMP(19);  // W sweep
    LANES {
        if (k > 0) {
#pragma unroll
            for (int i = 0; i < M; i++) LV(w)[i] -= ialpha * LV(pc)[i];
        }
        // torque_grav_ = W^+ (A^-1 N_c G)[6:]   (wbd.cpp:190)
        real_t tg1[1];
        lds_rows_dot<1, M, 2 * ((M + 1) / 2), 1, 0, S::batch_reads && S::c_vec % 2 == 0>(L + S::c_vec + 6, LV(w), tg1);
        if (lane < M) L[S::tg + lane] = tg1[0];
    }
    if (dump) {
        LANES {
            if (lane < M) {
#pragma unroll
                for (int i = 0; i < M; i++) dump[dl.W_inv + i * M + lane] = LV(w)[i];
            }
        }
        for (int idx = th.tid; idx < M * k; idx += NT) dump[dl.NwJw + idx] = L[S::NwJw + idx];
        for (int i = th.tid; i < cd; i += NT) dump[dl.P_C + i] = L[S::PC + i];
    }
    DWBC_SYNC();
    DWBC_STAMP(4);  // W^+ and gravity compensation done

    DWBC_FSTAMP(20);  // W^+ correction + gravity torque
    // ================= stage 3a: task-space dynamics for every level (wbd.cpp:207-261) =================
    int rankbad = 0;
    for (int lv = 0; lv < su.n_levels; lv++) {
        // the level body is instantiated for 3 and 6 task rows; in the lean build (no TASK_CUSTOM level) the row count of a
        // level is exactly one of the two, so `t` is a compile-time constant there and every t-dependent loop unrolls
        auto level_body = [&](auto ttl) {
        constexpr int TTL = decltype(ttl)::value;
        const int t = kExtras ? su.t_dof[lv] : TTL;
        const real_t *Lt = L + S::c_Lt + lv * T * T;
        const FastDiv fdt(t);
        const real_t *T1rl = (lv < NLV - 1) ? L + S::T1r + lv * T * S::MS : L + S::c_Q;
        real_t *Q = L + S::c_QW, *QW = L + S::c_QWp, *Pi = L + S::c_Pi;  // (Lds2: c_Jt/c_T1 are free again)
        real_t *Ul = L + S::U + lv * M * T;
        real_t *Xs = (lv < NLV - 1) ? L + S::xl(lv) : Ul;
        int cond = 1;
        DWBC_SYNC();
        // the usual case (see the task-Jacobian stage): J_kt = (T1r W^+)^T, X = J_kt Lambda -- row `lane` of both from the lane's own
        // column of W^+, nothing passes through LDS but the result
        auto jkt_fast = [&](auto ttc) {
            constexpr int TT = decltype(ttc)::value;
            const bool exact = t == TT;
            LANES {
                real_t tw[TT];
                lds_rows_dot<TT, M, S::MS, 1, 0, S::batch_reads && (S::T1r % 2 == 0) && (S::c_Q % 2 == 0)>(T1rl, LV(w), tw);  // (T1r W^+)[r][lane] = J_kt[lane][r]
#pragma unroll
                for (int r = 0; r < TT; r++) {
                    tw[r] = (exact || r < t) ? tw[r] : real_t(0.0);
                    if (dump && lane < M && r < t) dump[dl.J_kt + lv * M * T + lane * t + r] = tw[r];
                }
#pragma unroll
                for (int r3 = 0; r3 < TT; r3++) {
                    real_t acc = real_t(0.0);
                    if (exact) {
#pragma unroll
                        for (int r2 = 0; r2 < TT; r2++) acc += tw[r2] * Lt[r2 * TT + r3];
                    } else {
#pragma unroll
                        for (int r2 = 0; r2 < TT; r2++) acc += (r2 < t && r3 < t) ? tw[r2] * Lt[r2 * t + r3] : real_t(0.0);
                    }
                    if (lane < M) {
                        Xs[lane * T + r3] = acc;
                        Ul[lane * T + r3] = acc;
                    }
                }
            }
        };
        const bool fast = (fastmask >> lv) & 1;
        if (fast) {
            if (t <= 3) jkt_fast(std::integral_constant<int, 3>{}); else jkt_fast(std::integral_constant<int, T>{});
        } else {
        for (int idx = th.tid; idx < t * M; idx += NT) {  // Q = Lambda T1[:,6:]
            const int i = idx / M, j = idx - i * M;
            real_t acc = real_t(0.0);
            if (i < t)
                _Pragma("unroll 8")
                for (int p = 0; p < t; p++) acc += Lt[i * t + p] * T1rl[p * S::MS + j];
            Q[idx] = acc;
        }
        DWBC_SYNC();
        }
        auto jkt_rows = [&](auto ttc) {
            constexpr int TT = decltype(ttc)::value;
            for (int r = 0; r < TT; r++) {
                LANES {
                    real_t a4[4] = {real_t(0.0), real_t(0.0), real_t(0.0), real_t(0.0)};
#pragma unroll
                    for (int i = 0; i < M; i++) a4[i & 3] += Q[r * M + i] * LV(w)[i];
                    const real_t acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
                    if (lane < M) QW[r * M + lane] = acc;  // (Q W^+)[r][lane]
                }
            }
            DWBC_SYNC();
            if (lv == 0) DWBC_FSTAMP(21);  // level 0: Q, QW
            mm_nt<NT>(th, L + S::c_s2b, t, QW, M, Q, M, t, M, t);  // Q W^+ Q^T
            real_t pr_ = real_t(1.0);
            cond = spd_inverse_small(L + S::c_s2b, t, t, Pi, t, L + S::c_s1, &pr_);  // PinvCODWB, full-rank case: the SPD inverse
            DWBC_SYNC();
            if (!cond || pr_ < kCodCheck) {
                // ill-conditioned block (a nearly straight knee under a pelvis level): decide the rank as the reference's complete
                // orthogonal decomposition does (threshold 1e-6 on the pivoted QR) and take the truncated pseudo-inverse if it truncates
                const int rk = pinv_cod_small<NT>(th, L + S::c_s2b, t, kCodThreshold, Pi, L + S::cod_Q, L + S::cod_G, L + S::cod_T, L + S::cod_v);
                if (rk < t) cond = 1;
            }
            if (lv == 0) DWBC_FSTAMP(22);  // level 0: QWQ^T + inverse
            // J_kt = W^+ Q^T pinv(.) ; X = J_kt Lambda ; U = Null_{lv-1} X   -- row `lane` of each in registers
            LANES {
                real_t jk[TT], qw[TT];
#pragma unroll
                for (int r = 0; r < TT; r++) qw[r] = QW[r * M + (lane < M ? lane : 0)];
                const bool exact = t == TT;  // the usual case (task dof 3 or 6): compile-time stride, no per-element guards
#pragma unroll
                for (int r2 = 0; r2 < TT; r2++) {
                    real_t acc = real_t(0.0);
                    if (exact) {
#pragma unroll
                        for (int r = 0; r < TT; r++) acc += qw[r] * Pi[r * TT + r2];
                    } else {
#pragma unroll
                        for (int r = 0; r < TT; r++) acc += (r < t && r2 < t) ? qw[r] * Pi[r * t + r2] : real_t(0.0);
                    }
                    jk[r2] = acc;
                    if (dump && lane < M && r2 < t) dump[dl.J_kt + lv * M * T + lane * t + r2] = acc;
                }
#pragma unroll
                for (int r3 = 0; r3 < TT; r3++) {
                    real_t acc = real_t(0.0);
                    if (exact) {
#pragma unroll
                        for (int r2 = 0; r2 < TT; r2++) acc += jk[r2] * Lt[r2 * TT + r3];
                    } else {
#pragma unroll
                        for (int r2 = 0; r2 < TT; r2++) acc += (r2 < t && r3 < t) ? jk[r2] * Lt[r2 * t + r3] : real_t(0.0);
                    }
                    if (lane < M) {
                        Xs[lane * T + r3] = acc;
                        Ul[lane * T + r3] = acc;
                    }
                }
            }
        };
        if (!fast) { if (t <= 3) jkt_rows(std::integral_constant<int, 3>{}); else jkt_rows(std::integral_constant<int, T>{}); }
        DWBC_SYNC();
        for (int pl = lv - 1; pl >= 0; pl--) {  // U <- (I - X_pl Y_pl) U,  Y_pl = T1r[pl]
            const int tp = su.t_dof[pl];
            const real_t *Xp = L + S::xl(pl), *Yp = L + S::T1r + pl * T * S::MS;
            for (int idx = th.tid; idx < tp * t; idx += NT) {
                const int i = fdt.div(idx), j = idx - i * t;
                real_t acc = real_t(0.0);
                _Pragma("unroll 8")
                for (int c = 0; c < M; c++) acc += Yp[i * S::MS + c] * Ul[c * T + j];
                L[S::c_Z + idx] = acc;
            }
            DWBC_SYNC();
            for (int idx = th.tid; idx < M * t; idx += NT) {
                const int i = fdt.div(idx), j = idx - i * t;
                real_t acc = Ul[i * T + j];
                _Pragma("unroll 8")
                for (int p = 0; p < tp; p++) acc -= Xp[i * T + p] * L[S::c_Z + p * t + j];
                Ul[i * T + j] = acc;
            }
            DWBC_SYNC();
        }
        if (!cond) rankbad |= (1 << lv);
        DWBC_STAMP(6 + 3 * lv);
            };
        if (su.t_dof[lv] <= 3) level_body(std::integral_constant<int, 3>{}); else level_body(std::integral_constant<int, T>{});
    }
    DWBC_SYNC();

    // ================= stage 3b + 4: the QP cascade (dwbc.cpp:818-873, 941-1127) followed by the contact
    // redistribution QP (dwbc.cpp:1372-1568).  One loop, ONE inlined copy of the solver: pass qi < n_levels is task
    // level qi, the last pass is the redistribution. =================
    const int nlim = su.has_tau_lim ? 2 * M : 0;
    const int ncone = 10 * nc;
    int st_task = 1, fail_level = -1, st_redis = 1;
    const real_t *fs_in = L + S::fs;  // filled by task_reference() after stage 0
    real_t *base = L + S::t_base, *fv = L + S::t_fv, *WM = L + S::wm, *wacc = L + S::t_wacc, *clast = L + S::t_cl;
    constexpr int WLD = S::WLD;
    const int colN = 1 + su.fstar_total;  // first column of the contact-null block of the wrench maps
    QpLaneConst qc;
    PL(real_t, sfin);  // slack of the lane's QP row at the point the last QP returned (qp_solve_wave)
    PL(real_t, grn);   // norm of the lane's row of the contact redistribution QP
    bool skip_redis = false;
    const bool wm_ok = kExtras ? io.hqp != 0 : true;  // the wrench maps exist (hqp = true)
    if (wm_ok) {
        qp_lane_consts<N>(su, act_c[0], act_c[1], qc);
        wrench_maps<N, NT, S>(th, su, L, JbT, cd, k, WM);
        // wacc = wrench of the torque committed so far (gravity torque to begin with) minus A_rot P_C, contact-local frames
        for (int i = th.tid; i < C; i += NT) {
            real_t acc = real_t(0.0);
            if (i < cd) {
                const int a = i / 6, h = (i % 6) / 3, x = i % 3;
                const real_t *R = L + S::Rc + a * 9;
                const real_t *pc3 = L + S::PC + 6 * a + 3 * h;
                acc = WM[i * WLD] - (R[x] * pc3[0] + R[3 + x] * pc3[1] + R[6 + x] * pc3[2]);
            }
            wacc[i] = acc;
            if (i < S::K) clast[i] = real_t(0.0);
        }
        DWBC_SYNC();
        DWBC_FSTAMP(43);  // wrench maps of the cascade
        redis_row_norms<N, S>(L, nlim, ncone, k, qc, WM + colN, grn);
    }
    for (int qi = 0; qi <= su.n_levels; qi++) {
        const bool is_task = qi < su.n_levels;
        if (is_task && !st_task) continue;  // cascade aborted (dwbc.cpp:836,845): later levels are skipped
        if (kExtras && !is_task && !io.hqp) {          // CalcContactRedistribute(hqp = false): closed form (dwbc.cpp:1570-1619)
            st_redis = redistribute_closed_form<S, N, NB, NT>(th, L, JbT, cd, k);
            break;
        }
        if (!is_task && k == 0) break;      // nothing to redistribute (dwbc.cpp:1562-1567)
        if (!is_task && skip_redis) {  // every row of the redistribution QP holds at c = 0 (seen from the last task QP's final slacks): no step
            if (diag && th.tid == 0) {
                diag[DG_QP_ITER + kMaxLevels] = 0;
                diag[DG_QP_NACT + kMaxLevels] = 0;
                if (EXTRAS)
                    for (int a = 0; a < kQpLd; a++) diag[DG_QP_ACT + kMaxLevels * kQpLd + a] = -1;
            }
            break;
        }
        const int t = is_task ? su.t_dof[qi] : 0;
        const real_t *Ul = L + S::U + (is_task ? qi : 0) * M * T;
        const real_t *fs = fs_in + (is_task ? su.fstar_off[qi] : 0);
        const int colL = 1 + (is_task ? su.fstar_off[qi] : 0);  // this level's block of the wrench maps
        if (is_task && (rankbad & (1 << qi))) { st_task = 0; fail_level = qi; continue; }  // rank-deficient task block
        if (kExtras && is_task && !io.hqp) {  // CalcTaskControlTorque(hqp = false): torque_task_ += Null_{i-1} J_kt Lambda f* (dwbc.cpp:856-873)
            DWBC_SYNC();
            for (int i = th.tid; i < M; i += NT) {
                real_t acc = real_t(0.0);
                for (int j = 0; j < t; j++) acc += Ul[i * T + j] * fs[j];
                L[S::tt + i] += acc;
            }
            DWBC_SYNC();
            continue;
        }
        DWBC_SYNC();
        // torque the level starts from (dwbc.cpp:1001-1016) and its contact wrench minus P_C in the contact frames, both from
        // what is already there: fv = wacc + F_l f*_l (task level) or wacc + F_N contact_qp_ (redistribution)
        {
            // straight-line: six unconditional reads per lane, the entries beyond t (or k) masked afterwards -- a loop with a run-time
            // trip count puts every read behind its own wait
            real_t f6[6];
#pragma unroll
            for (int j = 0; j < 6; j++) f6[j] = is_task ? (j < t ? fs[j] : real_t(0.0)) : (j < k ? clast[j] : real_t(0.0));
            for (int i = th.tid; i < M + C; i += NT) {
                const bool tq = i < M;
                const int r = tq ? 0 : i - M;
                const real_t *row = tq ? Ul + i * T : WM + r * WLD + (is_task ? colL : colN);
                real_t u6[6];
#pragma unroll
                for (int j = 0; j < 6; j++) u6[j] = row[j];
                real_t acc = tq ? L[S::tg + i] + L[S::tt + i] + (is_task ? real_t(0.0) : L[S::tc + i]) : wacc[r];
#pragma unroll
                for (int j = 0; j < 6; j++) acc += ((is_task ? j < t : (j < k && !tq)) ? u6[j] * f6[j] : real_t(0.0));
                if (tq) base[i] = acc; else fv[r] = acc;
            }
        }
        DWBC_SYNC();
        if (qi == 0) DWBC_FSTAMP(42);  // level 0: base torque and wrench
        if (is_task) DWBC_STAMP(7 + 3 * qi);
        QpResult qres;
        {
            // task level: x = [f*_qp (t) ; contact_qp (k)], rows [U | s NwJw];  redistribution: x = c (k), rows [NwJw]
            const real_t *P1 = is_task ? Ul : L + S::NwJw;
            // strides of NwJw: k is 0 (block unused) or 6, so the constant 6 serves both and folds after inlining
            static_assert(T == 6, "stride of U equals the stride of NwJw");
            const int n1 = is_task ? t : k, n2 = is_task ? k : 0;
            const real_t *W1 = WM + (is_task ? colL : colN);
            qp_rows_and_solve<N, NB, EXTRAS ? 1 : 0>(su, L, nlim, ncone, act_c[0], act_c[1], P1, 6, n1, L + S::NwJw, 6, n2,
                                     is_task ? kQpScaleGI : real_t(1.0), W1, WLD, WM + colN, WLD, fv, base, n1,
                                     is_task ? su.qp_max_iter_task : su.qp_max_iter_contact, qres, L + S::qp_V, L + S::qp_x,
                                     (EXTRAS && io.warm && diag) ? diag + DG_QP_ACT + (is_task ? qi : kMaxLevels) * kQpLd : nullptr, &qc,
                                     is_task ? kQpTol : kQpFeasTol, sfin);
        }
        const int slot = is_task ? qi : kMaxLevels;
        if (diag && th.tid == 0) {
            diag[DG_QP_ITER + slot] = qres.iters;
            diag[DG_QP_NACT + slot] = qres.nact;
            if (EXTRAS)  // working sets: diagnostics of the full build only
                for (int a = 0; a < kQpLd; a++) diag[DG_QP_ACT + slot * kQpLd + a] = qres.act[a];
        }
        if (dump && th.tid == 0) dump[dl.qp_viol + slot] = qres.viol;
#if defined(DWBC_STAGE_TIMERS) && !defined(DWBC_HOST_EMU)
        if (diag && th.tid == 0 && qi == 0)
            for (int i_ = 0; i_ < 9; i_++) diag[DG_FTIME + (i_ < 8 ? 23 + i_ : 44)] = (int)qres.tm[i_];
#endif
        if (is_task) DWBC_STAMP(8 + 3 * qi);
        const real_t *x = L + S::qp_x;
        if (is_task) {
            if (!qres.status) { st_task = 0; fail_level = qi; continue; }  // cascade aborts (dwbc.cpp:836,1119)
            {
                real_t fx[6], xc[6];  // f* + f*_qp of the level, contact_qp_ (uniform), entries beyond t / k zero
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    fx[j] = j < t ? fs[j] + x[j] : real_t(0.0);
                    xc[j] = j < k ? x[t + j] : real_t(0.0);
                }
                for (int i = th.tid; i < M + C; i += NT) {
                    const bool tq = i < M;
                    const int r = tq ? 0 : i - M;
                    const real_t *row = tq ? Ul + i * T : WM + r * WLD + colL;
                    real_t u6[6], n6[6];
#pragma unroll
                    for (int j = 0; j < 6; j++) { u6[j] = row[j]; n6[j] = L[S::NwJw + (tq ? i : 0) * 6 + j]; }
                    real_t acc = real_t(0.0), c = real_t(0.0);
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        acc += j < t ? u6[j] * fx[j] : real_t(0.0);
                        c += j < k ? n6[j] * xc[j] : real_t(0.0);
                    }
                    if (tq) {
                        L[S::tt + i] += acc;  // torque_task_ += Null_{i-1} J_kt Lambda (f* + f*_qp)   (dwbc.cpp:839-849)
                        L[S::tc + i] = c;     // torque_contact_ = NwJw contact_qp_              (dwbc.cpp:851)
                    } else {
                        wacc[r] += acc;       // the wrench of what this level commits
                        if (r < S::K) clast[r] = r < k ? x[t + r] : real_t(0.0);  // contact_qp_ for the redistribution's start
                    }
                }
            }
            if (qi == su.n_levels - 1 && k > 0 && (kExtras ? !io.warm : true)) {
                // The rows of the redistribution QP (dwbc.cpp:1458-1517) are the rows of this level's QP seen at c = 0 with the total
                // torque on the right-hand side, so its first look at them -- slack / |row| against the tolerance of canon rule 5 --
                // can be taken from the slacks this QP ended with.  Nothing violated: the redistribution would return c = 0 after
                // filling and normalising its 53 rows (6 k cycles in the stage table); it is skipped.
                LANES {
                    const bool tq = lane < M && nlim != 0, cn = lane >= M && lane - M < ncone;
                    LV(grn) = (tq || cn) ? LV(sfin) / LV(grn) : DWBC_QP_INF;  // (grn is not needed again)
                }
                int wl_;
                WAVE_ARGMIN_F32(grn, wl_);
                skip_redis = !(BCAST(grn, wl_) < -kQpFeasTol);
            }
            if (dump) {
                for (int j = th.tid; j < t; j += NT) dump[dl.fstar_qp + qi * T + j] = x[j];
                for (int j = th.tid; j < k; j += NT) dump[dl.contact_qp + qi * (C - 6) + j] = x[t + j];
            }
        } else if (qres.status) {
            for (int i = th.tid; i < M + S::K; i += NT) {
                if (i >= M) { clast[i - M] += (i - M < k) ? x[i - M] : real_t(0.0); continue; }  // contact_qp_ + redistribution: what torque_contact_ stands for
                real_t c = real_t(0.0);
                _Pragma("unroll 8")
                for (int j = 0; j < k; j++) c += L[S::NwJw + i * 6 + j] * x[j];
                L[S::tc + i] += c;    // torque_contact_ += NwJw c   (dwbc.cpp:1549)
            }
            if (dump)
                for (int j = th.tid; j < k; j += NT) dump[dl.cf_redis + j] = x[j];
        } else {
            st_redis = 0;
            for (int i = th.tid; i < M + S::K; i += NT) {
                if (i < M) L[S::tc + i] = real_t(0.0); else clast[i - M] = real_t(0.0);
            }  // dwbc.cpp:1553-1559
        }
        DWBC_SYNC();
    }
    if (k == 0) {
        for (int i = th.tid; i < M; i += NT) L[S::tc + i] = real_t(0.0);  // dwbc.cpp:1562-1567
    }
    DWBC_SYNC();
    DWBC_STAMP(15);

    // ================= outputs =================
    io_t *tau = io.tau + (size_t)inst * 3 * M;
    for (int i = th.tid; i < 3 * M; i += NT) tau[i] = too_many ? real_t(0.0) : L[S::tg + i];
    io_t *wr = io.wrench + (size_t)inst * 12;
    // getContactForce(tau_total) = Jbar[:, 6:] tau - P_C (wbd.cpp:268-271).  In the contact frames that is the running wrench of the
    // cascade plus the contact-null part (wacc + F_N (contact_qp_ + redistribution): every term a column combination of the wrench
    // maps); rotated back with blockdiag(R_a, R_a).  (Rounds 1-2 multiplied Jbar with the summed torque again: 33 x 4 LDS reads per lane.)
    for (int i = th.tid; i < 12; i += NT) {
        real_t acc = real_t(0.0);
        if (i < cd && !too_many && wm_ok) {
            const int a = i / 6, h = (i % 6) / 3, y = i % 3;
            const real_t *R = L + S::Rc + a * 9;
            real_t loc[3];
#pragma unroll
            for (int x_ = 0; x_ < 3; x_++) {
                const int r = 6 * a + 3 * h + x_;
                real_t v = wacc[r];
#pragma unroll
                for (int j = 0; j < 6; j++) v += (j < k) ? WM[r * WLD + colN + j] * clast[j] : real_t(0.0);
                loc[x_] = v;
            }
            acc = R[y * 3] * loc[0] + R[y * 3 + 1] * loc[1] + R[y * 3 + 2] * loc[2];
        } else if (i < cd && !too_many) {
            acc = -L[S::PC + i];
            _Pragma("unroll 8")
            for (int c = 0; c < M; c++) acc += JbT[i * N + 6 + c] * (L[S::tg + c] + L[S::tt + c] + L[S::tc + c]);
        }
        wr[i] = acc;  // getContactForce(tau_total), wbd.cpp:268-271
    }
    if (dump) { DWBC_SYNC(); dump_contacts_zmp(th, L + S::Pc, L + S::Rc, wr, nc, dump, dl); }
    if (th.tid == 0) {
        io.status[inst] = (st_contact && st_task && st_redis) ? 1 : 0;
        if (diag) {
            diag[DG_ST_CONTACT] = st_contact;
            diag[DG_ST_TASK] = st_task;
            diag[DG_ST_REDIS] = st_redis;
            diag[DG_FAIL_LEVEL] = fail_level;
        }
    }
}

}  // namespace dwbc

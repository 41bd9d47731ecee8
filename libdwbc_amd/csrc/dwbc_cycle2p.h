// dwbc_cycle2p.h -- the fused cycle with TWO wavefronts per robot instance (workgroup of 128 threads), for batches of at most one
// instance per SIMD (B <= 4 x CUs: BASELINE configs[1], 1024 instances on 1024 SIMDs).
//
// Why: with one wave per instance and one instance per SIMD nothing overlaps -- a lone wave issues one instruction per ~5 cycles and
// waits out every LDS round trip; yet two waves sharing a SIMD each run at nearly their solo speed.  So the
// cycle's SIDE CHAINS -- work on LDS-resident thin matrices that does not need the column-per-lane registers of the mass matrix --
// run on a second wave of the same workgroup beside the main chain; the two meet at workgroup barriers:
//
//   main wave (0)                                                  | helper wave (1)
//   kinematics up to the world transforms                          | f* -> LDS
//   -------------------------------------------------------------- B0 (link frames) -------------------------------------------------
//   world inertias; five of the ten components of the composite    | world inertias (again: no hand-over), the other five components;
//   inertias (DPP prefix scan); motion axes S                      |
//   -------------------------------------------------------------- B0a (composite inertias) -----------------------------------------
//   F, CRBA pairs, columns of A -> registers                       | contact frames, J_C, the task Jacobians of every level
//   -------------------------------------------------------------- B1a (J_C, J_t, G in LDS) -----------------------------------------
//   tree-sparse A^-1 sweep, several unrelated pivots per step, the | internal-wrench basis Vb, its Gram matrix, G^-1, VG = Vb G^-1;
//   pivot columns through LDS; the rows of J_C, of J_t of every    | Hb = (J_Cb^T J_Cb)^-1 J_Cb^T (no contact: A_bb^-1, A_jb A_bb^-1)
//   level and G ride through it in the 25 idle lanes               |
//   -------------------------------------------------------------- B1 (Y = J_C A^-1, J_t A^-1, A^-1 G in LDS) ------------------------
//   Lambda_c^-1 = Y J_C^T (MFMA), 12 x 12 SPD inverse, column of   | D = (J_t A^-1) J_C^T, Y G, the task Gram blocks (J_t A^-1) J_t^T (MFMA);
//   Jbar^T = Lambda_c Y                                            | J_t^T / G copied to where phase 4 accumulates
//   -------------------------------------------------------------- B1b, B2 (Jbar^T over J_t^T) ---------------------------------------
//   W^+ applied to the task rows and the gravity pre-vector as     | Lambda_task^-1 = Gram - (D Lambda_c) D^T of every level, JV = Jbar Vb,
//   constrained inverse dynamics: D Lambda_c, cv, E, tau_any =     | S = JV G^-1 JV^T (MFMA + two 6 x 6 products); the three 6 x 6 SPD
//   j_j - J_Cj^T E, (I - P) tau_any (26 MFMA, no product with A);  | inverses in one pass (lane groups); condition verdicts; NwJw = VG X^T
//   P_C, torque_grav_; the joint columns of T1                     |
//   -------------------------------------------------------------- B4 ---------------------------------------------------------------
//   X = J_kt Lambda_t row per lane, null-space chain, wrench maps  | (done)
//   (MFMA), QP cascade on rows kept per lane, outputs              |
//
// The arithmetic of the blocks this kernel shares with dwbc_cycle2.h is the same (same helpers, same order of operations inside a
// block); phases 3 - 4 are its own (see there).  Lean build only (no dump record, no optional paths: the launcher falls back to the
// one-wave kernels for those), hqp = true.
#pragma once
#include "dwbc_cycle2.h"

namespace dwbc {

// LDS map of the paired kernel: blocks that the two waves touch in the same phase never overlap; blocks of different phases do.
// 39 KB for two task levels: four workgroups (eight waves) per CU.
template <int N, int NB, int NLV>
struct Lds4 {
    static constexpr bool compact = true;   // (the code paths shared with Lds3: T1 split into base columns + M-wide home, G^-1 kept, ...)
    static constexpr bool batch_reads = true;
    static constexpr int M = N - 6;
    static constexpr int C = 6 * kMaxActiveContacts;
    static constexpr int K = C - 6;
    static constexpr int T = kMaxTaskDof;
    static constexpr int WLD = 32;
    static constexpr int max2(int a, int b) { return a > b ? a : b; }
    static constexpr int ev(int a) { return (a + 1) & ~1; }
    // ---- head
    static constexpr int tg = 0;
    static constexpr int tt = tg + M;
    static constexpr int tc = tt + M;
    static constexpr int PC = tc + M;
    static constexpr int Rc = PC + C;
    static constexpr int Pc = Rc + kMaxActiveContacts * 9;
    static constexpr int fs = Pc + kMaxActiveContacts * 3;
    static constexpr int flg = fs + kMaxLevels * kMaxTaskDof;  // 4 reals: status words passed between the waves
    static constexpr int q = ev(flg + 4);
    static constexpr int G = q + ev(N + 1);
    static constexpr int c_vec = G + ev(N);
    static constexpr int c_col = c_vec + ev(N);
    static constexpr int hend = c_col + ev(M);
    // ---- link frames (phase 0 .. 1a); then what rode through the A^-1 sweep next to Y (phase 1b .. 3); then the QP scratch (phase 5)
    static constexpr int Rw = hend;
    static constexpr int pw = Rw + NB * 9;
    static constexpr int aw = pw + NB * 3;
    static constexpr int Rw0 = Rw;
    static constexpr int fend = aw + NB * 3;
    // phase 1b -> 3: rows of J_t A^-1 (slot l * T + r, stride N) and A^-1 G (row NLV * T)
    static constexpr int NVS = NLV * T + 1;                    // vectors W^+ is applied to: one slot per task dof of every level + the gravity pre-vector
    static constexpr int ajt = Rw;
    static_assert(NVS * N <= fend - Rw && Rw % 2 == 0, "J_t A^-1 and A^-1 G borrow the link frames");
    static constexpr int VS = ev(NVS);                         // row stride of the per-slot blocks of phase 4
    // ---- long-lived
    static constexpr int c_JC = fend;                          // N x C: J_C, phase 1a .. 4 (the inverse-dynamics form of W^+ reads it)
    static constexpr int NwJw = c_JC + C * N;
    static constexpr int NL2 = NLV < 2 ? 2 : NLV;              // (the U / T1x-sized region hosts the staged mass matrix until phase 4)
    static constexpr int U = NwJw + M * K;                     // levels x (M x T), phase 5
    static constexpr int MS = ev(M);                           // row stride of T1x: even, so that every row is 16-byte aligned (lds_rows_dot)
    static constexpr int UAend = U + NL2 * M * T + NL2 * T * MS;
    static constexpr int c_Hb = UAend;                         // 6 x C: (J_Cb^T J_Cb)^-1 J_Cb^T (A_bb^-1 when no contact is active), phase 1b .. 4
    static constexpr int c_Lt = c_Hb + 6 * C;                  // levels x T x T
    static constexpr int Jtt = c_Lt + NLV * T * T;             // levels x (N x T): J_task transposed, phase 1a .. 2; then Jbar^T
    static constexpr int JbT = Jtt;                            // C x N, phase 2 .. 5
    static_assert(C * N <= NLV * N * T || NLV < 2, "Jbar^T takes the place of J_t^T");
    static constexpr int c_Vb = Jtt + max2(NLV * N * T, C * N);  // M x K
    static constexpr int c_VG = c_Vb + M * K;                  // M x K
    static constexpr int hs = c_VG + M * K;                    // helper's small scratch: 4 x 36 + 72
    static constexpr int hs_size = 4 * 36 + 72;
    static constexpr int ms = hs + hs_size;                    // main's small scratch: sweep column (64) + C x C
    static constexpr int c_s1 = ms;
    static constexpr int c_s2 = ms + max2(C * K, 64);
    static constexpr int c_Lam = c_s2;
    static constexpr int ms_size = max2(C * K, 64) + C * C;
    static constexpr int kin = ms + ms_size;                   // phase-1a scratch of the main wave; afterwards Y, D, T1x, then phase 4 / 5 scratch
    static constexpr int k_Iw = kin;
    static constexpr int k_Ic = k_Iw + NB * 10;
    static constexpr int k_Rl = k_Iw + NB * 3;
    static constexpr int k_anc = k_Ic + NB * 10 - 2 * NB;
    static_assert(k_Rl + NB * 9 <= k_anc, "local rotations must not reach the ancestor indices");
    static constexpr int k_S = k_Ic + NB * 10;
    static constexpr int k_F = k_S + N * 6;
    static constexpr int kin_end = k_F + N * 6;
    static constexpr bool a_overlay = false, a_packed = true;
    static constexpr int k_A = U;                              // packed lower triangle, phase 1a .. 4 (U itself is written in phase 5)
    static_assert(N * (N + 1) / 2 <= (UAend - U), "the staged mass matrix borrows the U / T1x-sized region");
    static constexpr int c_Y = kin;                            // N x C (phase 1b .. 3)
    static constexpr int c_D = kin + C * N;                    // phase 2 .. 3: D = J_t A^-1 J_C^T (NLV * T x C), then Y G (C)
    static constexpr int c_BJ = c_D + (NLV * T + 1) * C;       // phase 2 .. 3: (J_t A^-1) J_t^T of every level (T x T each)
    // (T1x: levels x (T x MS), the joint columns of T1 of every level, phase 3 .. 5 -- behind the task Gram blocks AND behind the
    // slow-route scratch of phase 5, which starts at `kin` as well)
    static constexpr int T1x = ev(max2(c_BJ + NLV * T * T, kin + 2 * T * M + 6 * T * T + 3 * T));
    static_assert(T1x + NLV * T * MS <= kin_end, "Y, D, the task Gram blocks and T1x borrow the phase-1a scratch");
    // phase 4 (main): D Lambda_c of every slot (NVS x C), E = (D Lambda_c)^T + Hb^T cv (C x VS), w = Vb^T tau (K x VS) over Y (dead after
    // phase 2); the torques tau (M x VS) and the base residuals cv (6 x VS) accumulate in place over the staged mass matrix (dead after
    // phase 1b), where the helper leaves their starting values J_t^T / G in phase 2
    static constexpr int p4_dl = kin;
    static constexpr int p4_e = p4_dl + ev(NVS * C);
    static constexpr int p4_w = p4_e + C * VS;
    static constexpr int p4_ts = U;
    static constexpr int p4_cv = p4_ts + M * VS;
    static_assert(kin % 2 == 0 && U % 2 == 0, "16-byte aligned rows of the phase-4 blocks (lds_rows_axpy)");
    static_assert(p4_w + K * VS <= c_D, "phase-4 scratch stays below D and the task Gram blocks (the helper reads them in parallel)");
    static_assert(p4_cv + 6 * VS <= UAend, "torques and base residuals borrow the place of the staged mass matrix");
    static constexpr int c_QW = kin;                           // Q (slow route)
    static constexpr int c_QWp = c_QW + T * M;                 // Q W^+ (slow route)
    static constexpr int c_Pi = c_QWp + T * M;
    static constexpr int c_Z = c_Pi + T * T;
    static constexpr int c_s2b = c_Z + T * T;
    static constexpr int cod_Q = c_s2b + T * T, cod_v = cod_Q + T * T, cod_G = cod_v + 3 * T, cod_T = cod_G + T * T;
    static_assert(cod_T + T * T <= T1x, "slow-route scratch stays below T1x");
    static constexpr int xl(int lv) { return U; }
    // QP scratch of phase 5, over the link frames
    static constexpr int t_base = Rw;                          // (f* + f*_qp of the committed levels: NLV x 6)
    static constexpr int wm = ev(t_base + NLV * 6);
    static constexpr int t_fv = wm + C * WLD;
    static constexpr int qp_V = t_fv + C;
    static constexpr int p5_end = qp_V + kQpLd;
    static_assert(p5_end <= fend, "QP scratch over the link frames");
    static constexpr int c_T1 = 0;                             // (base columns of T1: not formed in this kernel)
    static constexpr int total = kin_end;
    static constexpr int total_bytes = total * (int)sizeof(real_t) + 64;
    // names of the other maps that shared helpers mention but this kernel does not use
    static constexpr int FNl = NwJw, comp = flg, Jcm = 0, Pt = 0, c_Gi = 0, T1r = T1x, c_Q = T1x, c_Jt = Jtt, t_F = wm, t_s1 = wm, jk = 0;
};

#if defined(DWBC_HOST_EMU)
#define DWBC_PAIR_BARRIER(i) ((void)0)
#elif defined(DWBC_STAGE_TIMERS)
// diagnostic build: when each wave reaches barrier i and when the main wave leaves it (shader cycles since kernel start)
#define DWBC_PAIR_BARRIER(i)                                                                      \
    do {                                                                                          \
        if (diag && th.tid == 0) diag[(is_main ? DG_TIME : DG_FTIME) + (i)] = (int)(clock64() - t_start_); \
        __syncthreads();                                                                          \
        if (diag && th.tid == 0 && is_main) diag[DG_FTIME + 8 + (i)] = (int)(clock64() - t_start_); \
    } while (0)
#else
#define DWBC_PAIR_BARRIER(i) __syncthreads()
#endif
#if defined(DWBC_HOST_EMU)
#define DWBC_PAIR_BARRIER_X() ((void)0)
#else
#define DWBC_PAIR_BARRIER_X() __syncthreads()
#endif
#if defined(DWBC_STAGE_TIMERS) && !defined(DWBC_HOST_EMU)
#define DWBC_PSTAMP(i) do { DWBC_SYNC(); if (diag && th.tid == 0) diag[DG_FTIME + (i)] = (int)(clock64() - t_start_); } while (0)
// stamps inside the register-heavy phases 1-4: a build with all of them spills in the sweeps, so each is compiled in only when its bit
// of DWBC_PMASK is set (make experiment VARIANT=.. XFLAGS="-DDWBC_STAGE_TIMERS -DDWBC_PMASK=0x..ull"; bit i - 41)
#ifndef DWBC_PMASK
#define DWBC_PMASK 0ull
#endif
#define DWBC_PSTAMP_M(i) do { if constexpr (((DWBC_PMASK) >> ((i) - 41)) & 1ull) DWBC_PSTAMP(i); } while (0)
#else
#define DWBC_PSTAMP(i) ((void)0)
#define DWBC_PSTAMP_M(i) ((void)0)
#endif

// D = C0 + A B for the small dense products of phases 3 - 4 on the matrix cores: A is MR x KV, B is KV x NCV (NCV <= 16), tiles of 16
// rows, reduction blocks of 4 (v_mfma_f64_16x16x4_f64; operand layout as in wrench_maps, dwbc_cycle2.h: A[i = l & 15][k = l >> 4],
// B[k = l >> 4][j = l & 15], D[i = (l >> 4) + 4 r][j = l & 15]).
//   fa(tc, sc, i, k) / fb(sc, k, j): the operand entries, called with indices INSIDE the matrices (the padding of the last tile / block
//     is clamped and masked here, and only in the tile / block that has any): tc, sc are std::integral_constant tile and block numbers,
//     i = 16 tc + (l & 15), k = 4 sc + (l >> 4) wherever the tile / block is whole -- so an accessor written as base[lane part] +
//     constant gets its constant folded into the instruction's offset field, and one that needs index arithmetic (the packed
//     symmetric mass matrix) can choose its form per block at compile time;
//   fc(i, j): the value the accumulation starts from;  fstore(i, j, v): takes every entry of the result (i < MR, j < NCV).
// Every operand is loaded before the first MFMA is issued (one LDS round trip per product, not one per block -- the dependent
// load -> wait -> MFMA steps of a lone wave cost ~500 cycles per block).  SYNC_STORE: a wave-level fence between the last read and
// the first store, for results written over their own operands.  One wave.  The host emulation (and nothing else) takes plain loops.
template <int MR, int NCV, int KV, bool SYNC_STORE, class FA, class FB, class FS, class FC>
DWBC_WDEV void wave_gemm(FA fa, FB fb, FS fstore, FC fc) {
    constexpr int MT = (MR + 15) / 16, KB = (KV + 3) / 4;
    static_assert(NCV <= 16, "one column tile");
#if !defined(DWBC_HOST_EMU)
    static_assert(sizeof(real_t) == 8, "fp64 build");
    typedef double g_d4 __attribute__((ext_vector_type(4)));
    const int lane = (int)(threadIdx.x & 63u), li = lane & 15, lk = lane >> 4;
    const int jc = (NCV < 16 && li >= NCV) ? NCV - 1 : li;  // (columns beyond NCV repeat the last one: computed, not stored)
    double av[MT][KB], bv[KB];
    static_for<0, KB>([&](auto sc) {
        constexpr int s_ = decltype(sc)::value;
        constexpr bool kwhole = 4 * s_ + 3 < KV;
        const int kk = kwhole ? 4 * s_ + lk : (4 * s_ + lk < KV ? 4 * s_ + lk : KV - 1);
        const bool kin = kwhole || 4 * s_ + lk < KV;
        const double b_ = fb(sc, kk, jc);
        bv[s_] = kin ? b_ : 0.0;
        static_for<0, MT>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr bool rwhole = 16 * t + 15 < MR;
            const int ic = rwhole ? 16 * t + li : (16 * t + li < MR ? 16 * t + li : MR - 1);  // (rows beyond MR repeat the last one)
            const double a_ = fa(tc, sc, ic, kk);
            av[t][s_] = kin ? a_ : 0.0;
        });
    });
    g_d4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = 16 * t + lk + 4 * r;
            acc[t][r] = fc(i < MR ? i : MR - 1, jc);
        }
#pragma unroll
    for (int s_ = 0; s_ < KB; s_++)
#pragma unroll
        for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t][s_], bv[s_], acc[t], 0, 0, 0);
    if (SYNC_STORE) DWBC_SYNC();
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = 16 * t + lk + 4 * r;
            if (i < MR && li < NCV) fstore(i, li, acc[t][r]);
        }
#else
    real_t tmp[MR][NCV];
    for (int i = 0; i < MR; i++)
        for (int j = 0; j < NCV; j++) tmp[i][j] = fc(i, j);
    static_for<0, KB>([&](auto sc) {
        constexpr int s_ = decltype(sc)::value;
        static_for<0, MT>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            for (int li = 0; li < 16; li++)
                for (int lk = 0; lk < 4; lk++) {
                    const int i = 16 * t + li, kk = 4 * s_ + lk;
                    if (i >= MR || kk >= KV) continue;
                    const real_t a_ = fa(tc, sc, i, kk);
                    for (int j = 0; j < NCV; j++) tmp[i][j] += a_ * fb(sc, kk, j);
                }
        });
    });
    for (int i = 0; i < MR; i++)
        for (int j = 0; j < NCV; j++) fstore(i, j, tmp[i][j]);
#endif
}
template <int MR, int NCV, int KV, bool SYNC_STORE, class FA, class FB, class FS>
DWBC_WDEV void wave_gemm(FA fa, FB fb, FS fstore) {
    wave_gemm<MR, NCV, KV, SYNC_STORE>(fa, fb, fstore, [](int, int) { return real_t(0.0); });
}

// Lambda_c^-1 -> Lambda_c for two active contacts, IN PLACE (M: 12 x 12, dense): spd_inverse_small's Jacobi-scaled symmetric sweep
// (dwbc_cycle2.h: same scaling, same pivots, same arithmetic per row update) with the pivot column fed through LDS instead of
// v_readlane -- per pivot one ds_write_b64 and six broadcast ds_read_b128 against 24 v_readlane + 12 s_nop.  The matrix sits in
// registers during the sweep, so its own LDS block serves as the column buffer; scl: 12 doubles for the scale factors.
template <class R>
DWBC_WDEV int spd_inverse12_lds(R *Mx, R *Out, R *scl) {
#if defined(DWBC_HOST_EMU)
    return spd_inverse_small(Mx, 12, 12, Out, 12, scl);
#else
    static_assert(sizeof(R) == 8, "fp64 build");
    const int lane = (int)(threadIdx.x & 63u);
    DWBC_SYNC();
    double s[12], dg, dsc;
    {
        const int col = lane < 12 ? lane : 0;
        const double a = Mx[col * 12 + col];
        int e2 = 0;
        (void)frexp(a > 0.0 ? a : 1.0, &e2);
        dsc = (lane < 12 && a > 0.0) ? ldexp(1.0, -(e2 >> 1)) : 1.0;
        if (lane < 12) scl[lane] = dsc;
    }
    DWBC_SYNC();
    {
        const int col = lane < 12 ? lane : 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const double v_ = Mx[i * 12 + col] * scl[i] * dsc;
            s[i] = lane < 12 ? v_ : 0.0;
        }
        const double d_ = Mx[col * 12 + col] * dsc * dsc;
        dg = lane < 12 ? d_ : 1.0;
    }
    DWBC_SYNC();  // (the matrix is in registers: its block is the column buffer now)
    int ok = 1;
    {
        DWBC_LANE_OPAQUE(lp);
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = (i == lp) ? dg - 1.0 : s[i];
    }
    double dg2 = dg - 2.0;
    lds_pivot<12, 11, 0>(s, dg2, ok, (unsigned)(size_t)Mx, lane);
    ok = DWBC_FLAG_UNIFORM(ok);
    DWBC_SYNC();
    if (lane < 12) {
        DWBC_LANE_OPAQUE(le);
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const double v_ = (i == le) ? -dg2 : -s[i];
            Out[i * 12 + lane] = v_ * scl[i] * dsc;
        }
    }
    DWBC_SYNC();
    return ok;
#endif
}

// wave: 0 = main, 1 = helper (device); -1 = both roles one after the other in one thread of control (host emulation)
template <int N, int NB, int NLV, int NT, class Topo>
DWBC_DEV void cycle_instance_v2p(int wave, Thr th, const Setup &su, const BatchIO &io, int inst, real_t *L) {
    using S = Lds4<N, NB, NLV>;
    constexpr int M = S::M, C = S::C, T = S::T;
    constexpr bool kTree = !std::is_same<Topo, TopoGeneric>::value;
    static_assert(kTree, "the paired kernel is built for a constant kinematic tree");
    static_assert(kMaxActiveContacts == 2, "two 6D contacts");
    DWBC_LANE_DECL;
    const bool is_main = wave <= 0, is_help = wave != 0;
    const int nb = NB;
    const real_t *body = io.body;
    const int *topo = io.topo;
    const io_t *qin = io.q + (size_t)inst * (N + 1);
    int *diag = io.diag ? io.diag + (size_t)inst * DG_COUNT : nullptr;
    DWBC_STAMP_INIT();
    // ---- contact flags: both waves (uniform loads)
    const unsigned char *fl = io.flags + (size_t)inst * su.n_contacts;
    int act_c[kMaxActiveContacts] = {0, 0};
    int nc = 0, nflag = 0;
    for (int i = 0; i < su.n_contacts; i++) {
        if (fl[i] && nc < kMaxActiveContacts) act_c[nc++] = i;
        nflag += fl[i] ? 1 : 0;
    }
    const bool too_many = nflag > kMaxActiveContacts;
    const int cd = 6 * nc, k = cd > 6 ? cd - 6 : 0;
    real_t *JCt = L + S::c_JC, *Yt = L + S::c_Y, *Lam = L + S::c_Lam, *JbT = L + S::JbT, *Vb = L + S::c_Vb, *VG = L + S::c_VG;
    real_t *flg = L + S::flg;  // [0] helper's contact status, [1] fast-route mask of the levels, [2] (free)

    PLA(real_t, s, N);  // main wave: column `lane` of A -> A^-1 -> A^-1 N_c
    PL(real_t, dg);
    int st_contact = 1;

    // ================= phase 0: kinematics up to the world transforms (main) =================
    if (is_main) {
        for (int i = th.tid; i < N + 1; i += NT) L[S::q + i] = (real_t)qin[i];
        for (int i = th.tid; i < 3 * M; i += NT) L[S::tg + i] = real_t(0.0);
        DWBC_SYNC();
    }
    constexpr int kPairRounds = (N * (N + 1) / 2 + NT - 1) / NT;
    const int npair = topo[3 * nb];
    int pairw[kPairRounds];
    if (is_main) {
#pragma unroll
        for (int r = 0; r < kPairRounds; r++) pairw[r] = (r * NT + th.tid < npair) ? topo[3 * nb + 1 + r * NT + th.tid] : 0;
    }
    real_t *Rw = L + S::Rw, *pw = L + S::pw, *aw = L + S::aw, *Rl = L + S::k_Rl;
    if (is_main) {
        const real_t *q = L + S::q;
        for (int i = th.tid; i < nb; i += NT) {
            const real_t *bd = body + i * kBodyStride;
            if (i == 0) {
                const real_t x = q[3], y = q[4], z = q[5], w = q[N];
                real_t *R = Rw;
                R[0] = 1 - 2 * y * y - 2 * z * z; R[1] = 2 * x * y - 2 * w * z; R[2] = 2 * x * z + 2 * w * y;
                R[3] = 2 * x * y + 2 * w * z; R[4] = 1 - 2 * x * x - 2 * z * z; R[5] = 2 * y * z - 2 * w * x;
                R[6] = 2 * x * z - 2 * w * y; R[7] = 2 * y * z + 2 * w * x; R[8] = 1 - 2 * x * x - 2 * y * y;
                pw[0] = q[0]; pw[1] = q[1]; pw[2] = q[2];
            } else {
                const real_t ax = bd[BF_AXIS], ay = bd[BF_AXIS + 1], az = bd[BF_AXIS + 2];
                real_t sn, cs;
                sincos_r(q[6 + i - 1], &sn, &cs);
                const real_t c1 = real_t(1.0) - cs;
                real_t Rj[9];
                Rj[0] = cs + ax * ax * c1; Rj[1] = ax * ay * c1 - az * sn; Rj[2] = ax * az * c1 + ay * sn;
                Rj[3] = ay * ax * c1 + az * sn; Rj[4] = cs + ay * ay * c1; Rj[5] = ay * az * c1 - ax * sn;
                Rj[6] = az * ax * c1 - ay * sn; Rj[7] = az * ay * c1 + ax * sn; Rj[8] = cs + az * az * c1;
                for (int a = 0; a < 3; a++)
                    for (int b = 0; b < 3; b++)
                        Rl[i * 9 + a * 3 + b] = bd[BF_RT + a * 3] * Rj[b] + bd[BF_RT + a * 3 + 1] * Rj[3 + b] + bd[BF_RT + a * 3 + 2] * Rj[6 + b];
            }
        }
        // world transforms by pointer jumping (dwbc_cycle2_stage0.inc)
        {
            constexpr int md = Topo::maxdepth;
            int rounds = 0;
            while ((1 << rounds) < md + 1) rounds++;
            const bool odd = rounds & 1;
            real_t *Rc_ = odd ? Rl : Rw, *Rn = odd ? Rw : Rl;
            real_t *pc_ = odd ? L + S::k_Iw : pw, *pn = odd ? pw : L + S::k_Iw;
            real_t *ac = L + S::k_anc, *an_ = L + S::k_anc + NB;
            DWBC_SYNC();
            for (int i = th.tid; i < nb; i += NT) {
                const real_t *bd = body + i * kBodyStride;
                real_t Ri[9], pi[3];
                for (int a = 0; a < 9; a++) Ri[a] = (i == 0) ? Rw[a] : Rl[i * 9 + a];
                for (int a = 0; a < 3; a++) pi[a] = (i == 0) ? pw[a] : bd[BF_PT + a];
                const real_t an = (i == 0) ? -real_t(1.0) : (real_t)topo[i];
                for (int a = 0; a < 9; a++) Rc_[i * 9 + a] = Ri[a];
                for (int a = 0; a < 3; a++) pc_[i * 3 + a] = pi[a];
                ac[i] = an;
            }
            for (int r = 0; r < rounds; r++) {
                DWBC_SYNC();
                for (int i = th.tid; i < nb; i += NT) {
                    const int an = (int)ac[i];
                    const int aa = an < 0 ? 0 : an;
                    real_t Ri[9], pi[3], Ra[9], pa[3];
                    for (int a = 0; a < 9; a++) { Ri[a] = Rc_[i * 9 + a]; Ra[a] = Rc_[aa * 9 + a]; }
                    for (int a = 0; a < 3; a++) { pi[a] = pc_[i * 3 + a]; pa[a] = pc_[aa * 3 + a]; }
                    const real_t a2 = ac[aa];
                    real_t Ro[9], po[3];
                    for (int a = 0; a < 3; a++) {
                        for (int c = 0; c < 3; c++) Ro[a * 3 + c] = Ra[a * 3] * Ri[c] + Ra[a * 3 + 1] * Ri[3 + c] + Ra[a * 3 + 2] * Ri[6 + c];
                        po[a] = pa[a] + Ra[a * 3] * pi[0] + Ra[a * 3 + 1] * pi[1] + Ra[a * 3 + 2] * pi[2];
                    }
                    for (int a = 0; a < 9; a++) Rn[i * 9 + a] = an < 0 ? Ri[a] : Ro[a];
                    for (int a = 0; a < 3; a++) pn[i * 3 + a] = an < 0 ? pi[a] : po[a];
                    an_[i] = an < 0 ? -real_t(1.0) : a2;
                }
                { real_t *t_ = Rc_; Rc_ = Rn; Rn = t_; t_ = pc_; pc_ = pn; pn = t_; t_ = ac; ac = an_; an_ = t_; }
            }
        }
        DWBC_SYNC();
        // world joint axes (the point Jacobians of the helper need them: before the barrier)
        for (int i = th.tid; i < nb; i += NT) {
            const real_t *bd = body + i * kBodyStride;
            const real_t *R = Rw + i * 9;
            for (int a = 0; a < 3; a++) aw[i * 3 + a] = R[a * 3] * bd[BF_AXIS] + R[a * 3 + 1] * bd[BF_AXIS + 1] + R[a * 3 + 2] * bd[BF_AXIS + 2];
        }
    }
    if (is_help) {
        // f* of every level (the lean build has no on-device task reference: the SetTaskSpace values)
        const io_t *fin = io.fstar + (size_t)inst * su.fstar_total;
        for (int i = th.tid; i < su.fstar_total; i += NT) L[S::fs + i] = (real_t)fin[i];
        if (th.tid == 0) { flg[0] = real_t(1.0); flg[1] = real_t(0.0); }
    }
    DWBC_PAIR_BARRIER(0);  // ---- B0: Rw, pw, aw

    // ================= phase 1 =================
    // world inertias and composite inertias: both waves (each forms the world inertias for itself -- identical values to identical
    // places -- and takes five of the ten components through the prefix scan; the halves meet at B0a)
    real_t *Iw = L + S::k_Iw, *Icm = L + S::k_Ic;
    auto world_inertias = [&]() {
        for (int i = th.tid; i < nb; i += NT) {
            const real_t *bd = body + i * kBodyStride;
            const real_t *R = Rw + i * 9;
            const real_t m = bd[BF_MASS];
            real_t r[3];
            for (int a = 0; a < 3; a++)
                r[a] = pw[i * 3 + a] + R[a * 3] * bd[BF_COM] + R[a * 3 + 1] * bd[BF_COM + 1] + R[a * 3 + 2] * bd[BF_COM + 2] - pw[a];
            const real_t Ic[9] = {bd[BF_ICOM], bd[BF_ICOM + 1], bd[BF_ICOM + 2], bd[BF_ICOM + 1], bd[BF_ICOM + 3],
                                  bd[BF_ICOM + 4], bd[BF_ICOM + 2], bd[BF_ICOM + 4], bd[BF_ICOM + 5]};
            real_t Tm[9];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) Tm[a * 3 + b] = R[a * 3] * Ic[b] + R[a * 3 + 1] * Ic[3 + b] + R[a * 3 + 2] * Ic[6 + b];
            const real_t rr2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
            real_t *o = Iw + i * 10;
            o[0] = m;
            o[1] = m * r[0]; o[2] = m * r[1]; o[3] = m * r[2];
            int c = 4;
            for (int a = 0; a < 3; a++)
                for (int b = a; b < 3; b++) {
                    real_t v = Tm[a * 3] * R[b * 3] + Tm[a * 3 + 1] * R[b * 3 + 1] + Tm[a * 3 + 2] * R[b * 3 + 2];
                    v += m * ((a == b ? rr2 : real_t(0.0)) - r[a] * r[b]);
                    o[c++] = v;
                }
        }
        DWBC_SYNC();
    };
    auto composite_part = [&](auto c0c, auto c1c) {
            // composite inertia of the subtree [b, b + len_b) (bodies are numbered depth first): inclusive prefix sums over the body order
            // in registers (DPP), Ic[b] = P[b + len_b - 1] - P[b - 1].  Everything is expressed about the pelvis origin, where the smallest
            // subtree (a wrist link, m r^2 ~ 0.1) is within 1e3 of the total: the difference keeps 13 digits.  (The window sums by doubling
            // of the one-wave kernels -- six rounds through LDS -- were 7.3 k of this wave's 17 k cycles before the A^-1 sweep.)
            constexpr int C0 = decltype(c0c)::value, C1 = decltype(c1c)::value, NCP = C1 - C0;
            PLA(real_t, pf, NCP);
            PL(int, len);
            LANES {
                const int bi = lane < nb ? lane : 0;
                LV(len) = lane < nb ? topo[2 * nb + bi] : 1;
#pragma unroll
                for (int c = 0; c < NCP; c++) {
                    const real_t v_ = Iw[bi * 10 + C0 + c];
                    LV(pf)[c] = lane < nb ? v_ : real_t(0.0);
                }
            }
#pragma unroll
            for (int c = 0; c < NCP; c++) WAVE_PREFIX_A(pf, c);
            LANES {
                int e_ = lane + LV(len) - 1;
                e_ = e_ < 63 ? e_ : 63;
                const int s_ = lane > 0 ? lane - 1 : 0;
#pragma unroll
                for (int c = 0; c < NCP; c++) {
                    const real_t hi_ = SHFLA(pf, c, e_), lo_ = SHFLA(pf, c, s_);
                    if (lane < nb) Icm[lane * 10 + C0 + c] = hi_ - (lane > 0 ? lo_ : real_t(0.0));
                }
            }
            DWBC_SYNC();
    };
    if (is_main) {
        world_inertias();
        DWBC_PSTAMP_M(51);  // world inertias
        composite_part(std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{});
    }
    if (is_help) {
        world_inertias();
        composite_part(std::integral_constant<int, 5>{}, std::integral_constant<int, 10>{});
    }
    real_t *Sm = L + S::k_S, *Fm = L + S::k_F;
    if (is_main) {
        // motion axes S (frames only: before the halves meet), then F, CRBA, A -> registers, A^-1  (dwbc_cycle2_stage0.inc)
        for (int j = th.tid; j < N; j += NT) {
            real_t w[3] = {0, 0, 0}, v[3] = {0, 0, 0};
            if (j < 3) {
                v[j] = real_t(1.0);
            } else if (j < 6) {
                for (int a = 0; a < 3; a++) w[a] = Rw[a * 3 + (j - 3)];
            } else {
                const int b = j - 5;
                for (int a = 0; a < 3; a++) w[a] = aw[b * 3 + a];
                const real_t d0 = pw[b * 3] - pw[0], d1 = pw[b * 3 + 1] - pw[1], d2 = pw[b * 3 + 2] - pw[2];
                v[0] = d1 * w[2] - d2 * w[1];
                v[1] = d2 * w[0] - d0 * w[2];
                v[2] = d0 * w[1] - d1 * w[0];
            }
            for (int a = 0; a < 3; a++) { Sm[j * 6 + a] = w[a]; Sm[j * 6 + 3 + a] = v[a]; }
        }
        DWBC_SYNC();
    }
    DWBC_PAIR_BARRIER_X();  // ---- B0a: all ten components of the composite inertias
    if (is_main) {
        DWBC_PSTAMP_M(52);  // composite inertias, S
        for (int j = th.tid; j < N; j += NT) {
            const int b = j < 6 ? 0 : j - 5;
            const real_t *I = Icm + b * 10;
            const real_t *sv = Sm + j * 6;
            const real_t m = I[0], h0 = I[1], h1 = I[2], h2 = I[3];
            const real_t w0 = sv[0], w1 = sv[1], w2 = sv[2], v0 = sv[3], v1 = sv[4], v2 = sv[5];
            Fm[j * 6 + 0] = I[4] * w0 + I[5] * w1 + I[6] * w2 + (h1 * v2 - h2 * v1);
            Fm[j * 6 + 1] = I[5] * w0 + I[7] * w1 + I[8] * w2 + (h2 * v0 - h0 * v2);
            Fm[j * 6 + 2] = I[6] * w0 + I[8] * w1 + I[9] * w2 + (h0 * v1 - h1 * v0);
            Fm[j * 6 + 3] = m * v0 + (w1 * h2 - w2 * h1);
            Fm[j * 6 + 4] = m * v1 + (w2 * h0 - w0 * h2);
            Fm[j * 6 + 5] = m * v2 + (w0 * h1 - w1 * h0);
        }
        DWBC_SYNC();
        DWBC_PSTAMP_M(53);  // forces F
        real_t *A = L + S::k_A;  // lower triangle, row-packed: (i, j <= i) at i (i + 1) / 2 + j
        for (int idx = th.tid; idx < N * (N + 1) / 2; idx += NT) A[idx] = real_t(0.0);
        DWBC_SYNC();
#pragma unroll
        for (int r = 0; r < kPairRounds; r++) {
            if (r * NT + th.tid < npair) {
                const int j = pairw[r] >> 8, kk = pairw[r] & 255;
                const real_t *f = Fm + j * 6, *sv = Sm + kk * 6;
                A[j * (j + 1) / 2 + kk] = sv[0] * f[0] + sv[1] * f[1] + sv[2] * f[2] + sv[3] * f[3] + sv[4] * f[4] + sv[5] * f[5];
            }
        }
        DWBC_SYNC();
        DWBC_PSTAMP_M(54);  // mass matrix pairs
        LANES {
            const int col = lane < N ? lane : 0;
            const int cbase = col * (col + 1) / 2;
#pragma unroll
            for (int i = 0; i < N; i++) LV(s)[i] = (lane < N) ? A[i >= col ? i * (i + 1) / 2 + col : cbase + i] : real_t(0.0);
            LV(dg) = (lane < N) ? A[cbase + col] : real_t(1.0);
            if (lane < N) L[S::G + lane] = kGrav * A[col >= 2 ? cbase + 2 : 3 + col];  // G_ = 9.81 A[2,:] (dwbc.cpp:358)
        }
        DWBC_PSTAMP_M(41);  // CRBA done, columns of A in registers
    }
    if (is_help) {
        // contact frames, J_C and the task Jacobians of every level first: they ride through the A^-1 sweep (B1a)
        for (int a = 0; a < nc; a++) {
            const int ci = act_c[a], link = su.c_link[ci];
            const real_t *R = Rw + link * 9;
            for (int r = th.tid; r < 12; r += NT) {
                if (r < 9) L[S::Rc + a * 9 + r] = R[r];
                else {
                    const int x = r - 9;
                    L[S::Pc + a * 3 + x] = pw[link * 3 + x] + R[x * 3] * su.c_point[ci][0] + R[x * 3 + 1] * su.c_point[ci][1] + R[x * 3 + 2] * su.c_point[ci][2];
                }
            }
        }
        for (int idx = th.tid; idx < C * N; idx += NT) JCt[idx] = real_t(0.0);
        DWBC_SYNC();
        for (int a = 0; a < nc; a++)
            point_jacobian<N, NB, NT>(th, Rw, pw, aw, topo, nb, su.c_link[act_c[a]], L + S::Pc + a * 3, JCt, 1, 6 * a, 6, 0, C);
        DWBC_SYNC();
        for (int lv = 0; lv < su.n_levels; lv++) {
            real_t *Jtt = L + S::Jtt + lv * N * T;
            for (int idx = th.tid; idx < T * N; idx += NT) Jtt[idx] = real_t(0.0);
            DWBC_SYNC();
            int row = 0;
            for (int li = 0; li < su.t_nlinks[lv]; li++) {
                const int mode = su.t_mode[lv][li], link = su.t_link[lv][li];
                real_t pl[3] = {0, 0, 0};
                if ((mode == TASK_LINK_6D_COM_FRAME || mode == TASK_LINK_POSITION_COM_FRAME) && link < nb)
                    for (int a = 0; a < 3; a++) pl[a] = body[link * kBodyStride + BF_COM + a];
                else if (mode == TASK_LINK_6D_CUSTOM_FRAME || mode == TASK_LINK_POSITION_CUSTOM_FRAME)
                    for (int a = 0; a < 3; a++) pl[a] = su.t_point[lv][li][a];
                const int rsel = mode <= TASK_LINK_6D_CUSTOM_FRAME ? 0 : (mode <= TASK_LINK_POSITION_CUSTOM_FRAME ? 1 : 2);
                const real_t *R = Rw + link * 9;
                real_t P[3];
                for (int a = 0; a < 3; a++) P[a] = pw[link * 3 + a] + R[a * 3] * pl[0] + R[a * 3 + 1] * pl[1] + R[a * 3 + 2] * pl[2];
                point_jacobian<N, NB, NT>(th, Rw, pw, aw, topo, nb, link, P, Jtt, 1, row, rsel == 0 ? 6 : 3, rsel, T);
                row += rsel == 0 ? 6 : 3;
            }
            DWBC_SYNC();
        }
    }
    DWBC_PAIR_BARRIER_X();  // ---- B1a: J_C, J_t, G in LDS; the columns of A in the main wave's registers
    real_t *AJt = L + S::ajt;
    constexpr int NVS = S::NVS, VS = S::VS, GV = NLV * T;  // vector slots: l * T + r for task dof r of level l, GV = the gravity pre-vector
    if (is_main) {
        // The A^-1 sweep uses N = 39 of the 64 lanes; the other 25 carry right-hand sides through the same pivots (a Gauss-Jordan
        // sweep of [A | X] leaves A^-1 X in the extra columns, with the FMAs every non-pivot lane executes anyway; a row outside the
        // pivot's relatives would receive c_i * h with c_i == 0 there as well): lanes N .. N+11 the rows of J_C (-> Y = J_C A^-1,
        // wbd.cpp:113), the next NLV * T lanes the rows of J_t (slot l * T + r -> J_t A^-1), one more G (-> A^-1 G).  Nothing else of
        // A^-1 is used in this kernel: every later product is written in terms of these vectors.
        static_assert(N + C + NVS <= 64, "rhs lanes");
        LANES {
            if (lane >= N) {
                const int jj = lane - N, jt = jj - C;
                const int lvl = (jt >= 0 && jt < GV) ? jt / T : 0, r_ = (jt >= 0 && jt < GV) ? jt - lvl * T : 0;
                const bool isc = jj < C, ist = jt >= 0 && jt < GV && lvl < su.n_levels && r_ < su.t_dof[lvl], isg = jt == GV;
                const real_t *bp = isc ? JCt + jj : (isg ? L + S::G : L + S::Jtt + lvl * N * T + r_);
                const int st = isc ? C : (isg ? 1 : T);
#pragma unroll
                for (int i = 0; i < N; i++) {
                    const real_t v_ = bp[i * st];
                    LV(s)[i] = (isc || ist || isg) ? v_ : real_t(0.0);
                }
                LV(dg) = real_t(1.0);
            }
        }
        DWBC_SYNC();
#ifndef DWBC_SWEEP_W
#define DWBC_SWEEP_W 4
#endif
        static_assert(DWBC_SWEEP_W * 2 * ((N + 1) / 2) <= S::ms_size, "pivot columns of one step in the main wave's small scratch");
        if (!sweep_inverse_tree_lds_multi<Topo, N, DWBC_SWEEP_W>(s, dg, L + S::c_s1)) st_contact = 0;  // A_inv (dwbc.cpp:307), pivot column through LDS
        DWBC_SYNC();
        if (too_many) st_contact = 0;
        LANES {
            if (lane >= N && lane - N < C + NVS) {  // (negated by the sweep's epilogue like the matrix lanes)
                const int jj = lane - N;
                real_t *bp = jj < C ? Yt + jj : AJt + (jj - C) * N;
                const int st = jj < C ? C : 1;
#pragma unroll
                for (int i = 0; i < N; i++) bp[i * st] = -LV(s)[i];
            }
        }
        DWBC_SYNC();
    }
    real_t *Hb = L + S::c_Hb;
    const real_t *Ap = L + S::k_A;  // staged mass matrix: lower triangle, row-packed -- alive until phase 4 in this kernel
    if (is_help) {
        if (k > 0) {
            // internal-wrench basis and its Gram algebra (beside the sweep)
            constexpr int K6 = 6;
            internal_wrench_basis<N, NT>(th, L + S::Pc, JCt, Vb);
            DWBC_SYNC();
            real_t *Gi = L + S::hs + 36;
            mm_tn<NT>(th, Gi, K6, Vb, K6, Vb, K6, K6, M, K6);                      // G
            DWBC_SYNC();
            spd_inverse_small(Gi, K6, K6, Gi, K6, L + S::hs + 144);               // G^-1
            mm_nn<NT>(th, VG, K6, Vb, K6, Gi, K6, M, K6, K6);                       // VG = Vb G^-1
            DWBC_SYNC();
        }
        // Hb: what turns the contact rows into a base acceleration and a base wrench into contact forces in phase 4:
        //   contacts:     Hb = (J_Cb^T J_Cb)^-1 J_Cb^T  (6 x cd; J_Cb = the base columns of J_C, full column rank)
        //   no contact:   Hb[:, 0:6] = A_bb^-1
        {
            real_t *Gb = L + S::hs + 72, *Gbi = L + S::hs + 108;
            for (int idx = th.tid; idx < 36; idx += NT) {
                const int x = idx / 6, y = idx - x * 6;
                real_t acc = real_t(0.0);
                if (nc > 0) {
#pragma unroll
                    for (int p = 0; p < C; p++) acc += (p < cd) ? JCt[x * C + p] * JCt[y * C + p] : real_t(0.0);
                } else {
                    acc = x >= y ? Ap[x * (x + 1) / 2 + y] : Ap[y * (y + 1) / 2 + x];
                }
                Gb[idx] = acc;
            }
            DWBC_SYNC();
            if (!spd_inverse_small(Gb, 6, 6, Gbi, 6, L + S::hs + 144)) { if (th.tid == 0) flg[0] = real_t(0.0); }
            DWBC_SYNC();
            for (int idx = th.tid; idx < 6 * C; idx += NT) {
                const int x = idx / C, p = idx - x * C;
                real_t acc = real_t(0.0);
                if (nc > 0) {
#pragma unroll
                    for (int y = 0; y < 6; y++) acc += (p < cd) ? Gbi[x * 6 + y] * JCt[y * C + p] : real_t(0.0);
                } else {
                    acc = p < 6 ? Gbi[x * 6 + p] : real_t(0.0);
                }
                Hb[idx] = acc;
            }
            if (nc == 0) {  // A_jb A_bb^-1 (M x 6) in VG's place (no contact: no internal wrench basis)
                for (int idx = th.tid; idx < M * 6; idx += NT) {
                    const int i = idx / 6, x = idx - i * 6;
                    real_t acc = real_t(0.0);
#pragma unroll
                    for (int y = 0; y < 6; y++) acc += Ap[(6 + i) * (7 + i) / 2 + y] * Gbi[y * 6 + x];
                    VG[idx] = acc;
                }
            }
            DWBC_SYNC();
        }
    }
    DWBC_PAIR_BARRIER(1);  // ---- B1: Y, J_t A^-1, A^-1 G in LDS; J_C, Vb, VG, Hb

    // ================= phase 2: Lambda_c, Jbar^T (main) | D, Y G, the task Gram blocks (helper) =================
    if (is_main) {
        DWBC_PSTAMP_M(42);  // Y = J_C A^-1 stored
        wave_gemm<C, C, N, false>(  // J A^-1 J^T = Y J_C^T (rows and columns of an inactive contact are zero: Y and J_C are)
            [&](auto, auto, int i, int c) { return Yt[c * C + i]; },
            [&](auto, int c, int j) { return JCt[c * C + j]; },
            [&](int i, int j, real_t d) { L[S::c_s2 + i * C + j] = d; });
        if (cd == C) {
            if (!spd_inverse12_lds(L + S::c_s2, Lam, L + S::c_s1)) st_contact = 0;  // Lambda_c (wbd.cpp:115), in place
        } else if (cd > 0) {
            if (!spd_inverse_small(L + S::c_s2, C, cd, Lam, C, L + S::c_s1)) st_contact = 0;
        } else {
            for (int idx = th.tid; idx < C * C; idx += NT) Lam[idx] = real_t(0.0);
        }
        DWBC_SYNC();
    }
    if (is_main) DWBC_PSTAMP_M(43);  // Lambda_c
    PLA(real_t, jbk, C);  // main wave: column `lane` of Jbar^T = Lambda_c Y (wbd.cpp:116), kept for T1 in phase 3
    if (is_main) {
        LANES {
            const int col = lane < N ? lane : 0;
            real_t yck[C];
#pragma unroll
            for (int p = 0; p < C; p++) yck[p] = Yt[col * C + p];
            lds_rows_dot<C, C, C, 4, 0, S::c_Lam % 2 == 0>(Lam, yck, LV(jbk));
        }
        DWBC_SYNC();
    }
    real_t *Dm = L + S::c_D;   // D = J_t A^-1 J_C^T, one row per vector slot; row GV: Y G
    real_t *BJ = L + S::c_BJ;  // (J_t A^-1) J_t^T of every level
    if (is_help) {
        wave_gemm<NVS, C, N, false>(  // D = (J_t A^-1) J_C^T of every slot; row GV: Y G = (A^-1 G)^T J_C^T
            [&](auto, auto, int v, int c) { return AJt[v * N + c]; },
            [&](auto, int c, int p_) { return JCt[c * C + p_]; },
            [&](int v, int p_, real_t d) { Dm[v * C + p_] = d; });
        wave_gemm<GV, GV, N, false>(  // (J_t A^-1) J_t^T: the diagonal T x T blocks are the levels' own
            [&](auto, auto, int i, int c) { return AJt[i * N + c]; },
            [&](auto, int c, int j) { return L[S::Jtt + (j / T) * N * T + c * T + (j % T)]; },
            [&](int i, int j, real_t d) {
                const int lv = i / T, c_ = j - lv * T;
                if (c_ >= 0 && c_ < T && lv < su.n_levels) BJ[lv * T * T + (i - lv * T) * T + c_] = d;
            });
        // starting values of phase 4's accumulations, while J_t^T is still there: row c of [J_t^T of every slot | G], base rows to cv,
        // joint rows to tau (both transposed to slot-minor); the staged mass matrix they replace was last read in phase 1b
        for (int idx = th.tid; idx < N * NVS; idx += NT) {
            const int c = idx / NVS, v = idx - c * NVS, lv = v < GV ? v / T : 0, r = v - lv * T;
            const real_t jt = L[S::Jtt + lv * N * T + c * T + (v < GV ? r : 0)], gv = L[S::G + c];
            const real_t val = v == GV ? gv : ((lv < su.n_levels && r < su.t_dof[lv]) ? jt : real_t(0.0));
            L[(c < 6 ? S::p4_cv + c * VS : S::p4_ts + (c - 6) * VS) + v] = val;
        }
        DWBC_SYNC();
    }
    DWBC_PAIR_BARRIER_X();  // ---- B1b: the helper is done with J_t^T: Jbar^T takes its place
    if (is_main) {
        LANES {
#pragma unroll
            for (int p = 0; p < C; p++)
                if (lane < N) JbT[p * N + lane] = LV(jbk)[p];
        }
    }
    DWBC_PAIR_BARRIER(2);  // ---- B2: Jbar^T, Lambda_c, D, Y G, the task Gram blocks in LDS

    // ================= phases 3 + 4: the main wave applies W^+ without forming it; the helper finishes T1, NwJw and Lambda_task =================
    // W = (A^-1 N_c)[6:, 6:] is the map joint torque -> joint acceleration of the contact-constrained robot, so W^+ a -- all this cycle
    // ever asks of W^+, for a = the rows of T1r of each level (J_kt = W^+ T1r^T ..., wbd.cpp:207-213) and the gravity pre-vector
    // (wbd.cpp:190) -- is constrained INVERSE dynamics, and for THESE vectors it needs neither A nor A^-1: a is the joint part of
    // qdd = A^-1 N_c j (j = a row of J_t, or G), an acceleration that already satisfies the contacts (J_C A^-1 N_c = 0), and
    //     A qdd = N_c j = j - J_C^T Jbar j = j - J_C^T (Lambda_c d),     d = J_C A^-1 j  (a row of D = (J_t A^-1) J_C^T, resp. Y G)
    // so with cv = the base rows of that (j_b - J_Cb^T Lambda_c d), contact forces lam = -Hb^T cv that carry it (a particular solution of
    // J_Cb^T lam = -cv: Hb = (J_Cb^T J_Cb)^-1 J_Cb^T; without contacts the base takes it, Hb = A_bb^-1) and E = Lambda_c d - lam:
    //     tau_any = j_j - J_Cj^T E,        W^+ a = (I - P) tau_any        (P = VG Vb^T: the projector on null(W), internal wrenches)
    // -- five small products on the matrix cores (26 MFMA instructions for all 13 vectors) instead of the W + alpha P assembly (the
    // rank-12 update A^-1 N_c, the projector column), its 33-pivot sweep and the staging of what rode through it (33 k cycles of round
    // 4's first version of this kernel).  P_C = Lambda_c (Y G) (wbd.cpp:119) is the gravity slot's row of D Lambda_c.
    PLA(real_t, tv, VS);  // main wave, lane i < M: (W^+ a_v)[i] for every vector slot v: column i of J_kt of every level, torque_grav_[i]
    if (is_main) {
        real_t *DLn = L + S::p4_dl, *Cv = L + S::p4_cv, *Em = L + S::p4_e, *Wv = L + S::p4_w, *TS = L + S::p4_ts;
        const real_t zero = real_t(0.0);
        wave_gemm<NVS, C, C, false>(  // D Lambda_c, one row per slot
            [&](auto, auto, int v, int p) { return Dm[v * C + p]; },
            [&](auto, int p, int q_) { return Lam[p * C + q_]; },
            [&](int v, int q_, real_t d) {
                DLn[v * C + q_] = d;
                if (v == GV) L[S::PC + q_] = d;
            });
        DWBC_SYNC();
        DWBC_PSTAMP_M(45);  // D Lambda_c, P_C
        wave_gemm<6, VS, C, false>(  // cv = j_b - J_Cb^T (Lambda_c d): accumulated over the J_t^T / G rows the helper left there
            [&](auto, auto, int x, int p) { return -JCt[x * C + p]; },
            [&](auto, int p, int v) { const real_t v_ = DLn[(v < NVS ? v : 0) * C + p]; return v < NVS ? v_ : zero; },
            [&](int x, int v, real_t d) { Cv[x * VS + v] = d; },
            [&](int x, int v) { return Cv[x * VS + v]; });
        DWBC_SYNC();
        DWBC_PSTAMP_M(46);  // W^+: cv
        if (nc > 0) {
            wave_gemm<C, VS, 6, false>(  // E = Lambda_c d + Hb^T cv
                [&](auto, auto, int p, int x) { return Hb[x * C + p]; },
                [&](auto, int x, int v) { return Cv[x * VS + v]; },
                [&](int p, int v, real_t d) { Em[p * VS + v] = d; },
                [&](int p, int v) { const real_t v_ = DLn[(v < NVS ? v : 0) * C + p]; return v < NVS ? v_ : zero; });
        } else {
            for (int idx = th.tid; idx < C * VS; idx += NT) Em[idx] = idx < 6 * VS ? Cv[idx] : zero;  // (the base takes it: see Jx below)
        }
        DWBC_SYNC();
        DWBC_PSTAMP_M(47);  // W^+: E
        {
            // tau_any = j_j - J_Cj^T E in place; without contacts: j_j - (A_jb A_bb^-1) cv, the helper left A_jb A_bb^-1 (M x 6) in VG's place
            const real_t *Jx = nc > 0 ? JCt + 6 * C : VG;
            const int js = nc > 0 ? C : 6;
            wave_gemm<M, VS, C, false>(
                [&](auto, auto, int i, int p) { const real_t v_ = Jx[i * js + (p < js ? p : 0)]; return p < js ? -v_ : zero; },
                [&](auto, int p, int v) { return Em[p * VS + v]; },
                [&](int i, int v, real_t d) { TS[i * VS + v] = d; },
                [&](int i, int v) { return TS[i * VS + v]; });
        }
        DWBC_SYNC();
        DWBC_PSTAMP_M(48);  // W^+: tau_any
        if (k > 0) {  // W^+ a = (I - P) tau_any,  P tau = VG (Vb^T tau)
            wave_gemm<6, VS, M, false>(
                [&](auto, auto, int a, int i) { return Vb[i * 6 + a]; },
                [&](auto, int i, int v) { return TS[i * VS + v]; },
                [&](int a, int v, real_t d) { Wv[a * VS + v] = d; });
            DWBC_SYNC();
        }
        LANES {
            const int i = lane < M ? lane : 0;
#pragma unroll
            for (int v = 0; v < VS; v++) LV(tv)[v] = TS[i * VS + v];
            if (k > 0) {
                real_t ga[6];
#pragma unroll
                for (int a = 0; a < 6; a++) ga[a] = -VG[i * 6 + a];
                lds_rows_axpy<6, VS, VS, 3, S::p4_w % 2 == 0>(Wv, ga, LV(tv));
            }
        }
        DWBC_PSTAMP_M(49);  // W^+: projected
        LANES {
            if (lane < M) L[S::tg + lane] = LV(tv)[GV];  // torque_grav_ = W^+ (A^-1 N_c G)[6:]   (wbd.cpp:190)
        }
        // the joint columns of T1 = J_t A^-1 N_c = J_t A^-1 - D Jbar^T of every level (T1x: the null-space chain and the slow route of
        // phase 5 read them), the riding block as the starting value -- here, because the helper's chain is the longer one
        wave_gemm<M, GV, C, false>(
            [&](auto, auto, int j, int p) { return -JbT[p * N + 6 + j]; },
            [&](auto, int p, int v) { return Dm[v * C + p]; },
            [&](int j, int v, real_t d) { L[S::T1x + (v / T) * T * S::MS + (v % T) * S::MS + j] = d; },
            [&](int j, int v) { return AJt[v * N + 6 + j]; });
        DWBC_SYNC();
    }
    if (is_help) {
        // Lambda_task of every level from the Gram blocks: J_t A^-1 N_c J_t^T = (J_t A^-1) J_t^T - D Lambda_c D^T (wbd.cpp:210), and the
        // condition verdict (dwbc_cycle2.h, task-Jacobian stage)
        int fm = 0;
        real_t *DLx = L + S::q;  // D Lambda_c (GV x C) in the header scratch (q, G, ...: dead since phase 1b)
        static_assert(GV * C <= S::hend - S::q && NLV <= 2, "D Lambda_c borrows the header scratch; two Lambda_task blocks in the helper's scratch");
        constexpr int K6 = 6;
        // the helper's small scratch: Lambda_task^-1 of level 0 | G^-1 (since phase 1b) | Lambda_task^-1 of level 1 | JV | B, X | S
        real_t *Li0 = L + S::hs, *Gi = L + S::hs + 36, *Li1 = L + S::hs + 72, *JV = L + S::hs + 108, *Bm = L + S::hs + 144, *Sm6 = L + S::hs + 180;
        static_assert(S::hs_size >= 216, "six 6 x 6 blocks");
        DWBC_SYNC();
        wave_gemm<GV, C, C, false>(
            [&](auto, auto, int i, int p) { return Dm[i * C + p]; },
            [&](auto, int p, int q_) { return Lam[p * C + q_]; },
            [&](int i, int q_, real_t d) { DLx[i * C + q_] = d; });
        if (k > 0) {  // JV = Jbar[0:k, 6:] Vb  (NwJw = VG X^T with X = (JV G^-1 JV^T)^-1 JV: SPD inverses only, dwbc_cycle2.h)
            wave_gemm<K6, K6, M, false>(
                [&](auto, auto, int i, int c) { return JbT[i * N + 6 + c]; },
                [&](auto, int c, int j) { return Vb[c * K6 + j]; },
                [&](int i, int j, real_t d) { JV[i * K6 + j] = d; });
        }
        DWBC_SYNC();
        wave_gemm<GV, GV, C, false>(  // the diagonal blocks of (J_t A^-1) J_t^T - (D Lambda_c) D^T, row stride t_lv
            [&](auto, auto, int i, int p) { return -DLx[i * C + p]; },
            [&](auto, int p, int j) { return Dm[j * C + p]; },
            [&](int i, int j, real_t d) {
                const int lv = i / T, r = i - lv * T, c_ = j - lv * T;
                if (c_ >= 0 && c_ < T && lv < su.n_levels) {
                    const int t = su.t_dof[lv];
                    if (r < t && c_ < t) (lv == 0 ? Li0 : Li1)[r * t + c_] = d;
                }
            },
            [&](int i, int j) {
                const int lv = i / T, c_ = j - lv * T;
                const bool in = c_ >= 0 && c_ < T;
                const real_t v_ = BJ[lv * T * T + (i - lv * T) * T + (in ? c_ : 0)];
                return in ? v_ : real_t(0.0);
            });
        if (k > 0) {
            mm_nn<NT>(th, Bm, K6, JV, K6, Gi, K6, K6, K6, K6);                   // B = JV G^-1
            DWBC_SYNC();
            mm_nt<NT>(th, Sm6, K6, Bm, K6, JV, K6, K6, K6, K6);                  // S = B JV^T  (SPD)
        }
        DWBC_SYNC();
        // the diagonals of Lambda_task^-1 before the inverses overwrite nothing of them (Lt is a block of its own), then all three inverses
        const int t0 = su.n_levels > 0 ? su.t_dof[0] : 0, t1 = su.n_levels > 1 ? su.t_dof[1] : 0;
        int ok3[3];
        spd_inverse_chol6_x3(Li0, t0, L + S::c_Lt, Li1, t1, L + S::c_Lt + T * T, Sm6, k > 0 ? K6 : 0, Sm6, ok3);
        for (int lv = 0; lv < su.n_levels; lv++) {
            const int t = su.t_dof[lv];
            const real_t *Lt = L + S::c_Lt + lv * T * T, *Li = lv == 0 ? Li0 : Li1;
            real_t da = real_t(0.0), dl = real_t(0.0);
            for (int i = 0; i < t; i++) {
                const real_t a_ = Li[i * t + i], l_ = Lt[i * t + i];
                da = a_ > da ? a_ : da;
                dl = l_ > dl ? l_ : dl;
            }
            if (ok3[lv] && nc > 0 && da * dl < kCodCondFast) fm |= 1 << lv;
        }
        if (th.tid == 0) flg[1] = (real_t)fm;
        if (k > 0) {
            if (!ok3[2]) { if (th.tid == 0) flg[0] = real_t(0.0); }
            mm_nn<NT>(th, Bm, K6, Sm6, K6, JV, K6, K6, K6, K6);                  // X = S^-1 JV
            DWBC_SYNC();
            wave_gemm<M, K6, K6, false>(                                         // NwJw = VG X^T
                [&](auto, auto, int i, int a) { return VG[i * K6 + a]; },
                [&](auto, int a, int j) { return Bm[j * K6 + a]; },
                [&](int i, int j, real_t d) { L[S::NwJw + i * K6 + j] = d; });
        }
        DWBC_SYNC();
    }
    DWBC_PAIR_BARRIER(4);  // ---- B4: J_kt-side vectors in the main wave's registers; Lambda_task, the fast-route mask, NwJw in LDS
    if (!is_main) return;

    // ================= phase 5 (main): stage 3a, wrench maps, QP cascade in registers, outputs =================
    static_assert(NLV <= 2, "the null-space chain below takes X_0 = U_0 from the level-0 rows");
    if (flg[0] == real_t(0.0)) st_contact = 0;
    const int fastmask = (int)flg[1];
    int rankbad = 0;
    // Row `lane` of the cascade's operators, kept in registers from here to the outputs (lane r < M: torque row r; lane M + rr: cone
    // row rr): gd[l][:] = the task block of level l (U_l[r, :] resp. -cone(WM_{U_l})[rr, :]), gcn[:] = the contact-null block
    // (NwJw[r, :] resp. -cone(WM_N)[rr, :]).  Every right-hand side of the three QPs and every output torque is a lane-local
    // combination of these with uniform vectors (f*, the QP answers): no base / wrench vectors through LDS between the QPs.
    PLA(real_t, gd0, 6);  // (one array per level: a run-time level index into one array would send it to scratch)
    PLA(real_t, gd1, 6);
    PLA(real_t, gcn, 6);
    LANES {
#pragma unroll
        for (int j = 0; j < 6; j++) { LV(gd0)[j] = real_t(0.0); LV(gd1)[j] = real_t(0.0); }
    }
    for (int lv = 0; lv < su.n_levels; lv++) {
        auto level_body = [&](auto ttl) {
        constexpr int TTL = decltype(ttl)::value;
        const int t = TTL;
        const real_t *Lt = L + S::c_Lt + lv * T * T;
        const real_t *T1rl = L + S::T1x + lv * T * S::MS;
        real_t *Q = L + S::c_QW, *QW = L + S::c_QWp, *Pi = L + S::c_Pi;
        int cond = 1;
        DWBC_SYNC();
        const bool fast = (fastmask >> lv) & 1;
        PLA(real_t, xr, TTL);  // row `lane` of X = J_kt Lambda of this level
        if (fast) {
            // J_kt = W^+ T1r^T is in this lane's registers (vector slots lv * T + r): X = J_kt Lambda, row `lane` per lane
            LANES {
                real_t tw[TTL];
#pragma unroll
                for (int r = 0; r < TTL; r++) {
                    real_t v_ = LV(tv)[r];
                    if constexpr (NLV > 1) v_ = lv == 1 ? LV(tv)[T + r] : v_;
                    tw[r] = v_;
                }
#pragma unroll
                for (int r3 = 0; r3 < TTL; r3++) {
                    real_t acc = real_t(0.0);
#pragma unroll
                    for (int r2 = 0; r2 < TTL; r2++) acc += tw[r2] * Lt[r2 * TTL + r3];
                    LV(xr)[r3] = acc;
                }
            }
        } else {
            // the reference's own sequence (wbd.cpp:207-213) with the rank-revealing pseudo-inverse where the block calls for it
            for (int idx = th.tid; idx < t * M; idx += NT) {
                const int i = idx / M, j = idx - i * M;
                real_t acc = real_t(0.0);
                for (int p = 0; p < t; p++) acc += Lt[i * t + p] * T1rl[p * S::MS + j];
                Q[idx] = acc;
            }
            DWBC_SYNC();
            // Q W^+ = Lambda_t (W^+ T1r^T)^T: W^+ is linear and its values on the rows of T1r are in this lane's registers
            LANES {
                real_t tw[TTL];
#pragma unroll
                for (int r = 0; r < TTL; r++) {
                    real_t v_ = LV(tv)[r];
                    if constexpr (NLV > 1) v_ = lv == 1 ? LV(tv)[T + r] : v_;
                    tw[r] = v_;
                }
                if (lane < M) {
#pragma unroll
                    for (int r = 0; r < TTL; r++) {
                        real_t acc = real_t(0.0);
#pragma unroll
                        for (int r2 = 0; r2 < TTL; r2++) acc += Lt[r * TTL + r2] * tw[r2];
                        QW[r * M + lane] = acc;
                    }
                }
            }
            DWBC_SYNC();
            mm_nt<NT>(th, L + S::c_s2b, t, QW, M, Q, M, t, M, t);
            real_t pr_ = real_t(1.0);
            cond = spd_inverse_small(L + S::c_s2b, t, t, Pi, t, L + S::c_s1, &pr_);
            DWBC_SYNC();
            if (!cond || pr_ < kCodCheck) {
                const int rk = pinv_cod_small<NT>(th, L + S::c_s2b, t, kCodThreshold, Pi, L + S::cod_Q, L + S::cod_G, L + S::cod_T, L + S::cod_v);
                if (rk < t) cond = 1;
            }
            LANES {
                real_t jk[TTL], qw[TTL];
#pragma unroll
                for (int r = 0; r < TTL; r++) qw[r] = QW[r * M + (lane < M ? lane : 0)];
#pragma unroll
                for (int r2 = 0; r2 < TTL; r2++) {
                    real_t acc = real_t(0.0);
#pragma unroll
                    for (int r = 0; r < TTL; r++) acc += qw[r] * Pi[r * TTL + r2];
                    jk[r2] = acc;
                }
#pragma unroll
                for (int r3 = 0; r3 < TTL; r3++) {
                    real_t acc = real_t(0.0);
#pragma unroll
                    for (int r2 = 0; r2 < TTL; r2++) acc += jk[r2] * Lt[r2 * TTL + r3];
                    LV(xr)[r3] = acc;
                }
            }
        }
        DWBC_PSTAMP(16 + 2 * lv);  // rows of J_kt / X of level lv in registers
        LANES {
#pragma unroll
            for (int r = 0; r < TTL; r++) {
                const real_t v_ = lane < M ? LV(xr)[r] : real_t(0.0);
                LV(gd0)[r] = lv == 0 ? v_ : LV(gd0)[r];
                if constexpr (NLV > 1) LV(gd1)[r] = lv == 1 ? v_ : LV(gd1)[r];
            }
        }
        if (!cond) rankbad |= (1 << lv);
            };
        if (su.t_dof[lv] <= 3) level_body(std::integral_constant<int, 3>{}); else level_body(std::integral_constant<int, T>{});
    }
    // second pass, once nothing reads the staged mass matrix any more (U shares its place): U_0 = X_0; U_1 = (I - X_0 Y_0) X_1
    for (int lv = 0; lv < su.n_levels; lv++) {
        auto chain_body = [&](auto ttl) {
        constexpr int TTL = decltype(ttl)::value;
        real_t *Ul = L + S::U + lv * M * T;
        PLA(real_t, xr, TTL);
        LANES {
#pragma unroll
            for (int r = 0; r < TTL; r++) {
                real_t v_ = LV(gd0)[r];
                if constexpr (NLV > 1) v_ = lv == 1 ? LV(gd1)[r] : v_;
                LV(xr)[r] = v_;
            }
        }
        if (lv > 0) {
            // U_lv = (I - X_0 Y_0) X_lv, Y_0 = T1x[0] (wbd.cpp:228-261): Z = Y_0 X_lv is the one cross-lane reduction of the chain
            const int tp = su.t_dof[0];
            const real_t *Yp = L + S::T1x;
            LANES {
                if (lane < M) {
#pragma unroll
                    for (int r = 0; r < TTL; r++) Ul[lane * T + r] = LV(xr)[r];
                }
            }
            DWBC_SYNC();
            wave_gemm<T, TTL, M, false>(  // (rows of Y_0 beyond tp are zero: empty slots)
                [&](auto, auto, int i, int c) { return Yp[i * S::MS + c]; },
                [&](auto, int c, int j) { return Ul[c * T + j]; },
                [&](int i, int j, real_t d) { L[S::c_Z + i * TTL + j] = d; });
            DWBC_SYNC();
            LANES {
#pragma unroll
                for (int j = 0; j < TTL; j++) {
                    real_t acc = LV(xr)[j];
#pragma unroll
                    for (int p = 0; p < T; p++) acc -= LV(gd0)[p] * L[S::c_Z + p * TTL + j];  // (rows of Z beyond tp are zero)
                    LV(xr)[j] = acc;
                }
            }
            DWBC_SYNC();
        }
        LANES {
#pragma unroll
            for (int r = 0; r < TTL; r++) {
                const real_t v_ = lane < M ? LV(xr)[r] : real_t(0.0);
                LV(gd0)[r] = lv == 0 ? v_ : LV(gd0)[r];
                if constexpr (NLV > 1) LV(gd1)[r] = lv == 1 ? v_ : LV(gd1)[r];
                if (lane < M) Ul[lane * T + r] = LV(xr)[r];  // (the wrench maps read U_l from LDS)
            }
        }
        DWBC_PSTAMP(17 + 2 * lv);  // null-space chain of level lv done
            };
        if (su.t_dof[lv] <= 3) chain_body(std::integral_constant<int, 3>{}); else chain_body(std::integral_constant<int, T>{});
    }
    DWBC_SYNC();
    DWBC_STAMP(5);  // stage 3a done

    // ---- the QP cascade (dwbc.cpp:818-873, 941-1127) and the contact redistribution QP (dwbc.cpp:1372-1568)
    const int nlim = su.has_tau_lim ? 2 * M : 0;
    const int ncone = 10 * nc;
    int st_task = 1, fail_level = -1, st_redis = 1;
    const real_t *fs_in = L + S::fs;
    real_t *WM = L + S::wm;
    constexpr int WLD = S::WLD;
    const int colN = 1 + su.fstar_total;
    QpLaneConst qc;
    PL(real_t, sfin);  // slack of the lane's QP row at the point the last QP returned (qp_solve_wave)
    PL(real_t, tgl);   // torque lanes: torque_grav_[lane]; cone lanes: the cone row applied to the gravity wrench J̄ tau_grav - P_C
    PL(real_t, tta);   // what the committed levels add to it: sum_l gd_l . (f*_l + f*_qp_l)  (torque lanes: torque_task_[lane])
    PL(real_t, Tl);    // torque lanes: the torque limit; cone lanes: 0
    bool skip_redis = false;
    qp_lane_consts<N>(su, act_c[0], act_c[1], qc);
    wrench_maps<N, NT, S>(th, su, L, JbT, cd, k, WM);
    DWBC_SYNC();
    LANES {
        const bool tqr = lane < M;
        const bool cn = lane >= M && lane - M < ncone;
        const int row2 = cn ? LV(qc.row2) : 0, rowo = cn ? LV(qc.rowo) : 0;
        const real_t ca = -LV(qc.c2), cb = -LV(qc.sg);
        // unconditional reads from clamped addresses, masked afterwards (one straight-line path for torque and cone lanes)
        const real_t *pa = tqr ? L + S::NwJw + lane * 6 : WM + row2 * WLD + colN;
        const real_t *pb = WM + rowo * WLD + colN;
        real_t va[6], vb[6];
#pragma unroll
        for (int j = 0; j < 6; j++) { va[j] = pa[j]; vb[j] = pb[j]; }
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const real_t v = tqr ? va[j] : ca * va[j] + cb * vb[j];
            LV(gcn)[j] = ((tqr || cn) && j < k) ? v : real_t(0.0);
        }
#pragma unroll
        for (int l = 0; l < NLV; l++) {
            const int off = l < su.n_levels ? 1 + su.fstar_off[l] : 1, tl = l < su.n_levels ? su.t_dof[l] : 0;
            real_t ua[6], ub[6];
#pragma unroll
            for (int j = 0; j < 6; j++) { ua[j] = WM[row2 * WLD + off + j]; ub[j] = WM[rowo * WLD + off + j]; }
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const real_t v = (cn && j < tl) ? ca * ua[j] + cb * ub[j] : real_t(0.0);
                if (l == 0) LV(gd0)[j] = tqr ? LV(gd0)[j] : v;
                else LV(gd1)[j] = tqr ? LV(gd1)[j] : v;
            }
        }
        // gravity part of the right-hand sides: torque_grav_[lane], resp. the cone row on J̄[:, 6:] tau_grav - P_C (contact frame)
        const int a2 = row2 / 6, x2 = row2 % 3, h2 = (row2 % 6) / 3, ao = rowo / 6, xo = rowo % 3, ho = (rowo % 6) / 3;
        const real_t *R2 = L + S::Rc + a2 * 9, *Ro = L + S::Rc + ao * 9;
        const real_t *p2 = L + S::PC + 6 * a2 + 3 * h2, *po = L + S::PC + 6 * ao + 3 * ho;
        const real_t w2 = WM[row2 * WLD] - (R2[x2] * p2[0] + R2[3 + x2] * p2[1] + R2[6 + x2] * p2[2]);
        const real_t wo = WM[rowo * WLD] - (Ro[xo] * po[0] + Ro[3 + xo] * po[1] + Ro[6 + xo] * po[2]);
        const real_t tgv = L[S::tg + (tqr ? lane : 0)];
        LV(tgl) = tqr ? tgv : (cn ? ca * w2 + cb * wo : real_t(0.0));
        LV(tta) = real_t(0.0);
        LV(Tl) = tqr ? LV(qc.taul) : real_t(0.0);
        LV(sfin) = real_t(0.0);
    }
    // The rows leave the registers again: 18 doubles per lane [gd_0 | gd_1 | gcn] in an LDS block of their own (over J_C, NwJw and U,
    // all dead now that the wrench maps and the rows exist), read back in 16-byte pieces where a QP or an output needs them --
    // kept in registers across the three solves they pushed the solver over the 256-register cap (56 spilled registers, reloaded
    // behind serial waits inside the cascade).
    constexpr int RSW = 18;
    static_assert(64 * RSW <= S::UAend - S::c_JC && S::c_JC % 2 == 0, "row block over the dead J_C, NwJw, U region");
    real_t *rowblk = L + S::c_JC;
    DWBC_SYNC();
    LANES {
        real_t *rw = rowblk + lane * RSW;
#pragma unroll
        for (int j = 0; j < 6; j++) { rw[j] = LV(gd0)[j]; rw[6 + j] = LV(gd1)[j]; rw[12 + j] = LV(gcn)[j]; }
    }
    DWBC_SYNC();
    DWBC_STAMP(6);  // wrench maps done, rows stored
    real_t *fx = L + S::t_base;  // f* + f*_qp of the committed levels (NLV x 6, LDS: only the wrench output reads it again)
    real_t ctot[6];      // contact_qp_ of the last committed level (+ the redistribution's answer): what torque_contact_ stands for
    for (int j = th.tid; j < NLV * 6; j += NT) fx[j] = real_t(0.0);
    DWBC_SYNC();
#pragma unroll
    for (int j = 0; j < 6; j++) ctot[j] = real_t(0.0);
    for (int qi = 0; qi <= su.n_levels; qi++) {
        const bool is_task = qi < su.n_levels;
        if (is_task && !st_task) continue;
        if (!is_task && k == 0) break;
        if (!is_task && skip_redis) {  // every row of the redistribution QP holds at c = 0 (seen from the last task QP's final slacks): no step
            if (diag && th.tid == 0) {
                diag[DG_QP_ITER + kMaxLevels] = 0;
                diag[DG_QP_NACT + kMaxLevels] = 0;
            }
            break;
        }
        const int t = is_task ? su.t_dof[qi] : 0;
        const bool t6 = t > 3;
        if (is_task && (rankbad & (1 << qi))) { st_task = 0; fail_level = qi; continue; }
        real_t f6[6];  // f* of the level (task QP) resp. contact_qp_ handed over by the last level (redistribution); beyond t / k: 0
#pragma unroll
        for (int j = 0; j < 6; j++) f6[j] = is_task ? (j < t ? fs_in[su.fstar_off[qi] + j] : real_t(0.0)) : ctot[j];
        QpRows R;
        const int nv = is_task ? t + k : k;
        LANES {
            const bool tq = lane < M && nlim != 0, cn = lane >= M && lane - M < ncone;
            const bool act = tq || cn;
            const real_t *rw = rowblk + lane * RSW;
            real_t gl[6], gc6[6];  // the level's task block (its offset in the row is uniform) and the contact-null block
#pragma unroll
            for (int j = 0; j < 6; j++) { gl[j] = rw[(qi == 1 ? 6 : 0) + j]; gc6[j] = rw[12 + j]; }
            real_t dot = real_t(0.0);
#pragma unroll
            for (int j = 0; j < 6; j++) dot += (is_task ? gl[j] : gc6[j]) * f6[j];
            const real_t v = (LV(tgl) + LV(tta)) + dot;   // torque lanes: tau_grav + tau_task so far + U_l f* (+ NwJw contact_qp_)
            // variables [delta (t = 6 | 3); c_hat (k)] of a task QP, [c (k)] of the redistribution -- static register indices, uniform selects
            const real_t sc = kQpScaleGI;
#pragma unroll
            for (int j = 0; j < kQpN; j++) {
                real_t g_;
                if (j < 3) g_ = is_task ? gl[j] : gc6[j];
                else if (j < 6) g_ = is_task ? (t6 ? gl[j] : sc * gc6[j - 3]) : gc6[j];
                else if (j < 9) g_ = is_task ? (t6 ? sc * gc6[j - 6] : sc * gc6[j - 3]) : real_t(0.0);
                else g_ = (is_task && t6) ? sc * gc6[j - 6] : real_t(0.0);
                LV(R.g)[j] = (act && j < nv) ? g_ : real_t(0.0);
            }
            LV(R.hi) = act ? LV(Tl) - v : DWBC_QP_INF;   // (reference src/dwbc.cpp:1001-1016 torque rows, :1041-1053 cone rows)
            LV(R.lo) = tq ? LV(Tl) + v : DWBC_QP_INF;
            LV(R.id_hi) = tq ? lane : (cn ? nlim + (lane - M) : -1);
            LV(R.id_lo) = tq ? M + lane : -1;
        }
        if (qi == 0) DWBC_PSTAMP(24);  // level 0: rows ready
        if (!is_task) DWBC_PSTAMP(26);  // redistribution: rows ready
        QpResult qres;
        {
            const int tv = is_task ? t : k, mi = is_task ? su.qp_max_iter_task : su.qp_max_iter_contact;
            const real_t tol = is_task ? kQpTol : kQpFeasTol;
            if (nv <= 6) qp_solve_wave<0, 6>(R, nv, tv, mi, qres, L + S::qp_V, nullptr, tol, sfin);
            else if (nv <= 9) qp_solve_wave<0, 9>(R, nv, tv, mi, qres, L + S::qp_V, nullptr, tol, sfin);
            else qp_solve_wave<0, 12>(R, nv, tv, mi, qres, L + S::qp_V, nullptr, tol, sfin);
        }
        const int slot = is_task ? qi : kMaxLevels;
        if (diag && th.tid == 0) {
            diag[DG_QP_ITER + slot] = qres.iters;
            diag[DG_QP_NACT + slot] = qres.nact;
        }
        if (is_task) DWBC_STAMP(7 + qi);  // QP of level qi solved
#if defined(DWBC_STAGE_TIMERS) && !defined(DWBC_HOST_EMU)
        if (diag && th.tid == 0 && qi == 0)
            for (int i_ = 0; i_ < 9; i_++) diag[DG_FTIME + 32 + i_] = (int)qres.tm[i_];
#endif
        if (is_task) {
            if (!qres.status) { st_task = 0; fail_level = qi; continue; }
            real_t fxl[6];
#pragma unroll
            for (int j = 0; j < 6; j++) {
                fxl[j] = j < t ? f6[j] + qres.x[j] : real_t(0.0);                     // f* + f*_qp (dwbc.cpp:839-849)
                ctot[j] = j < k ? (t6 ? qres.x[6 + j] : qres.x[3 + j]) : real_t(0.0);  // contact_qp_ (dwbc.cpp:851)
            }
            LANES {
                real_t v_ = real_t(0.0);
#pragma unroll
                for (int j = 0; j < 6; j++) v_ = (lane == j) ? fxl[j] : v_;
                if (lane < 6) fx[qi * 6 + lane] = v_;
            }
            LANES {
                const real_t *rw = rowblk + lane * RSW + (qi == 1 ? 6 : 0);
                real_t acc = real_t(0.0);
#pragma unroll
                for (int j = 0; j < 6; j++) acc += rw[j] * fxl[j];
                LV(tta) += acc;  // torque lanes: torque_task_ += Null_{i-1} J_kt Lambda (f* + f*_qp); cone lanes: the wrench it adds, on the cone row
            }
            if (qi == su.n_levels - 1 && k > 0) {
                // The rows of the redistribution QP (dwbc.cpp:1458-1517) are the rows of this level's QP seen at c = 0 with the total
                // torque on the right-hand side, so its first look at them -- slack / |row| against the tolerance of canon rule 5 --
                // can be taken from the slacks this QP ended with.  Nothing violated: the redistribution would return c = 0; it is skipped.
                PL(real_t, grn);
                LANES {
                    const bool tq = lane < M && nlim != 0, cn = lane >= M && lane - M < ncone;
                    const real_t *rw = rowblk + lane * RSW + 12;
                    real_t s2 = real_t(0.0);
#pragma unroll
                    for (int j = 0; j < 6; j++) { const real_t g_ = rw[j]; s2 += g_ * g_; }
                    const real_t gn = s2 < kQpZeroRow * kQpZeroRow ? real_t(1.0) : sqrt(s2);
                    LV(grn) = (tq || cn) ? LV(sfin) / gn : DWBC_QP_INF;
                }
                int wl_;
                WAVE_ARGMIN_F32(grn, wl_);
                skip_redis = !(BCAST(grn, wl_) < -kQpFeasTol);
            }
        } else if (qres.status) {
#pragma unroll
            for (int j = 0; j < 6; j++) ctot[j] += j < k ? qres.x[j] : real_t(0.0);  // contact_qp_ + redistribution (dwbc.cpp:1549)
        } else {
            st_redis = 0;
#pragma unroll
            for (int j = 0; j < 6; j++) ctot[j] = real_t(0.0);
        }
        if (qi == 0) DWBC_PSTAMP(25);  // level 0: committed
        if (!is_task) DWBC_PSTAMP(27);  // redistribution QP done and committed
    }
    DWBC_SYNC();
    // ---- outputs: the three torques from the rows the lanes hold (straight to HBM), the wrench from the wrench maps
    io_t *tau = io.tau + (size_t)inst * 3 * M;
    LANES {
        if (lane < M) {
            const real_t *rw = rowblk + lane * RSW + 12;
            real_t tcv = real_t(0.0);
#pragma unroll
            for (int j = 0; j < 6; j++) tcv += rw[j] * ctot[j];   // torque_contact_ = NwJw (contact_qp_ [+ redistribution])
            tau[lane] = too_many ? real_t(0.0) : LV(tgl);
            tau[M + lane] = too_many ? real_t(0.0) : LV(tta);
            tau[2 * M + lane] = (too_many || k == 0) ? real_t(0.0) : tcv;
        }
    }
    DWBC_PSTAMP(28);  // torques stored
    io_t *wr = io.wrench + (size_t)inst * 12;
    // getContactForce(tau_total) = Jbar[:, 6:] tau - P_C (wbd.cpp:268-271).  In the contact frames that is a column combination of the
    // wrench maps: gravity column - P_C + sum_l WM_{U_l} (f* + f*_qp)_l + WM_N (contact_qp_ + redistribution); rotated back with
    // blockdiag(R_a, R_a).
    // (lane r forms row r of the local wrench -- its reads in one straight-line batch --, the rows regroup through LDS for the rotation:
    // twelve lanes that each walked three rows one dependent read after the other were 6 k cycles of this phase)
    real_t *wloc = L + S::t_fv;
    LANES {
        const int r = lane < C ? lane : 0;
        const int a = r / 6, h = (r % 6) / 3, x_ = r % 3;
        const real_t *R = L + S::Rc + a * 9;
        const real_t *pc3 = L + S::PC + 6 * a + 3 * h;
        const real_t *row = WM + r * WLD;
        real_t wv[1 + NLV * 6 + 6];
#pragma unroll
        for (int j = 0; j < 1 + NLV * 6 + 6; j++) wv[j] = row[j < 1 + NLV * 6 ? j : colN + (j - 1 - NLV * 6)];
        real_t v = wv[0] - (R[x_] * pc3[0] + R[3 + x_] * pc3[1] + R[6 + x_] * pc3[2]);
#pragma unroll
        for (int l = 0; l < NLV; l++) {
            const int off = l < su.n_levels ? su.fstar_off[l] : 0, tl = l < su.n_levels ? su.t_dof[l] : 0;
#pragma unroll
            for (int j = 0; j < 6; j++) {
                // (column 1 + off + j of the row: the levels' blocks follow each other, so with two levels the offsets are 0 and t_0)
                real_t wj = wv[1 + j];
                if (l == 1) wj = (off == 3) ? wv[1 + 3 + j] : wv[1 + 6 + j];
                v += (j < tl) ? wj * fx[l * 6 + j] : real_t(0.0);
            }
        }
#pragma unroll
        for (int j = 0; j < 6; j++) v += (j < k) ? wv[1 + NLV * 6 + j] * ctot[j] : real_t(0.0);
        if (lane < C) wloc[lane] = (lane < cd && !too_many) ? v : real_t(0.0);
    }
    DWBC_SYNC();
    LANES {
        const int i = lane < C ? lane : 0;
        const int a = i / 6, y = i % 3, g3 = 3 * (i / 3);
        const real_t *R = L + S::Rc + a * 9;
        const real_t acc = R[y * 3] * wloc[g3] + R[y * 3 + 1] * wloc[g3 + 1] + R[y * 3 + 2] * wloc[g3 + 2];
        if (lane < 12) wr[lane] = (lane < cd && !too_many) ? acc : real_t(0.0);
    }
    DWBC_STAMP(15);
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

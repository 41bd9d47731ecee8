// dwbc_cycle2p.h -- the fused cycle with TWO wavefronts per robot instance (workgroup of 128 threads), for batches of at most one
// instance per SIMD (B <= 4 x CUs: BASELINE configs[1], 1024 instances on 1024 SIMDs).
//
// Why: with one wave per instance and one instance per SIMD nothing overlaps -- a lone wave issues one instruction per ~5 cycles and
// waits out every LDS round trip (SQ_WAIT_ANY 41 %); yet two waves sharing a SIMD each run at nearly their solo speed (the compact
// kernel at two waves per SIMD does 19.1 M cycles/s against 10.6 M at one: DESIGN.md section 3).  So the cycle's SIDE CHAINS -- work
// that does not need the column-per-lane registers of the big matrices -- move to a second wave of the same workgroup and run
// beside the main chain; the two meet at five workgroup barriers:
//
//   main wave (0)                                              | helper wave (1)
//   kinematics up to the world transforms                      | f* -> LDS
//   ---------------------------------------------------------- B0 (link frames ready) ----------------------------------------------
//   world inertias, composite inertias, CRBA, A -> registers,  | contact frames, J_C, internal-wrench basis Vb, its Gram matrix
//   tree-sparse A^-1 sweep                                     | and G^-1, VG = Vb G^-1; the task Jacobians of every level
//   ---------------------------------------------------------- B1 (A^-1 in registers, J_C in LDS) ----------------------------------
//   Y = J_C A^-1, Lambda_c (MFMA tile + 12 x 12 inverse), Jbar^T | (waits)
//   ---------------------------------------------------------- B2 (Jbar^T in LDS) --------------------------------------------------
//   A^-1 N_c update, gravity pre-vector, P_C,                  | NwJw = VG X^T chain (J̄_1 Vb, two 6 x 6 SPD inverses, four products)
//   T1 = J_t A^-1 N_c of every level                           |
//   ---------------------------------------------------------- B3 (T1, NwJw in LDS) ------------------------------------------------
//   W + alpha P, the 33-pivot LDS-fed sweep, W^+ correction,   | J_t A^-1 N_c J_t^T, Lambda_task and the condition verdict of every level
//   gravity torque                                             |
//   ---------------------------------------------------------- B4 -------------------------------------------------------------------
//   J_kt / X / null-space chain, wrench maps (MFMA), QP cascade, | (done)
//   outputs                                                    |
//
// The arithmetic of every block is that of dwbc_cycle2.h (same helpers, same order of operations inside a block), so the parity
// tests of the one-wave kernel apply unchanged; only who executes a block and where it sits in LDS differ.  Lean build only
// (no dump record, no optional paths: the launcher falls back to the one-wave kernels for those), hqp = true.
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
    // ---- link frames (stage 0 .. B1)
    static constexpr int Rw = hend;
    static constexpr int pw = Rw + NB * 9;
    static constexpr int aw = pw + NB * 3;
    static constexpr int Rw0 = Rw;
    static constexpr int fend = aw + NB * 3;
    // ---- long-lived
    static constexpr int JbT = fend;                           // C x N; J_C (N x C) until Jbar^T is written over it
    static constexpr int c_JC = JbT;
    static constexpr int NwJw = JbT + C * N;
    static constexpr int NL2 = NLV < 2 ? 2 : NLV;              // (the U / T1x regions also host the staged mass matrix)
    static constexpr int U = NwJw + M * K;                     // levels x (M x T)
    static constexpr int MS = ev(M);                           // row stride of T1x: even, so that every row is 16-byte aligned (lds_rows_dot)
    static constexpr int T1x = U + NL2 * M * T;                // levels x (T x MS): joint columns of T1 of every level
    static constexpr int c_T1 = T1x + NL2 * T * MS;            // levels x (T x 6): base columns of T1
    static constexpr int c_Lt = c_T1 + NLV * T * 6;            // levels x T x T
    static constexpr int Jtt = c_Lt + NLV * T * T;             // levels x (N x T): J_task transposed, helper phase 1 -> main phase 3, helper phase 4
    static constexpr int c_Vb = Jtt + NLV * N * T;             // M x K
    static constexpr int c_VG = c_Vb + M * K;                  // M x K
    static constexpr int hs = c_VG + M * K;                    // helper's small scratch: 4 x 36 + 72
    static constexpr int hs_size = 4 * 36 + 72;
    static constexpr int ms = hs + hs_size;                    // main's small scratch: sweep column (64) + C x C
    static constexpr int c_s1 = ms;
    static constexpr int c_s2 = ms + max2(C * K, 64);
    static constexpr int c_Lam = c_s2;
    static constexpr int ms_size = max2(C * K, 64) + C * C;
    static constexpr int kin = ms + ms_size;                   // stage-0 scratch of the main wave; afterwards Y, then stage 3a / QP scratch
    static constexpr int k_Iw = kin;
    static constexpr int k_Ic = k_Iw + NB * 10;
    static constexpr int k_Rl = k_Iw + NB * 3;
    static constexpr int k_anc = k_Ic + NB * 10 - 2 * NB;
    static_assert(k_Rl + NB * 9 <= k_anc, "local rotations must not reach the ancestor indices");
    static constexpr int k_S = k_Ic + NB * 10;
    static constexpr int k_F = k_S + N * 6;
    static constexpr int kin_end = k_F + N * 6;
    static constexpr bool a_overlay = false, a_packed = true;
    static constexpr int k_A = U;                              // packed lower triangle over U / T1x (written from phase 3 / stage 3a on)
    static_assert(N * (N + 1) / 2 <= (c_T1 - U), "the staged mass matrix borrows the U / T1x region");
    static constexpr int c_Y = kin;                            // N x C (phase 2-3)
    static_assert(C * N <= kin_end - kin, "Y borrows the stage-0 scratch");
    // stage 3a scratch of the main wave (phase 5) and the QP scratch, over the dead stage-0 scratch
    static constexpr int c_QW = kin;                           // Q (slow route)
    static constexpr int c_QWp = c_QW + T * M;                 // Q W^+ (slow route)
    static constexpr int c_Pi = c_QWp + T * M;
    static constexpr int c_Z = c_Pi + T * T;
    static constexpr int c_s2b = c_Z + T * T;
    static constexpr int cod_Q = c_s2b + T * T, cod_v = cod_Q + T * T, cod_G = cod_v + 3 * T, cod_T = cod_G + T * T;
    static constexpr int Xl = cod_T + T * T;                   // X of levels 1 .. NLV-2
    static constexpr int xl(int lv) { return lv == 0 ? U : Xl + (lv - 1) * M * T; }
    static constexpr int t_base = Xl + (NLV > 2 ? (NLV - 2) * M * T : 0);
    static constexpr int wm = ev(t_base + M);
    static constexpr int t_fv = wm + C * WLD;
    static constexpr int t_wacc = t_fv + C;
    static constexpr int t_cl = t_wacc + C;
    static constexpr int qp_V = t_cl + K;
    static constexpr int qp_x = qp_V + kQpLd;
    static constexpr int p5_end = qp_x + kQpLd;
    static constexpr int total = max2(kin_end, p5_end);
    static constexpr int total_bytes = total * (int)sizeof(real_t) + 64;
    // names of the other maps that shared helpers mention but this kernel does not use
    static constexpr int FNl = NwJw, comp = flg, Jcm = 0, Pt = 0, c_Gi = 0, T1r = T1x, c_Q = T1x, c_Jt = Jtt, t_F = wm, t_s1 = wm;
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
#if defined(DWBC_STAGE_TIMERS) && !defined(DWBC_HOST_EMU)
#define DWBC_PSTAMP(i) do { DWBC_SYNC(); if (diag && th.tid == 0) diag[DG_FTIME + (i)] = (int)(clock64() - t_start_); } while (0)
#else
#define DWBC_PSTAMP(i) ((void)0)
#endif

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
    if (is_main) {
        // world inertias, composite inertias, S, F, CRBA, A -> registers, A^-1  (dwbc_cycle2_stage0.inc)
        real_t *Iw = L + S::k_Iw;
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
        real_t *Icm = L + S::k_Ic;
        {
            real_t *Sc = Iw, *Sn = Icm;
            PLA(real_t, sk, 10);
            PLA(real_t, acc, 10);
            PL(int, len);
            LANES {
                const int bi = lane < nb ? lane : 0;
                LV(len) = lane < nb ? topo[2 * nb + bi] : 0;
#pragma unroll
                for (int c = 0; c < 10; c++) { LV(sk)[c] = Sc[bi * 10 + c]; LV(acc)[c] = real_t(0.0); }
            }
            for (int kbit = 0, off = 1; off <= nb; kbit++, off <<= 1) {
                LANES {
                    const bool take = (LV(len) >> kbit) & 1;
                    int pos = lane + (LV(len) & (off - 1));
                    pos = (take && pos < nb) ? pos : 0;
                    const bool nbr = lane + off < nb;
                    const int pn2 = nbr ? lane + off : 0;
                    real_t a_[10], b_[10];
#pragma unroll
                    for (int c = 0; c < 10; c++) { a_[c] = Sc[pos * 10 + c]; b_[c] = Sc[pn2 * 10 + c]; }
#pragma unroll
                    for (int c = 0; c < 10; c++) {
                        LV(acc)[c] += take ? a_[c] : real_t(0.0);
                        LV(sk)[c] += nbr ? b_[c] : real_t(0.0);
                        if (lane < nb) Sn[lane * 10 + c] = LV(sk)[c];
                    }
                }
                DWBC_SYNC();
                { real_t *t_ = Sc; Sc = Sn; Sn = t_; }
            }
            LANES {
                if (lane < nb) {
#pragma unroll
                    for (int c = 0; c < 10; c++) Icm[lane * 10 + c] = LV(acc)[c];
                }
            }
            DWBC_SYNC();
        }
        real_t *Sm = L + S::k_S, *Fm = L + S::k_F;
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
        LANES {
            const int col = lane < N ? lane : 0;
            const int cbase = col * (col + 1) / 2;
#pragma unroll
            for (int i = 0; i < N; i++) LV(s)[i] = (lane < N) ? A[i >= col ? i * (i + 1) / 2 + col : cbase + i] : real_t(0.0);
            LV(dg) = (lane < N) ? A[cbase + col] : real_t(1.0);
            if (lane < N) L[S::G + lane] = kGrav * A[col >= 2 ? cbase + 2 : 3 + col];  // G_ = 9.81 A[2,:] (dwbc.cpp:358)
        }
        if (!sweep_inverse_tree<Topo, N>(s, dg)) st_contact = 0;  // A_inv (dwbc.cpp:307)
    }
    if (is_help) {
        // contact frames, J_C, internal-wrench basis and its Gram algebra; the task Jacobians of every level
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
        if (k > 0) {
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
    DWBC_PAIR_BARRIER(1);  // ---- B1: A^-1 in the main wave's registers; J_C, Vb, VG, J_t in LDS

    // ================= phase 2 (main): Y = J_C A^-1, Lambda_c, Jbar^T =================
    const unsigned long long cm0 = nc > 0 ? su.c_dofmask[act_c[0]] : 0ull, cm1 = nc > 1 ? su.c_dofmask[act_c[1]] : 0ull;
    if (is_main) {
        if (too_many) st_contact = 0;
        for (int idx = th.tid; idx < C * N; idx += NT) Yt[idx] = real_t(0.0);
        DWBC_SYNC();
        LANES {
            real_t yc[C];
#pragma unroll
            for (int p = 0; p < C; p++) yc[p] = real_t(0.0);
            // three columns of J_C (rows of its transpose) at a time, both contacts together: the entries of the contact that a dof
            // does not move are exact zeros, and one batch of 18 reads beats two half-filled ones behind their own waits
            static_assert(N % 3 == 0, "column blocks of three");
#pragma unroll
            for (int ib = 0; ib < N; ib += 3) {
                if (((cm0 | cm1) >> ib) & 7) {
                    const real_t x3[3] = {LV(s)[ib], LV(s)[ib + 1], LV(s)[ib + 2]};
                    lds_rows_axpy<3, C, C, S::c_JC % 2 == 0>(JCt + ib * C, x3, yc);
                }
            }
            if (lane < N) {
#pragma unroll
                for (int p = 0; p < C; p++) Yt[lane * C + p] = yc[p];
            }
        }
        DWBC_SYNC();
#if !defined(DWBC_HOST_EMU)
        if constexpr (sizeof(real_t) == 8) {  // J A^-1 J^T = Y J_C^T on one accumulator tile (see dwbc_cycle2_stage1.inc)
            typedef double lc_d4 __attribute__((ext_vector_type(4)));
            const int li = lane & 15, lk = lane >> 4;
            lc_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s_ = 0; s_ < (N + 3) / 4; s_++) {
                const int c = 4 * s_ + lk;
                const bool in = c < N;
                const int cc = in ? c : N - 1;
                real_t av = Yt[cc * C + (li < C ? li : 0)], bv = JCt[cc * C + (li < C ? li : 0)];
                av = in ? av : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
            if (li < C) {
#pragma unroll
                for (int r = 0; r < 3; r++) L[S::c_s2 + (lk + 4 * r) * C + li] = acc[r];
            }
        } else
#endif
        for (int idx = th.tid; idx < C * C; idx += NT) {
            const int i = idx / C, j = idx - i * C;
            real_t a4[4] = {real_t(0.0), real_t(0.0), real_t(0.0), real_t(0.0)};
            if (i < cd && j < cd) {
#pragma unroll
                for (int c = 0; c < N; c++) a4[c & 3] += Yt[c * C + i] * JCt[c * C + j];
            }
            L[S::c_s2 + idx] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        }
        if (cd > 0) {
            if (!spd_inverse_small(L + S::c_s2, C, cd, Lam, C, L + S::c_s1)) st_contact = 0;  // Lambda_c (wbd.cpp:115)
        }
        DWBC_SYNC();
    }
    // Jbar^T = Lambda J A^-1 (wbd.cpp:116) is written over J_C: keep it in registers until the barrier has been passed? no -- J_C is
    // read by this wave only from here on (the helper's readers of J_C finished before B1), so the overwrite is safe at once.
    PLA(real_t, jbk, C);  // main wave: column `lane` of Jbar^T, kept for the A^-1 N_c update of phase 3
    PLA(real_t, yck, C);
    if (is_main) {
        LANES {
            const int col = lane < N ? lane : 0;
#pragma unroll
            for (int p = 0; p < C; p++) LV(yck)[p] = Yt[col * C + p];
            lds_rows_dot<C, C, C, 4, 0, S::c_Lam % 2 == 0>(Lam, LV(yck), LV(jbk));  // column `lane` of Jbar^T = Lambda_c Y
        }
        DWBC_SYNC();  // every lane has read what it needs of Y; J_C is dead
        LANES {
#pragma unroll
            for (int p = 0; p < C; p++)
                if (lane < N) JbT[p * N + lane] = LV(jbk)[p];
        }
    }
    DWBC_PAIR_BARRIER(2);  // ---- B2: Jbar^T in LDS

    // ================= phase 3 =================
    if (is_main) {
        // A^-1 N_c = A^-1 - Y^T Jbar^T (wbd.cpp:117-118), gravity pre-vector, P_C, T1 of every level
        LANES {
            real_t dsub = real_t(0.0);
#pragma unroll
            for (int p = 0; p < C; p++) dsub += LV(yck)[p] * LV(jbk)[p];
            LV(dg) -= dsub;
            lds_rows_dot<N, C, C, 4, 1>(Yt, LV(jbk), LV(s));  // s[i] -= Y^T[i, :] . Jbar^T[:, lane]
        }
        DWBC_SYNC();
        LANES {
            real_t gv1[1];
            lds_rows_dot<1, N, 2 * ((N + 1) / 2), 1, 0>(L + S::G, LV(s), gv1);  // gravity pre-vector (A^-1 N_c G)[lane]
            if (lane < N) L[S::c_vec + lane] = gv1[0];
        }
        mv_n<NT>(th, L + S::PC, JbT, N, L + S::G, cd, N);
        for (int lv = 0; lv < su.n_levels; lv++) {
            const real_t *Jtt = L + S::Jtt + lv * N * T;
            real_t *T1 = L + S::c_T1 + lv * T * 6, *T1x = L + S::T1x + lv * T * S::MS;
            const unsigned long long tm = su.t_dofmask[lv];
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
                        if (lane < 6) T1[r * 6 + lane] = tc_[r];
                        else if (lane < N) T1x[r * S::MS + (lane - 6)] = tc_[r];
                    }
                }
            };
            if (su.t_dof[lv] <= 3) t1_rows(std::integral_constant<int, 3>{}); else t1_rows(std::integral_constant<int, T>{});
        }
        DWBC_SYNC();
    }
    if (is_help && k > 0) {
        // NwJw = VG X^T with X = (JV G^-1 JV^T)^-1 JV, JV = Jbar[0:k, 6:] Vb (dwbc_cycle2.h: SPD inverses only)
        constexpr int K6 = 6;
        real_t *JV = L + S::hs, *Gi = L + S::hs + 36, *Bm = L + S::hs + 72, *Sm6 = L + S::hs + 108;
        for (int idx = th.tid; idx < K6 * K6; idx += NT) {
            const int i = idx / 6, j = idx - i * 6;
            real_t acc = real_t(0.0);
            _Pragma("unroll 8")
            for (int c = 0; c < M; c++) acc += JbT[i * N + 6 + c] * Vb[c * K6 + j];
            JV[idx] = acc;
        }
        DWBC_SYNC();
        mm_nn<NT>(th, Bm, K6, JV, K6, Gi, K6, K6, K6, K6);                       // B = JV G^-1
        DWBC_SYNC();
        mm_nt<NT>(th, Sm6, K6, Bm, K6, JV, K6, K6, K6, K6);                      // S = B JV^T  (SPD)
        DWBC_SYNC();
        if (!spd_inverse_small(Sm6, K6, K6, Sm6, K6, L + S::hs + 144)) { if (th.tid == 0) flg[0] = real_t(0.0); }
        mm_nn<NT>(th, Bm, K6, Sm6, K6, JV, K6, K6, K6, K6);                      // X = S^-1 JV
        DWBC_SYNC();
        mm_nt<NT>(th, L + S::NwJw, K6, VG, K6, Bm, K6, M, K6, K6);              // NwJw = VG X^T
        DWBC_SYNC();
    }
    DWBC_PAIR_BARRIER(3);  // ---- B3: T1 of every level, NwJw in LDS

    // ================= phase 4 =================
    PLA(real_t, w, M);  // main wave: column `lane` of W -> W^+
    PL(real_t, dw);
    if (is_main) {
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
        PLA(real_t, vbr, 6);
        PLA(real_t, pc, M);
        LANES {
#pragma unroll
            for (int a = 0; a < 6; a++) LV(vbr)[a] = (k > 0 && lane < M) ? Vb[lane * 6 + a] : real_t(0.0);
#pragma unroll
            for (int i = 0; i < M; i++) LV(pc)[i] = real_t(0.0);
            if (k > 0) {
                DWBC_LANE_OPAQUE(lw);
                real_t dp = real_t(0.0);
                lds_rows_dot<M, 6, 6, 6, 0>(VG, LV(vbr), LV(pc));  // column `lane` of P = VG Vb^T
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
        DWBC_SYNC();
        if (!sweep_inverse_lds<M>(w, dw, L + S::c_s1)) st_contact = 0;
        DWBC_SYNC();
        LANES {
            if (k > 0) {
#pragma unroll
                for (int i = 0; i < M; i++) LV(w)[i] -= ialpha * LV(pc)[i];
            }
            real_t tg1[1];  // torque_grav_ = W^+ (A^-1 N_c G)[6:]   (wbd.cpp:190)
            lds_rows_dot<1, M, 2 * ((M + 1) / 2), 1, 0>(L + S::c_vec + 6, LV(w), tg1);
            if (lane < M) L[S::tg + lane] = tg1[0];
        }
        DWBC_SYNC();
    }
    if (is_help) {
        // J_t A^-1 N_c J_t^T, Lambda_task (wbd.cpp:210) and the condition verdict of every level (dwbc_cycle2.h, task-Jacobian stage)
        int fm = 0;
        for (int lv = 0; lv < su.n_levels; lv++) {
            const int t = su.t_dof[lv];
            const real_t *Jtt = L + S::Jtt + lv * N * T;
            const real_t *T1 = L + S::c_T1 + lv * T * 6, *T1x = L + S::T1x + lv * T * S::MS;
            real_t *Lt = L + S::c_Lt + lv * T * T, *Li = L + S::hs;
            const unsigned long long tm = su.t_dofmask[lv];
            DWBC_SYNC();
            for (int idx = th.tid; idx < t * t; idx += NT) {
                const int i = idx / t, j = idx - i * t;
                real_t acc = real_t(0.0);
#pragma unroll
                for (int c = 0; c < N; c++)
                    if ((tm >> c) & 1) acc += (c < 6 ? T1[i * 6 + c] : T1x[i * S::MS + (c < 6 ? 0 : c - 6)]) * Jtt[c * T + j];
                Li[idx] = acc;
            }
            const int ok_lt = spd_inverse_small(Li, t, t, Lt, t, L + S::hs + 144);
            real_t da = real_t(0.0), dl = real_t(0.0);
            for (int i = 0; i < t; i++) {
                const real_t a_ = Li[i * t + i], l_ = Lt[i * t + i];
                da = a_ > da ? a_ : da;
                dl = l_ > dl ? l_ : dl;
            }
            if (ok_lt && nc > 0 && da * dl < kCodCondFast) fm |= 1 << lv;
        }
        DWBC_SYNC();
        if (th.tid == 0) flg[1] = (real_t)fm;
    }
    DWBC_PAIR_BARRIER(4);  // ---- B4: W^+ in the main wave's registers; Lambda_task, the fast-route mask in LDS
    if (!is_main) return;

    // ================= phase 5 (main): stage 3a, wrench maps, QP cascade, outputs (dwbc_cycle2.h) =================
    if (flg[0] == real_t(0.0)) st_contact = 0;
    const int fastmask = (int)flg[1];
    int rankbad = 0;
    for (int lv = 0; lv < su.n_levels; lv++) {
        auto level_body = [&](auto ttl) {
        constexpr int TTL = decltype(ttl)::value;
        const int t = TTL;
        const real_t *Lt = L + S::c_Lt + lv * T * T;
        const FastDiv fdt(t);
        const real_t *T1rl = L + S::T1x + lv * T * S::MS;
        real_t *Q = L + S::c_QW, *QW = L + S::c_QWp, *Pi = L + S::c_Pi;
        real_t *Ul = L + S::U + lv * M * T;
        real_t *Xs = (lv < NLV - 1) ? L + S::xl(lv) : Ul;
        int cond = 1;
        DWBC_SYNC();
        const bool fast = (fastmask >> lv) & 1;
        if (fast) {
            LANES {
                real_t tw[TTL];
                lds_rows_dot<TTL, M, S::MS, 1, 0>(T1rl, LV(w), tw);  // (T1r W^+)[r][lane] = J_kt[lane][r]
#pragma unroll
                for (int r3 = 0; r3 < TTL; r3++) {
                    real_t acc = real_t(0.0);
#pragma unroll
                    for (int r2 = 0; r2 < TTL; r2++) acc += tw[r2] * Lt[r2 * TTL + r3];
                    if (lane < M) {
                        Xs[lane * T + r3] = acc;
                        Ul[lane * T + r3] = acc;
                    }
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
            for (int r = 0; r < TTL; r++) {
                LANES {
                    real_t a4[4] = {real_t(0.0), real_t(0.0), real_t(0.0), real_t(0.0)};
#pragma unroll
                    for (int i = 0; i < M; i++) a4[i & 3] += Q[r * M + i] * LV(w)[i];
                    const real_t acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
                    if (lane < M) QW[r * M + lane] = acc;
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
                    if (lane < M) {
                        Xs[lane * T + r3] = acc;
                        Ul[lane * T + r3] = acc;
                    }
                }
            }
        }
        DWBC_SYNC();
        DWBC_PSTAMP(16 + 2 * lv);  // rows of J_kt / X of level lv written
        for (int pl = lv - 1; pl >= 0; pl--) {  // U <- (I - X_pl Y_pl) U,  Y_pl = T1x[pl]
            const int tp = su.t_dof[pl];
            const real_t *Xp = L + S::xl(pl), *Yp = L + S::T1x + pl * T * S::MS;
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
        DWBC_PSTAMP(17 + 2 * lv);  // null-space chain of level lv done
            };
        if (su.t_dof[lv] <= 3) level_body(std::integral_constant<int, 3>{}); else level_body(std::integral_constant<int, T>{});
    }
    DWBC_SYNC();
    DWBC_STAMP(5);  // stage 3a done

    // ---- the QP cascade (dwbc.cpp:818-873, 941-1127) and the contact redistribution QP (dwbc.cpp:1372-1568): dwbc_cycle2.h
    const int nlim = su.has_tau_lim ? 2 * M : 0;
    const int ncone = 10 * nc;
    int st_task = 1, fail_level = -1, st_redis = 1;
    const real_t *fs_in = L + S::fs;
    real_t *base = L + S::t_base, *fv = L + S::t_fv, *WM = L + S::wm, *wacc = L + S::t_wacc, *clast = L + S::t_cl;
    constexpr int WLD = S::WLD;
    const int colN = 1 + su.fstar_total;
    QpLaneConst qc;
    PL(real_t, sfin);  // slack of the lane's QP row at the point the last QP returned (qp_solve_wave)
    PL(real_t, grn);   // norm of the lane's row of the contact redistribution QP
    bool skip_redis = false;
    constexpr bool wm_ok = true;
    qp_lane_consts<N>(su, act_c[0], act_c[1], qc);
    wrench_maps<N, NT, S>(th, su, L, JbT, cd, k, WM);
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
    redis_row_norms<N, S>(L, nlim, ncone, k, qc, WM + colN, grn);
    DWBC_STAMP(6);  // wrench maps done
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
        const real_t *Ul = L + S::U + (is_task ? qi : 0) * M * T;
        const real_t *fs = fs_in + (is_task ? su.fstar_off[qi] : 0);
        const int colL = 1 + (is_task ? su.fstar_off[qi] : 0);
        if (is_task && (rankbad & (1 << qi))) { st_task = 0; fail_level = qi; continue; }
        DWBC_SYNC();
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
        if (qi == 0) DWBC_PSTAMP(24);  // level 0: base torque and wrench right-hand side
        if (!is_task) DWBC_PSTAMP(26);  // redistribution: base torque and wrench right-hand side
        QpResult qres;
        {
            const real_t *P1 = is_task ? Ul : L + S::NwJw;
            static_assert(T == 6, "stride of U equals the stride of NwJw");
            const int n1 = is_task ? t : k, n2 = is_task ? k : 0;
            const real_t *W1 = WM + (is_task ? colL : colN);
            qp_rows_and_solve<N, NB, 0>(su, L, nlim, ncone, act_c[0], act_c[1], P1, 6, n1, L + S::NwJw, 6, n2,
                                        is_task ? kQpScaleGI : real_t(1.0), W1, WLD, WM + colN, WLD, fv, base, n1,
                                        is_task ? su.qp_max_iter_task : su.qp_max_iter_contact, qres, L + S::qp_V, L + S::qp_x, nullptr, &qc,
                                        is_task ? kQpTol : kQpFeasTol, sfin);
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
        const real_t *x = L + S::qp_x;
        if (is_task) {
            if (!qres.status) { st_task = 0; fail_level = qi; continue; }
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
            if (qi == su.n_levels - 1 && k > 0 && (true)) {
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
        } else if (qres.status) {
            for (int i = th.tid; i < M + S::K; i += NT) {
                if (i >= M) { clast[i - M] += (i - M < k) ? x[i - M] : real_t(0.0); continue; }  // contact_qp_ + redistribution: what torque_contact_ stands for
                real_t c = real_t(0.0);
                _Pragma("unroll 8")
                for (int j = 0; j < k; j++) c += L[S::NwJw + i * 6 + j] * x[j];
                L[S::tc + i] += c;
            }
        } else {
            st_redis = 0;
            for (int i = th.tid; i < M + S::K; i += NT) {
                if (i < M) L[S::tc + i] = real_t(0.0); else clast[i - M] = real_t(0.0);
            }
        }
        DWBC_SYNC();
        if (qi == 0) DWBC_PSTAMP(25);  // level 0: committed (torque_task_, torque_contact_, running wrench)
        if (!is_task) DWBC_PSTAMP(27);  // redistribution QP done and committed
    }
    if (k == 0) {
        for (int i = th.tid; i < M; i += NT) L[S::tc + i] = real_t(0.0);
    }
    DWBC_SYNC();
    io_t *tau = io.tau + (size_t)inst * 3 * M;
    for (int i = th.tid; i < 3 * M; i += NT) tau[i] = too_many ? real_t(0.0) : L[S::tg + i];
    DWBC_PSTAMP(28);  // torques stored
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
        wr[i] = acc;
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

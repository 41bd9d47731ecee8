// dwbc_cycle_gc.h -- the cycle for MORE THAN TWO simultaneously active 6D contacts (the reference stacks any number of flagged
// contacts: src/dwbc.cpp:445-453; its own tests register both hands next to the feet: tests/dwbc_test.cpp:68-69).
//
// The product kernels (dwbc_cycle2.h, dwbc_cycle2p.h) are built around two contacts: 12 contact rows fit the register / LDS budget
// that gives them their speed, and their small blocks are unrolled for k in {0, 6}.  This kernel is the general statement of the same
// cycle, one wavefront per instance, every matrix resident in LDS (80 KB: two workgroups per CU) and every loop over run-time
// dimensions cd = 6 nc, k = cd - 6, t (the dense products on matrix-core tiles, mmg below; the inverses in registers) -- correctness first; a batch opts into it with dwbc_batch_set_max_active_contacts(b, 3).
// NCC = 3: 18 contact rows, 12 contact-null variables, QPs of up to 18 variables and 33 + 30 = 63 rows -- still one row per lane of
// the wave-level active-set solver (dwbc_qp_wave.h, instantiated for 18 variables).  Four contacts would need 73 lanes: not built.
//
// Same arithmetic as the product path wherever the reference fixes it (stage by stage below, reference file:line); where the product
// kernels use a closed form (null(W) from the internal-wrench basis, W^+ from one SPD sweep of W + alpha P) this kernel uses the same
// closed form for any number of contacts.  Lean scope: hqp = true, link tasks, f* from SetTaskSpace, cold-started QPs.
#pragma once
#include "dwbc_cycle2.h"
#include "dwbc_cycle2p.h"  // wave_gemm

namespace dwbc {

// C (m x n) = op(A) op(B) for LDS-resident matrices of run-time size on the matrix cores (wave_gemm, dwbc_cycle2p.h): tiles of 16 x 16
// up to the compile-time bounds MAXM x MAXN x MAXK, every entry outside the run-time sizes read as zero and not stored.
// OP 0: A (m x k) B (k x n);  1: A (m x k) B^T (B n x k);  2: A^T (A k x m) B (k x n).  The thread-per-output loops this replaces
// (mm_nn / mm_nt / mm_tn of dwbc_cycle.h: every term of a dot product waits out its own pair of LDS reads) were 60 % of this
// kernel's stage 1.  C must not alias A or B.  One wave; the caller fences.
template <int MAXM, int MAXN, int MAXK, int OP>
DWBC_WDEV void mmg(real_t *Cm, int ldc, const real_t *A, int lda, const real_t *B, int ldb, int m, int kk, int n) {
    constexpr int NCT = (MAXN + 15) / 16;
    static_for<0, NCT>([&](auto ctc) {
        constexpr int c0 = 16 * decltype(ctc)::value;
        if (c0 < n) {
            wave_gemm<MAXM, (MAXN - c0 < 16 ? MAXN - c0 : 16), MAXK, false>(
                [&](auto, auto, int i, int k_) {
                    const bool in = i < m && k_ < kk;
                    const int ii = i < m ? i : 0, k2 = k_ < kk ? k_ : 0;
                    const real_t v_ = OP == 2 ? A[k2 * lda + ii] : A[ii * lda + k2];
                    return in ? v_ : real_t(0.0);
                },
                [&](auto, int k_, int j) {
                    const int col = c0 + j;
                    const bool in = k_ < kk && col < n;
                    const int cc = col < n ? col : 0, k2 = k_ < kk ? k_ : 0;
                    const real_t v_ = OP == 1 ? B[cc * ldb + k2] : B[k2 * ldb + cc];
                    return in ? v_ : real_t(0.0);
                },
                [&](int i, int j, real_t d) {
                    if (i < m && c0 + j < n) Cm[i * ldc + c0 + j] = d;
                });
        }
    });
}

// TG: task dof per level the map and the QPs are sized for (6: the product kernels' limit; 12: two 6D links on one level)
template <int N, int NB, int NCC, int TG = kMaxTaskDof>
struct LdsG {
    static constexpr int M = N - 6;
    static constexpr int C = 6 * NCC;
    static constexpr int K = C - 6;
    static constexpr int T = TG;
    static constexpr int QN = T + K;                  // QP variables: task block + contact-null block
    static constexpr int max2(int a, int b) { return a > b ? a : b; }
    // persistent
    static constexpr int bufA = 0;                    // A -> A^-1 ; later W^+ (M x M)
    static constexpr int bufN = bufA + N * N;         // A^-1 N_c
    static constexpr int Rw = bufN + N * N;           // body -> world rotations
    static constexpr int pw = Rw + NB * 9;
    static constexpr int aw = pw + NB * 3;            // world joint axes
    static constexpr int JC = aw + NB * 3;            // C x N
    static constexpr int JbT = JC + C * N;            // Jbar_c^T  C x N
    static constexpr int Lam = JbT + C * N;           // C x C
    static constexpr int NwJw = Lam + C * C;          // M x k
    static constexpr int FNl = NwJw + M * K;          // A_rot Jbar[:,6:] NwJw   cd x k
    static constexpr int G = FNl + C * K;
    static constexpr int tg = G + N;                  // torque_grav_ | torque_task_ | torque_contact_
    static constexpr int tt = tg + M;
    static constexpr int tc = tt + M;
    static constexpr int PC = tc + M;                 // C
    static constexpr int q = PC + C;                  // N + 1
    static constexpr int Rc = q + N + 1;              // contact rotations (active)
    static constexpr int Pc = Rc + NCC * 9;           // contact points (world)
    static constexpr int Xl = Pc + NCC * 3;           // per level X = J_kt Lambda (M x T)
    static constexpr int Yl = Xl + (kMaxLevels - 1) * M * T;   // per level Y = (J_t A^-1 N_c)[:,6:] (T x M)
    static constexpr int tmp = Yl + (kMaxLevels - 1) * T * M;  // phase-local scratch
    // --- scratch, kinematics phase
    static constexpr int k_Rl = tmp;
    static constexpr int k_Iw = k_Rl + NB * 9;
    static constexpr int k_Ic = k_Iw + NB * 10;
    static constexpr int k_S = k_Ic + NB * 10;
    static constexpr int k_F = k_S + N * 6;
    static constexpr int k_end = k_F + N * 6;
    // --- scratch, contact phase.  Y = J_C A^-1 is dead once A^-1 N_c exists, before the projector P is formed: P takes its place
    //     (and reaches past it, so Vb and the small scratch sit behind P's end).  W + alpha P is built and inverted in bufA (A^-1 is
    //     dead by then).  78 KB instead of 93: two workgroups per CU.
    static constexpr int c_P = tmp;                   // M x M projector on null(W)
    static constexpr int c_Y = tmp;                   // J_C A^-1 (C x N)
    static constexpr int c_Vb = c_P + max2(M * M, C * N);  // M x k (behind P and behind Y, whichever reaches further: small models)
    static constexpr int c_s1 = c_Vb + M * K;         // C x 2C Gauss-Jordan scratch
    static constexpr int c_s2 = c_s1 + C * 2 * C;     // C x C
    static constexpr int c_VG = c_s2 + C * C;         // M x k: Vb G^-1
    static constexpr int c_vec = c_VG + M * K;        // N
    static constexpr int c_end = c_vec + N;
    // --- scratch, task / QP phase
    static constexpr int t_Jt = tmp;                  // T x N
    static constexpr int t_T1 = t_Jt + T * N;         // T x N
    static constexpr int t_Lt = t_T1 + T * N;         // T x T
    static constexpr int t_Q = t_Lt + T * T;          // T x M
    static constexpr int t_QW = t_Q + T * M;          // T x M
    static constexpr int t_Jkt = t_QW + T * M;        // M x T
    static constexpr int t_U = t_Jkt + M * T;         // M x T
    static constexpr int t_s1 = t_U + M * T;          // max(T x 2T, C x (T+1))
    static constexpr int t_s2 = t_s1 + max2(T * 2 * T, C * (T + 1));  // T x T
    static constexpr int t_s3 = t_s2 + T * T;         // T x T
    static constexpr int t_cod = t_s3 + T * T;        // scratch of pinv_cod_small: 3 x (T x T) + 3 T
    static constexpr int t_base = t_cod + 3 * T * T + 3 * T;  // M
    static constexpr int t_F = t_base + M;            // C x T
    static constexpr int t_fv = t_F + C * T;          // C
    static constexpr int qp_V = t_fv + C;             // QN
    static constexpr int qp_x = qp_V + QN;            // QN
    static constexpr int t_end = qp_x + QN;
    static constexpr int colA = (tmp + 1) & ~1, colW = (c_s1 + 1) & ~1;  // 16-byte aligned pivot-column buffers of the two big sweeps (64 doubles each)
    static_assert(colA + 64 <= k_end && colW + 64 <= c_s2, "pivot-column buffers");
    static constexpr int total = max2(max2(k_end, c_end), t_end);
    static constexpr int total_bytes = total * (int)sizeof(real_t) + 64;
};

// SPD inverse with one column per lane in registers (the symmetric sweep of dwbc_cycle2.h), any size up to 56: Sin NN x NN (ld) ->
// Out (ldo).  Same arithmetic role as Eigen's llt().solve(I) (reference src/dwbc.cpp:307).
// colbuf: 64 doubles of LDS -- the pivot column goes through it (sweep_inverse_lds: one ds_write_b64 and the column read back in 16-byte
// broadcast pieces per pivot, against two v_readlane and a hazard s_nop per ROW of every pivot); nullptr: the v_readlane feed
template <int NN>
DWBC_DEVN int spd_inverse_reg(const real_t *Sin, int ld, real_t *Out, int ldo, real_t *colbuf = nullptr) {
    DWBC_LANE_DECL;
    PLA(real_t, s, NN);
    PL(real_t, dg);
    DWBC_SYNC();
    LANES {
        const int col = lane < NN ? lane : 0;
#pragma unroll
        for (int i = 0; i < NN; i++) LV(s)[i] = (lane < NN) ? Sin[i * ld + col] : real_t(0.0);
        LV(dg) = (lane < NN) ? Sin[col * ld + col] : real_t(1.0);
    }
    DWBC_SYNC();
    const int ok = colbuf ? sweep_inverse_lds<NN>(s, dg, colbuf) : sweep_inverse_rl<NN>(s, dg, NN);
#if !defined(DWBC_HOST_EMU)
#pragma unroll
    for (int i = 0; i < NN; i++) asm volatile("" : "+v"(s[i]));  // (the stores below are conditional: see sweep_inverse_tree_lds, dwbc_cycle2.h)
#endif
    DWBC_SYNC();
    LANES {
        if (lane < NN) {
#pragma unroll
            for (int i = 0; i < NN; i++) Out[i * ldo + lane] = LV(s)[i];
        }
    }
    DWBC_SYNC();
    return ok;
}

// SPD inverse of the leading n x n block (n <= NN <= 24) by the same register sweep on the Jacobi-scaled matrix D^-1 A D^-1, D = sqrt(diag A)
// (see spd_inverse_small, dwbc_cycle2.h: a hand next to a foot mixes rows of 1e-1 with rows of 1e3).  Sin (ld) -> Out (ldo), in place allowed;
// colbuf: NN doubles of LDS.  Used for the Gram matrix of the internal-wrench basis (well conditioned).
template <int NN>
DWBC_DEVN int spd_inverse_scaled(const real_t *Sin, int ld, int n, real_t *Out, int ldo, real_t *colbuf) {
    DWBC_LANE_DECL;
    PLA(real_t, s, NN);
    PL(real_t, dg);
    PL(real_t, dsc);
    DWBC_SYNC();
    LANES {
        const int col = lane < n ? lane : 0;
        const real_t a = Sin[col * ld + col];
        // a power of two next to 1 / sqrt(a): the scaling is then exact (no rounding of its own) and costs two instructions
        int e2 = 0;
        (void)frexp(a > real_t(0.0) ? a : real_t(1.0), &e2);
        LV(dsc) = (lane < n && a > real_t(0.0)) ? ldexp(real_t(1.0), -(e2 >> 1)) : real_t(1.0);
        if (lane < NN) colbuf[lane] = LV(dsc);
    }
    DWBC_SYNC();
    LANES {
        const int col = lane < n ? lane : 0;
        const real_t dc = LV(dsc);
#pragma unroll
        for (int i = 0; i < NN; i++) LV(s)[i] = (lane < n && i < n) ? Sin[i * ld + col] * colbuf[i] * dc : real_t(0.0);
        LV(dg) = (lane < n) ? Sin[col * ld + col] * dc * dc : real_t(1.0);
    }
    const int ok = sweep_inverse_rl<NN>(s, dg, n);
    DWBC_SYNC();
    LANES {
        if (lane < n) {
#pragma unroll
            for (int i = 0; i < NN; i++)
                if (i < n) Out[i * ldo + lane] = LV(s)[i] * colbuf[i] * LV(dsc);
        }
    }
    DWBC_SYNC();
    return ok;
}

// Inverse of a small general matrix as the reference's .inverse() behaves where it matters here: Gauss-Jordan with partial pivoting on
// [A | I] (largest remaining element of the column), the matrix in REGISTERS, one row per lane: the factor of row i is the lane's own element, the scaled pivot row goes through LDS once per pivot
// (W: 2 NMAX doubles) and every lane updates its row with 2 NMAX FMAs -- no element of the augmented matrix travels through LDS per
// update (round 3 kept the augmented matrix in LDS and touched all n x 2n elements there per pivot: 45 k cycles for 18 x 18, ~17 k now).  Rows are never exchanged: the lane that
// served as the pivot of column c holds row c of the inverse at the end.  NMAX: compile-time bound of n (the pivot loop is unrolled
// over it, so the pivot element is a static register).  In place allowed.  Returns the smallest / largest pivot magnitude.
template <int NMAX, int NT>
DWBC_DEVN real_t gj_inverse_rows(Thr th, const real_t *A, int lda, int n, real_t *Ai, int ldi, real_t *W) {
    DWBC_LANE_DECL;
    constexpr int W2 = 2 * NMAX;
    static_assert(NMAX <= 32, "one row per lane, 2 NMAX registers per row");
    PLA(real_t, r, W2);
    PL(int, used);
    PL(int, mycol);
    PL(real_t, pv);
    PL(int, pk);
    DWBC_SYNC();
    LANES {
        const int row = lane < n ? lane : 0;
#pragma unroll
        for (int j = 0; j < NMAX; j++) {
            const real_t v_ = A[row * lda + (j < n ? j : 0)];
            LV(r)[j] = (lane < n && j < n) ? v_ : real_t(0.0);
            LV(r)[NMAX + j] = (lane < n && j == lane) ? real_t(1.0) : real_t(0.0);
        }
        LV(used) = lane < n ? 0 : 1;
        LV(mycol) = -1;
    }
    real_t pmin = kF32 ? real_t(1e30) : real_t(1e300), pmax = real_t(0.0);
#pragma unroll
    for (int c = 0; c < NMAX; c++) {
        if (c < n) {
            LANES {
                LV(pv) = LV(used) ? DWBC_QP_INF : -fabs(LV(r)[c]);
                LV(pk) = lane;
            }
            real_t bneg;
            int p;
            WAVE_ARGMIN(pv, pk, bneg, p);
            const real_t best = -bneg;
            pmin = best < pmin ? best : pmin;
            pmax = best > pmax ? best : pmax;
            DWBC_SYNC();  // (the previous pivot row has been read by every lane)
            LANES {
                if (lane == p) {
                    const real_t piv = LV(r)[c];
                    const real_t inv = piv != real_t(0.0) ? real_t(1.0) / piv : real_t(0.0);
#pragma unroll
                    for (int j = 0; j < W2; j++) {
                        LV(r)[j] *= inv;
                        W[j] = LV(r)[j];
                    }
                    LV(used) = 1;
                    LV(mycol) = c;
                }
            }
            DWBC_SYNC();
            LANES {
                if (lane != p) {
                    const real_t f = LV(r)[c];
#pragma unroll
                    for (int j = 0; j < W2; j++) LV(r)[j] -= f * W[j];
                    LV(r)[c] = real_t(0.0);
                }
            }
        }
    }
    DWBC_SYNC();
    LANES {
        if (lane < n && LV(mycol) >= 0) {
#pragma unroll
            for (int j = 0; j < NMAX; j++)
                if (j < n) Ai[LV(mycol) * ldi + j] = LV(r)[NMAX + j];
        }
    }
    DWBC_SYNC();
    return pmax > real_t(0.0) ? pmin / pmax : real_t(0.0);
}

// QP rows into lanes + solve, for up to NCC contacts (same rows as qp_rows_and_solve of dwbc_cycle.h):
//   torque rows:  [P1 | s2 P2][r,:] x  in  [-(lim + base), lim - base]        (reference src/dwbc.cpp:1001-1016)
//   cone rows:    -cone(W1 | s2 W2)[rr,:] x <= cone(fv)[rr]                     (reference src/dwbc.cpp:1041-1053, src/wbd.cpp:59-97)
template <int N, int NCC, int TG = kMaxTaskDof>
DWBC_DEV void qp_rows_and_solve_gc(const Setup &su, int nlim, int ncone, const int *act_c, const real_t *P1, int ld1, int t1,
                                   const real_t *P2, int ld2, int t2, real_t s2, const real_t *W1, int ldw1, const real_t *W2, int ldw2,
                                   const real_t *fv, const real_t *base, int tvars, int max_iter, QpResultT<TG + 6 * NCC - 6> &res,
                                   real_t *Vlds, real_t *xlds, real_t vtol, bool fixed_layout = false) {
    constexpr int M = N - 6, QN = TG + 6 * NCC - 6;
    static_assert(M + 10 * NCC <= 64, "one QP row per lane");
    DWBC_LANE_DECL;
    QpRowsT<QN> R;
    PL(real_t, sfin);
    // fixed_layout (the task QPs): the task variables take positions 0 .. 5 and the contact-null variables positions 6 .. 17 whatever t and k
    // are, the unused positions are zero columns (variables no row touches stay zero).  The solver then always sees (t, k) = (6, 12), the layout
    // whose lexicographic solve has compile-time positions; with the packed layout a three-dof level (t = 3) or two contacts (k = 6) went
    // through per-entry selects (216 per product with H).  x comes back in the same positions.
    const int off2 = fixed_layout ? TG : t1;
    const int nv = fixed_layout ? QN : t1 + t2;
    if (fixed_layout) tvars = TG;
    LANES {
#pragma unroll
        for (int j = 0; j < QN; j++) LV(R.g)[j] = real_t(0.0);
        LV(R.hi) = DWBC_QP_INF;
        LV(R.lo) = DWBC_QP_INF;
        LV(R.id_hi) = -1;
        LV(R.id_lo) = -1;
        if (lane < M) {
            if (nlim) {
#pragma unroll
                for (int j = 0; j < QN; j++) {
                    real_t v = real_t(0.0);
                    if (j < t1) v = P1[lane * ld1 + j];
                    else if (j >= off2 && j < off2 + t2) v = P2[lane * ld2 + (j - off2)] * s2;
                    LV(R.g)[j] = v;
                }
                LV(R.hi) = (real_t)su.tau_lim[lane] - base[lane];
                LV(R.lo) = (real_t)su.tau_lim[lane] + base[lane];
                LV(R.id_hi) = lane;
                LV(R.id_lo) = M + lane;
            }
        } else if (lane - M < ncone) {
            // cone row r10 of contact a acts on the local wrench w as  c2 * w[2] + sg * w[oi]   (reference src/wbd.cpp:59-97)
            const int rr = lane - M, a = rr / 10, r10 = rr - 10 * a;
            const int ci = act_c[a];
            const int pr = r10 >> 1;
            const real_t c2 = -(real_t)(pr == 0 ? su.c_lx[ci] : pr == 1 ? su.c_ly[ci] : pr == 4 ? su.c_muz[ci] : su.c_mu[ci]);
            const int oi = pr == 0 ? 4 : pr == 1 ? 3 : pr == 2 ? 0 : pr == 3 ? 1 : 5;
            const real_t sg = (pr < 2) ? ((r10 & 1) ? real_t(1.0) : -real_t(1.0)) : ((r10 & 1) ? -real_t(1.0) : real_t(1.0));
            const int row2 = 6 * a + 2, rowo = 6 * a + oi;
#pragma unroll
            for (int j = 0; j < QN; j++) {
                real_t v = real_t(0.0);
                if (j < t1) v = c2 * W1[row2 * ldw1 + j] + sg * W1[rowo * ldw1 + j];
                else if (j >= off2 && j < off2 + t2) v = (c2 * W2[row2 * ldw2 + (j - off2)] + sg * W2[rowo * ldw2 + (j - off2)]) * s2;
                LV(R.g)[j] = -v;
            }
            LV(R.hi) = c2 * fv[row2] + sg * fv[rowo];
            LV(R.id_hi) = nlim + rr;
        }
    }
    // WS = 1: the general (t, k) layout of the lexicographic point
    qp_solve_wave<1, QN, QN, 6 * NCC - 6>(R, nv, tvars, max_iter, res, Vlds, nullptr, vtol, sfin);
    LANES {
        if (lane < QN) {
            real_t v = real_t(0.0);
#pragma unroll
            for (int i = 0; i < QN; i++) v = (lane == i) ? res.x[i] : v;
            xlds[lane] = v;
        }
    }
    DWBC_SYNC();
}

// ----------------------------------------------------------------------------------------------
// the cycle for one instance
// ----------------------------------------------------------------------------------------------
template <int N, int NB, int NCC, int NT, int TG = kMaxTaskDof>
DWBC_DEV void cycle_instance_gc(Thr th, const Setup &su, const BatchIO &io, int inst, real_t *L) {
    using S = LdsG<N, NB, NCC, TG>;
    constexpr int M = S::M, C = S::C, T = S::T, QN = S::QN;
    const int nb = su.nb;
    const real_t *body = io.body;
    const int *topo = io.topo;  // parent[nb] depth[nb] subtree[nb]
    const io_t *qin = io.q + (size_t)inst * (N + 1);
    int *diag = io.diag ? io.diag + (size_t)inst * DG_COUNT : nullptr;
    DWBC_STAMP_INIT();

    // ================= stage 0: kinematics, A, A_inv, G  (src/dwbc.cpp:279-371) =================
    for (int i = th.tid; i < N + 1; i += NT) L[S::q + i] = (real_t)qin[i];
    for (int i = th.tid; i < 3 * M; i += NT) L[S::tg + i] = real_t(0.0);
    DWBC_SYNC();
    {
        real_t *Rw = L + S::Rw, *pw = L + S::pw, *aw = L + S::aw, *Rl = L + S::k_Rl;
        const real_t *q = L + S::q;
        // local joint transforms R_T * Rot(axis, q_i)
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
        for (int d = 1; d <= su.maxdepth; d++) {
            DWBC_SYNC();
            for (int i = th.tid; i < nb; i += NT) {
                if (topo[nb + i] != d) continue;
                const int par = topo[i];
                const real_t *bd = body + i * kBodyStride;
                const real_t *Rp = Rw + par * 9;
                for (int a = 0; a < 3; a++) {
                    for (int b = 0; b < 3; b++)
                        Rw[i * 9 + a * 3 + b] = Rp[a * 3] * Rl[i * 9 + b] + Rp[a * 3 + 1] * Rl[i * 9 + 3 + b] + Rp[a * 3 + 2] * Rl[i * 9 + 6 + b];
                    pw[i * 3 + a] = pw[par * 3 + a] + Rp[a * 3] * bd[BF_PT] + Rp[a * 3 + 1] * bd[BF_PT + 1] + Rp[a * 3 + 2] * bd[BF_PT + 2];
                }
            }
        }
        DWBC_SYNC();
        // world axes, world-frame spatial inertia of each body about O = pelvis origin
        real_t *Iw = L + S::k_Iw;
        for (int i = th.tid; i < nb; i += NT) {
            const real_t *bd = body + i * kBodyStride;
            const real_t *R = Rw + i * 9;
            for (int a = 0; a < 3; a++) aw[i * 3 + a] = R[a * 3] * bd[BF_AXIS] + R[a * 3 + 1] * bd[BF_AXIS + 1] + R[a * 3 + 2] * bd[BF_AXIS + 2];
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
        // composite inertia: subtree of body i is the contiguous DFS range [i, i + subtree[i])
        real_t *Icm = L + S::k_Ic;
        {
            // inclusive prefix sums over the body order in registers (DPP), Ic[b] = P[b + len_b - 1] - P[b - 1] (dwbc_cycle2p.h: everything is
            // about the pelvis origin, where the smallest subtree is within 1e3 of the total) -- the sum over each subtree, one dependent
            // LDS read per body of it, was 8 k cycles of this stage
            static_assert(NB <= 64, "one body per lane");
            DWBC_LANE_DECL;
            PLA(real_t, pf, 10);
            PL(int, len);
            LANES {
                const int bi = lane < nb ? lane : 0;
                LV(len) = lane < nb ? topo[2 * nb + bi] : 1;
#pragma unroll
                for (int c = 0; c < 10; c++) {
                    const real_t v_ = Iw[bi * 10 + c];
                    LV(pf)[c] = lane < nb ? v_ : real_t(0.0);
                }
            }
#pragma unroll
            for (int c = 0; c < 10; c++) WAVE_PREFIX_A(pf, c);
            LANES {
                int e_ = lane + LV(len) - 1;
                e_ = e_ < 63 ? e_ : 63;
                const int s_ = lane > 0 ? lane - 1 : 0;
#pragma unroll
                for (int c = 0; c < 10; c++) {
                    const real_t hi_ = SHFLA(pf, c, e_), lo_ = SHFLA(pf, c, s_);
                    if (lane < nb) Icm[lane * 10 + c] = hi_ - (lane > 0 ? lo_ : real_t(0.0));
                }
            }
        }
        // motion axes S_j = [omega; v_O] about O
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
        for (int idx = th.tid; idx < N * N; idx += NT) L[S::bufA + idx] = real_t(0.0);
        DWBC_SYNC();
        for (int j = th.tid; j < N; j += NT) {
            const int b = j < 6 ? 0 : j - 5;
            const real_t *I = Icm + b * 10;
            const real_t *s = Sm + j * 6;
            const real_t m = I[0], h0 = I[1], h1 = I[2], h2 = I[3];
            const real_t w0 = s[0], w1 = s[1], w2 = s[2], v0 = s[3], v1 = s[4], v2 = s[5];
            // L = I w + h x v ; p = m v + w x h
            Fm[j * 6 + 0] = I[4] * w0 + I[5] * w1 + I[6] * w2 + (h1 * v2 - h2 * v1);
            Fm[j * 6 + 1] = I[5] * w0 + I[7] * w1 + I[8] * w2 + (h2 * v0 - h0 * v2);
            Fm[j * 6 + 2] = I[6] * w0 + I[8] * w1 + I[9] * w2 + (h0 * v1 - h1 * v0);
            Fm[j * 6 + 3] = m * v0 + (w1 * h2 - w2 * h1);
            Fm[j * 6 + 4] = m * v1 + (w2 * h0 - w0 * h2);
            Fm[j * 6 + 5] = m * v2 + (w0 * h1 - w1 * h0);
        }
        DWBC_SYNC();
        // A[j][k] = S_k . F_j for k on the path from j to the root (CRBA, [ext] RBDL CompositeRigidBodyAlgorithm)
        real_t *A = L + S::bufA;
        for (int j = th.tid; j < N; j += NT) {
            const real_t *f = Fm + j * 6;
            int k = j;
            for (;;) {
                const real_t *s = Sm + k * 6;
                const real_t v = s[0] * f[0] + s[1] * f[1] + s[2] * f[2] + s[3] * f[3] + s[4] * f[4] + s[5] * f[5];
                A[j * N + k] = v;
                A[k * N + j] = v;
                if (k == 0) break;
                if (k < 6) k = k - 1;
                else {
                    const int pb = topo[k - 5];
                    k = pb == 0 ? 5 : pb + 5;
                }
            }
        }
        DWBC_SYNC();
        for (int j = th.tid; j < N; j += NT) L[S::G + j] = kGrav * A[2 * N + j];  // G_ = -J_com_lin^T m g (dwbc.cpp:358)
    }
    DWBC_STAMP(0);  // kinematics + CRBA
    int st_contact = 1;
    // A_inv = llt(A).solve(I)  (dwbc.cpp:307), in place (the columns live in registers meanwhile)
    if (!spd_inverse_reg<N>(L + S::bufA, N, L + S::bufA, N, L + S::colA)) st_contact = 0;  // (pivot column over the dead kinematics scratch)
    DWBC_STAMP(1);  // A^-1

    // ================= stage 1: contacts (dwbc.h:432-474, dwbc.cpp:433-478, wbd.cpp:108-143) =================
    const unsigned char *fl = io.flags + (size_t)inst * su.n_contacts;
    int act_c[NCC];
    for (int a = 0; a < NCC; a++) act_c[a] = 0;
    int nc = 0, nflag = 0;
    for (int i = 0; i < su.n_contacts; i++) {
        if (fl[i] && nc < NCC) act_c[nc++] = i;
        nflag += fl[i] ? 1 : 0;
    }
    // more simultaneous contacts than this kernel stacks: never solved with a subset -- the instance fails (status 0, zero outputs)
    const bool too_many = nflag > NCC;
    if (too_many) st_contact = 0;
    const int cd = 6 * nc, k = cd > 6 ? cd - 6 : 0;
    const FastDiv fdk(k);  // idx / k by one multiply (idx < 2048)
    DWBC_SYNC();
    for (int a = 0; a < nc; a++) {
        const int ci = act_c[a], link = su.c_link[ci];
        const real_t *R = L + S::Rw + link * 9;
        for (int r = th.tid; r < 12; r += NT) {
            if (r < 9) L[S::Rc + a * 9 + r] = R[r];
            else {
                const int x = r - 9;
                L[S::Pc + a * 3 + x] = L[S::pw + link * 3 + x] + R[x * 3] * (real_t)su.c_point[ci][0] + R[x * 3 + 1] * (real_t)su.c_point[ci][1] +
                                       R[x * 3 + 2] * (real_t)su.c_point[ci][2];
            }
        }
    }
    for (int idx = th.tid; idx < C * N; idx += NT) { L[S::JC + idx] = real_t(0.0); L[S::JbT + idx] = real_t(0.0); }
    DWBC_SYNC();
    for (int a = 0; a < nc; a++)
        point_jacobian<N, NB, NT>(th, L + S::Rw, L + S::pw, L + S::aw, topo, nb, su.c_link[act_c[a]], L + S::Pc + a * 3, L + S::JC, N, 6 * a, 6, 0);
    DWBC_SYNC();
    {
        real_t *Ai = L + S::bufA, *JC = L + S::JC, *Y = L + S::c_Y, *Lam = L + S::Lam, *JbT = L + S::JbT, *AiNc = L + S::bufN;
        mmg<C, N, N, 0>(Y, N, JC, N, Ai, N, cd, N, N);                // Y = J_C A^-1
        DWBC_SYNC();
        mmg<C, C, N, 1>(L + S::c_s2, cd, Y, N, JC, N, cd, N, cd);     // J A^-1 J^T
        if (cd > 0) {
            // Lambda_c (wbd.cpp:115): pivoted Gauss-Jordan, as the reference's .inverse() -- worth its pivot search: the
            // scaled register sweep (spd_inverse_scaled) left two three-contact instances in 6000 at 3e-6 .. 5e-6 Nm
            const real_t cond = gj_inverse_rows<C, NT>(th, L + S::c_s2, cd, cd, Lam, cd, L + S::c_s1);
            if (!(cond > real_t(1e-14))) st_contact = 0;
        }
        mmg<C, N, C, 0>(JbT, N, Lam, cd, Y, N, cd, cd, N);            // Jbar^T = Lambda J A^-1 (wbd.cpp:116)
        DWBC_SYNC();
        DWBC_STAMP(2);  // J_C, Y, Lambda_c, Jbar
        // A^-1 N_c = A^-1 - Y^T Jbar^T   (wbd.cpp:117-118 without materialising N_c)
        static_for<0, (N + 15) / 16>([&](auto ctc) {
            constexpr int c0 = 16 * decltype(ctc)::value;
            wave_gemm<N, (N - c0 < 16 ? N - c0 : 16), C, false>(
                [&](auto, auto, int i, int p) { const real_t v_ = Y[(p < cd ? p : 0) * N + i]; return p < cd ? -v_ : real_t(0.0); },
                [&](auto, int p, int j) { const real_t v_ = JbT[(p < cd ? p : 0) * N + c0 + j]; return p < cd ? v_ : real_t(0.0); },
                [&](int i, int j, real_t d) { AiNc[i * N + c0 + j] = d; },
                [&](int i, int j) { return Ai[i * N + c0 + j]; });
        });
        DWBC_SYNC();
        DWBC_STAMP(3);  // A^-1 N_c
        // ---- W^+ and NwJw.  null(W) is known in closed form: W = S A^-1 N_c S^T vanishes exactly on
        //      { J_C[:,6:]^T lam : J_C[:,:6]^T lam = 0 } (internal wrenches), so V2's span needs no pivoted QR (wbd.cpp:5-53, 120-128)
        real_t *Winv = L + S::bufA;  // A_inv is dead from here on
        real_t *W1 = L + S::bufA, *P = L + S::c_P, *Vb = L + S::c_Vb;  // (W + alpha P and its inverse live where A^-1 was)
        if (k > 0) {
            // basis of internal wrenches: (f_i, m_i) = e_a on contact i >= 1, balanced on contact 0
            const real_t *Pc = L + S::Pc;
            for (int idx = th.tid; idx < M * k; idx += NT) {
                const int r = fdk.div(idx), a = idx - r * k;
                const int ci = 1 + a / 6, e = a % 6;
                real_t f2[3] = {0, 0, 0}, m2[3] = {0, 0, 0};
                if (e < 3) f2[e] = real_t(1.0); else m2[e - 3] = real_t(1.0);
                const real_t d0 = Pc[ci * 3] - Pc[0], d1 = Pc[ci * 3 + 1] - Pc[1], d2 = Pc[ci * 3 + 2] - Pc[2];
                const real_t m1x = -m2[0] - (d1 * f2[2] - d2 * f2[1]);
                const real_t m1y = -m2[1] - (d2 * f2[0] - d0 * f2[2]);
                const real_t m1z = -m2[2] - (d0 * f2[1] - d1 * f2[0]);
                const real_t *J0 = JC, *J1 = JC + 6 * ci * N;
                const int col = 6 + r;
                real_t s = -f2[0] * J0[0 * N + col] - f2[1] * J0[1 * N + col] - f2[2] * J0[2 * N + col];
                s += m1x * J0[3 * N + col] + m1y * J0[4 * N + col] + m1z * J0[5 * N + col];
                s += f2[0] * J1[0 * N + col] + f2[1] * J1[1 * N + col] + f2[2] * J1[2 * N + col];
                s += m2[0] * J1[3 * N + col] + m2[1] * J1[4 * N + col] + m2[2] * J1[5 * N + col];
                Vb[idx] = s;
            }
            DWBC_SYNC();
            // G = Vb^T Vb, G^-1, VG = Vb G^-1;  projector on null(W):  P = VG Vb^T
            real_t *Gm = L + S::c_s2, *cb = L + S::c_s1;
            mmg<S::K, S::K, M, 2>(Gm, k, Vb, k, Vb, k, k, M, k);
            DWBC_SYNC();
            spd_inverse_scaled<S::K>(Gm, k, k, Gm, k, cb);
            mmg<M, S::K, S::K, 0>(L + S::c_VG, k, Vb, k, Gm, k, M, k, k);
            DWBC_SYNC();
            mmg<M, M, S::K, 1>(P, M, L + S::c_VG, k, Vb, k, M, k, M);
            DWBC_SYNC();
            // NwJw = Vb (Jbar[0:k,6:] Vb)^-1   (wbd.cpp:128; invariant to the choice of basis of span(V2^T)).  Jbar1 Vb is a general k x k
            // matrix: pivoted Gauss-Jordan, as the reference's .inverse().  (The SPD-only form of the product kernels, VG JV^T (JV G^-1 JV^T)^-1,
            // squares its condition number: measured 5e-6 Nm on one three-contact instance in 6000, 1e-8 with the direct inverse.)
            mmg<S::K, S::K, M, 0>(Gm, k, JbT + 6, N, Vb, k, k, M, k);  // Jbar[0:k, 6:] Vb
            DWBC_SYNC();
            const real_t cond = gj_inverse_rows<S::K, NT>(th, Gm, k, k, Gm, k, L + S::c_s1);
            if (!(cond > real_t(1e-13))) st_contact = 0;
            mmg<M, S::K, S::K, 0>(L + S::NwJw, k, Vb, k, Gm, k, M, k, k);
            DWBC_SYNC();
        }
        DWBC_STAMP(4);  // Vb, NwJw, projector
        // alpha = trace(W) / M;  W + alpha P is SPD;  W^+ = (W + alpha P)^-1 - P / alpha
        real_t alpha = real_t(0.0);
        for (int i = 0; i < M; i++) alpha += AiNc[(6 + i) * N + 6 + i];
        alpha /= M;
        DWBC_SYNC();
        for (int idx = th.tid; idx < M * M; idx += NT) {
            const int i = idx / M, j = idx - i * M;
            const real_t w = real_t(0.5) * (AiNc[(6 + i) * N + 6 + j] + AiNc[(6 + j) * N + 6 + i]);  // W is symmetric in exact arithmetic
            W1[idx] = w + (k > 0 ? alpha * P[idx] : real_t(0.0));
        }
        DWBC_SYNC();
        {
            if (!spd_inverse_reg<M>(W1, M, W1, M, L + S::colW)) st_contact = 0;  // (pivot column over the idle Gauss-Jordan scratch)
            const real_t ia = alpha != real_t(0.0) ? real_t(1.0) / alpha : real_t(0.0);
            for (int idx = th.tid; idx < M * M; idx += NT) Winv[idx] = W1[idx] - (k > 0 ? P[idx] * ia : real_t(0.0));  // (in place: Winv is W1)
            DWBC_SYNC();
        }
        DWBC_STAMP(5);  // W^+
        // FNl = A_rot (Jbar[:,6:] NwJw)   (cd x k), contact-local frames
        if (k > 0) {
            for (int idx = th.tid; idx < cd * k; idx += NT) {
                const int i = fdk.div(idx), j = idx - i * k;
                real_t s = real_t(0.0);
                _Pragma("unroll 8")
                for (int c = 0; c < M; c++) s += JbT[i * N + 6 + c] * L[S::NwJw + c * k + j];
                L[S::c_s1 + idx] = s;
            }
            DWBC_SYNC();
            for (int idx = th.tid; idx < cd * k; idx += NT) {
                const int i = fdk.div(idx), j = idx - i * k;
                const int a = i / 6, h = (i % 6) / 3, x = i % 3;
                const real_t *R = L + S::Rc + a * 9;
                const real_t *src = L + S::c_s1 + (6 * a + 3 * h) * k + j;
                L[S::FNl + idx] = R[0 * 3 + x] * src[0] + R[1 * 3 + x] * src[k] + R[2 * 3 + x] * src[2 * k];
            }
        }
        DWBC_SYNC();
        // ================= stage 2: gravity compensation (wbd.cpp:186-192) =================
        mv_n<NT>(th, L + S::c_vec, AiNc + 6 * N, N, L + S::G, M, N);  // A^-1[6:,:] N_c G
        DWBC_SYNC();
        mv_n<NT>(th, L + S::tg, Winv, M, L + S::c_vec, M, M);
        mv_n<NT>(th, L + S::PC, JbT, N, L + S::G, cd, N);
        DWBC_SYNC();
    }
    DWBC_STAMP(6);  // FNl, gravity torque, P_C

    // ================= stage 3: task cascade (dwbc.cpp:685-873, 941-1127; wbd.cpp:207-261) =================
    const int nlim = su.has_tau_lim ? 2 * M : 0;
    const int ncone = 10 * nc;
    int st_task = 1, fail_level = -1;
    const io_t *fs_in = io.fstar + (size_t)inst * su.fstar_total;
    {
        real_t *Winv = L + S::bufA, *AiNc = L + S::bufN, *JbT = L + S::JbT;
        for (int lv = 0; lv < su.n_levels && st_task; lv++) {
            const int t = su.t_dof[lv];
            const FastDiv fdt(t), fdt1(t + 1);
            real_t *Jt = L + S::t_Jt, *T1 = L + S::t_T1, *Lt = L + S::t_Lt, *Q = L + S::t_Q, *QW = L + S::t_QW, *Jkt = L + S::t_Jkt, *U = L + S::t_U;
            // --- J_task rows by link mode (dwbc.cpp:709-788)
            DWBC_SYNC();
            int row = 0;
            for (int li = 0; li < su.t_nlinks[lv]; li++) {
                const int mode = su.t_mode[lv][li], link = su.t_link[lv][li];
                real_t pl[3] = {0, 0, 0};
                if (mode == TASK_LINK_6D_COM_FRAME || mode == TASK_LINK_POSITION_COM_FRAME)
                    for (int a = 0; a < 3; a++) pl[a] = body[link * kBodyStride + BF_COM + a];
                else if (mode == TASK_LINK_6D_CUSTOM_FRAME || mode == TASK_LINK_POSITION_CUSTOM_FRAME)
                    for (int a = 0; a < 3; a++) pl[a] = (real_t)su.t_point[lv][li][a];
                const real_t *R = L + S::Rw + link * 9;
                real_t P[3];
                for (int a = 0; a < 3; a++) P[a] = L[S::pw + link * 3 + a] + R[a * 3] * pl[0] + R[a * 3 + 1] * pl[1] + R[a * 3 + 2] * pl[2];
                if (mode <= TASK_LINK_6D_CUSTOM_FRAME) { point_jacobian<N, NB, NT>(th, L + S::Rw, L + S::pw, L + S::aw, topo, nb, link, P, Jt, N, row, 6, 0); row += 6; }
                else if (mode <= TASK_LINK_POSITION_CUSTOM_FRAME) { point_jacobian<N, NB, NT>(th, L + S::Rw, L + S::pw, L + S::aw, topo, nb, link, P, Jt, N, row, 3, 1); row += 3; }
                else { point_jacobian<N, NB, NT>(th, L + S::Rw, L + S::pw, L + S::aw, topo, nb, link, P, Jt, N, row, 3, 2); row += 3; }
            }
            DWBC_SYNC();
            // --- CalculateJKT (wbd.cpp:207-213)
            mmg<T, N, N, 0>(T1, N, Jt, N, AiNc, N, t, N, N);          // J_t A^-1 N_c
            DWBC_SYNC();
            mmg<T, T, N, 1>(L + S::t_s2, t, T1, N, Jt, N, t, N, t);
            gj_inverse_rows<T, NT>(th, L + S::t_s2, t, t, Lt, t, L + S::t_s1);  // Lambda_task (plain inverse, wbd.cpp:210)
            mmg<T, M, T, 0>(Q, M, Lt, t, T1 + 6, N, t, t, M);           // Q = (Lambda J A^-1 N_c)[:,6:]
            DWBC_SYNC();
            mmg<T, M, M, 0>(QW, M, Q, M, Winv, M, t, M, M);            // Q W^+
            DWBC_SYNC();
            mmg<T, T, M, 1>(L + S::t_s2, t, QW, M, Q, M, t, M, t);      // Q W^+ Q^T
            DWBC_SYNC();
            // PinvCODWB (wbd.cpp:5-30, 212): the inverse when the block has full rank, the rank-revealing pseudo-inverse otherwise
            // (threshold 1e-6 on the pivots of a column-pivoted QR, as in the product kernels)
            const real_t piv = gj_inverse_rows<T, NT>(th, L + S::t_s2, t, t, L + S::t_s3, t, L + S::t_s1);
            if (!(piv > kCodCheck)) {
                real_t *cod = L + S::t_cod;
                pinv_cod_small<NT>(th, L + S::t_s2, t, kCodThreshold, L + S::t_s3, cod, cod + T * T, cod + 2 * T * T, cod + 3 * T * T);
                DWBC_SYNC();
            }
            mmg<M, T, T, 2>(Jkt, t, QW, M, L + S::t_s3, t, M, t, t);     // J_kt = W^+ Q^T pinv(.)
            DWBC_SYNC();
            // X = J_kt Lambda ;  Y = (J_t A^-1 N_c)[:,6:]   => Null_i = Null_{i-1} (I - X Y)   (wbd.cpp:257-261)
            real_t *X = (lv < kMaxLevels - 1) ? L + S::Xl + lv * M * T : L + S::t_QW;
            mmg<M, T, T, 0>(X, T, Jkt, t, Lt, t, M, t, t);
            DWBC_SYNC();
            for (int idx = th.tid; idx < M * t; idx += NT) {
                const int i = fdt.div(idx), j = idx - i * t;
                U[i * T + j] = X[i * T + j];
            }
            if (lv < kMaxLevels - 1)
                for (int idx = th.tid; idx < t * M; idx += NT) {
                    const int i = idx / M, j = idx - i * M;
                    L[S::Yl + lv * T * M + idx] = T1[i * N + 6 + j];
                }
            DWBC_SYNC();
            // U = Null_{lv-1} X = (I - X0 Y0)(I - X1 Y1)...(I - X_{lv-1} Y_{lv-1}) X   -- applied right to left
            for (int pl = lv - 1; pl >= 0; pl--) {
                const int tp = su.t_dof[pl];
                const real_t *Xp = L + S::Xl + pl * M * T, *Yp = L + S::Yl + pl * T * M;
                DWBC_SYNC();
                mmg<T, T, M, 0>(L + S::t_s2, t, Yp, M, U, T, tp, M, t);  // Z = Yp U  (tp x t)
                DWBC_SYNC();
                for (int idx = th.tid; idx < M * t; idx += NT) {
                    const int i = fdt.div(idx), j = idx - i * t;
                    real_t s = U[i * T + j];
                    for (int p = 0; p < tp; p++) s -= Xp[i * T + p] * L[S::t_s2 + p * t + j];
                    U[i * T + j] = s;
                }
            }
            DWBC_SYNC();
            // --- QP rows (dwbc.cpp:988-1053)
            if (lv < 2) DWBC_STAMP(7 + 3 * lv);  // level lv: task Jacobian, J_kt, null-space chain
            const io_t *fs = fs_in + su.fstar_off[lv];
            real_t *base = L + S::t_base, *F = L + S::t_F, *fv = L + S::t_fv;
            for (int i = th.tid; i < M; i += NT) {
                real_t s = L[S::tg + i] + L[S::tt + i];
                for (int j = 0; j < t; j++) s += U[i * T + j] * (real_t)fs[j];
                base[i] = s;
            }
            DWBC_SYNC();
            // contact wrench map in the contact-local frames: F (cd x t) = A_rot Jbar[:,6:] U ; fv = A_rot (Jbar[:,6:] base - P_C)
            for (int idx = th.tid; idx < cd * (t + 1); idx += NT) {
                const int i = fdt1.div(idx), j = idx - i * (t + 1);
                real_t s = real_t(0.0);
                if (j < t) { for (int c = 0; c < M; c++) s += JbT[i * N + 6 + c] * U[c * T + j]; }
                else { for (int c = 0; c < M; c++) s += JbT[i * N + 6 + c] * base[c]; s -= L[S::PC + i]; }
                L[S::t_s1 + i * (T + 1) + j] = s;
            }
            DWBC_SYNC();
            for (int idx = th.tid; idx < cd * (t + 1); idx += NT) {
                const int i = fdt1.div(idx), j = idx - i * (t + 1);
                const int a = i / 6, h = (i % 6) / 3, x = i % 3;
                const real_t *R = L + S::Rc + a * 9;
                const real_t *src = L + S::t_s1 + (6 * a + 3 * h) * (T + 1) + j;
                const real_t v = R[0 * 3 + x] * src[0] + R[1 * 3 + x] * src[T + 1] + R[2 * 3 + x] * src[2 * (T + 1)];
                if (j < t) F[i * T + j] = v; else fv[i] = v;
            }
            DWBC_SYNC();
            if (lv < 2) DWBC_STAMP(8 + 3 * lv);  // level lv: QP inputs
            QpResultT<QN> qres;
            qp_rows_and_solve_gc<N, NCC, TG>(su, nlim, ncone, act_c, U, T, t, L + S::NwJw, k, k, kQpScaleGI, F, T, L + S::FNl, k, fv, base, t,
                                         su.qp_max_iter_task, qres, L + S::qp_V, L + S::qp_x, kQpTol, true);
            if (lv < 2) DWBC_STAMP(9 + 3 * lv);  // level lv: QP solved
            if (diag && th.tid == 0) {
                diag[DG_QP_ITER + lv] = qres.iters;
                diag[DG_QP_NACT + lv] = qres.nact;
                for (int a = 0; a < kQpLd; a++) diag[DG_QP_ACT + lv * kQpLd + a] = -1;  // (working sets: product kernels only)
            }
            if (!qres.status) { st_task = 0; fail_level = lv; break; }  // f_star_qp_, contact_qp_ zero; cascade aborts (dwbc.cpp:836,1119)
            const real_t *x = L + S::qp_x;
            // torque_task_ += Null_{i-1} J_kt Lambda (f* + f*_qp) ; torque_contact_ = NwJw contact_qp_ (dwbc.cpp:839-851)
            for (int i = th.tid; i < M; i += NT) {
                real_t s = real_t(0.0);
                for (int j = 0; j < t; j++) s += U[i * T + j] * ((real_t)fs[j] + x[j]);
                L[S::tt + i] += s;
                real_t c = real_t(0.0);
                for (int j = 0; j < k; j++) c += L[S::NwJw + i * k + j] * x[T + j];  // (fixed layout: contact-null variables from position 6 on)
                L[S::tc + i] = c;
            }
            DWBC_SYNC();
        }
    }

    // ================= stage 4: contact redistribution (dwbc.cpp:1372-1568) =================
    int st_redis = 1;
    if (k > 0) {
        real_t *base = L + S::t_base, *fv = L + S::t_fv, *JbT = L + S::JbT;
        DWBC_SYNC();
        for (int i = th.tid; i < M; i += NT) base[i] = L[S::tg + i] + L[S::tt + i] + L[S::tc + i];
        DWBC_SYNC();
        for (int i = th.tid; i < cd; i += NT) {
            real_t s = -L[S::PC + i];
            _Pragma("unroll 8")
            for (int c = 0; c < M; c++) s += JbT[i * N + 6 + c] * base[c];
            L[S::t_s1 + i] = s;
        }
        DWBC_SYNC();
        for (int i = th.tid; i < cd; i += NT) {
            const int a = i / 6, h = (i % 6) / 3, x = i % 3;
            const real_t *R = L + S::Rc + a * 9;
            const real_t *src = L + S::t_s1 + 6 * a + 3 * h;
            fv[i] = R[0 * 3 + x] * src[0] + R[1 * 3 + x] * src[1] + R[2 * 3 + x] * src[2];
        }
        DWBC_SYNC();
        QpResultT<QN> qres;
        // (canon rule 5: the redistribution QP searches with the feasibility tolerance the task QPs were accepted at)
        qp_rows_and_solve_gc<N, NCC, TG>(su, nlim, ncone, act_c, L + S::NwJw, k, k, L + S::NwJw, k, 0, real_t(1.0), L + S::FNl, k, L + S::FNl, k, fv,
                                     base, k, su.qp_max_iter_contact, qres, L + S::qp_V, L + S::qp_x, kQpFeasTol);
        if (diag && th.tid == 0) {
            diag[DG_QP_ITER + kMaxLevels] = qres.iters;
            diag[DG_QP_NACT + kMaxLevels] = qres.nact;
            for (int a = 0; a < kQpLd; a++) diag[DG_QP_ACT + kMaxLevels * kQpLd + a] = -1;
        }
        const real_t *x = L + S::qp_x;
        if (qres.status) {
            for (int i = th.tid; i < M; i += NT) {
                real_t c = real_t(0.0);
                for (int j = 0; j < k; j++) c += L[S::NwJw + i * k + j] * x[j];
                L[S::tc + i] += c;    // torque_contact_ += NwJw c   (dwbc.cpp:1549)
            }
        } else {
            st_redis = 0;
            for (int i = th.tid; i < M; i += NT) L[S::tc + i] = real_t(0.0);  // dwbc.cpp:1553-1559
        }
    } else {
        for (int i = th.tid; i < M; i += NT) L[S::tc + i] = real_t(0.0);  // dwbc.cpp:1562-1567
    }
    DWBC_SYNC();
    DWBC_STAMP(13);  // redistribution QP

    // ================= outputs =================
    io_t *tau = io.tau + (size_t)inst * 3 * M;
    for (int i = th.tid; i < 3 * M; i += NT) tau[i] = too_many ? real_t(0.0) : L[S::tg + i];
    const int wld = io.wrench_ld > 0 ? io.wrench_ld : 12;
    io_t *wr = io.wrench + (size_t)inst * wld;
    for (int i = th.tid; i < wld; i += NT) {
        real_t s = real_t(0.0);
        if (i < cd && !too_many) {
            s = -L[S::PC + i];
            _Pragma("unroll 8")
            for (int c = 0; c < M; c++) s += L[S::JbT + i * N + 6 + c] * (L[S::tg + c] + L[S::tt + c] + L[S::tc + c]);
        }
        wr[i] = s;  // getContactForce(tau_total), wbd.cpp:268-271
    }
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

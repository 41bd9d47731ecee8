// dwbc_cycle.h -- the fused per-instance OSF/HQP control cycle as block-cooperative device code.
//
// One workgroup (NT threads, normally one 64-lane wavefront) owns ONE robot instance for the whole
// cycle; every intermediate (M, M^-1, J_C, Lambda_c, J̄_c^T, A^-1 N_c, W^+, NwJw, per-level J_kt/Lambda_t,
// the QP rows and the active-set state) lives in LDS and never touches HBM.  Only q, flags, f* are read
// and tau / wrench / status written (about 1.4 KB per instance).
//
// The code is written NT-generic with strided loops and explicit barriers so that the SAME source can be
// compiled by g++ with NT = 1 (tests/emu) to check the arithmetic on a machine without a GPU.  That build
// is a test harness only; the product library has no CPU path.
//
// Reference functions restated here (file:line in the reference tree):
//   RobotData::UpdateKinematics          src/dwbc.cpp:279-371   (FK, A_, A_inv_, G_)
//   ContactConstraint::Update            src/contact_constraint.cpp:51-77
//   CalculateContactConstraint           src/wbd.cpp:108-143
//   CalculateGravityCompensation         src/wbd.cpp:186-192
//   RobotData::UpdateTaskSpace           src/dwbc.cpp:685-793
//   CalculateJKT / CalculateTaskNullSpace src/wbd.cpp:207-261
//   RobotData::CalcSingleTaskTorqueWithQP src/dwbc.cpp:941-1127 (QP rows)  + cascade :818-873
//   RobotData::CalcContactRedistribute   src/dwbc.cpp:1372-1568
//   CalculateContactForce                src/wbd.cpp:268-271
//   CQuadraticProgram::SolveQPoases      src/qp_wrapper.cpp:192-380 -> replaced by qp_solve() below
#pragma once
#include <math.h>

#include "dwbc_types.h"

#if defined(DWBC_STAGE_TIMERS) && !defined(DWBC_HOST_EMU)
// diagnostic build only (libdwbc_hip_timed.so): stage stamps in shader cycles, written to diag[DG_TIME..]
#define DWBC_STAMP(i)                                                                   \
    do {                                                                                \
        DWBC_SYNC();                                                                    \
        if (diag && th.tid == 0) diag[DG_TIME + (i)] = (int)(clock64() - t_start_);     \
    } while (0)
#define DWBC_STAMP_INIT() const long long t_start_ = clock64()
#define DWBC_FSTAMP(i)                                                                  \
    do {                                                                                \
        DWBC_SYNC();                                                                    \
        if (dump && th.tid == 0) dump[dl.stamps + (i)] = (real_t)(clock64() - t_start_); \
    } while (0)
#else
#define DWBC_FSTAMP(i) ((void)0)
#define DWBC_STAMP(i) ((void)0)
#define DWBC_STAMP_INIT() ((void)0)
#endif

#ifdef DWBC_HOST_EMU
#define DWBC_DEV
#define DWBC_DEVN
#define DWBC_SYNC() ((void)0)
#else
#define DWBC_DEV __device__ __forceinline__
#ifdef DWBC_OUTLINE_HELPERS
#define DWBC_DEVN __device__ __noinline__
#else
// shared helpers are inlined too: a call forces the live register-resident matrix columns through the callee-saved / spill
// machinery, which costs more than the code growth (138.5 vs 142.2 us per launch at B = 1024; +4 % at B >= 8192)
#define DWBC_DEVN __device__ __forceinline__
#endif
// One wavefront per workgroup (static_assert NT == 64 in the kernels): LDS operations of a wave execute in order, so a
// "barrier" only has to stop the COMPILER from moving LDS accesses across it.  A wavefront-scope fence does that without
// the s_waitcnt lgkmcnt(0) that __syncthreads() costs at every one of the ~100 synchronisation points (135.3 -> 134.0 us).
#ifdef DWBC_BLOCK_BARRIER
#define DWBC_SYNC() __syncthreads()
#else
#define DWBC_SYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
#endif
#endif

namespace dwbc {

constexpr real_t kGrav = real_t(9.81);
#ifndef DWBC_F32_SCALE_GI
#define DWBC_F32_SCALE_GI 1.0e4      /* tuned in emulation (tests/emu, libdwbc_emu_f32.so): tools/f32_accuracy.py */
#define DWBC_F32_SCALE_POLISH 1.0e6
#define DWBC_F32_TOL 2.0e-5
#define DWBC_F32_FEAS 1.0e-3
#endif
// QP parameters: the first value is the fp64 canon (DESIGN.md "QP canon"); the fp32 build uses what single precision can resolve
constexpr real_t kQpScaleGI = kF32 ? real_t(DWBC_F32_SCALE_GI) : real_t(1.0e4);      // c = s * c_hat while the active set is searched
constexpr real_t kQpScalePolish = kF32 ? real_t(DWBC_F32_SCALE_POLISH) : real_t(1.0e9);  // weight of the final (row-sorted, column-pivoted) least-norm solve
constexpr real_t kQpTol = kF32 ? real_t(DWBC_F32_TOL) : real_t(1.0e-9);
constexpr real_t kQpZeroRow = kF32 ? real_t(1.0e-5) : real_t(1.0e-9);     // rows with a smaller norm are the constraint 0 <= hi
constexpr real_t kQpFeasTol = kF32 ? real_t(DWBC_F32_FEAS) : real_t(1.0e-7);     // acceptance of the lexicographic point (slack / |row|)
constexpr int kQpLd = 12;                 // max QP variables (6 task + 6 contact-null)
}  // namespace dwbc
#include "dwbc_qp_wave.h"
namespace dwbc {


// ----------------------------------------------------------------------------------------------
// LDS map (doubles).  N = system dof, M = N-6, C = 12 contact rows, K = 6
// ----------------------------------------------------------------------------------------------
template <int N, int NB>
struct Lds {
    static constexpr int M = N - 6;
    static constexpr int C = 6 * kMaxActiveContacts;
    static constexpr int K = C - 6;
    static constexpr int T = kMaxTaskDof;
    static constexpr int QR = 2 * M + 10 * kMaxActiveContacts;  // max QP rows
    // persistent
    static constexpr int bufA = 0;                    // A -> chol(A) -> A_inv ; later W_inv (M x M)
    static constexpr int bufN = bufA + N * N;         // L^-1 -> A_inv N_c
    static constexpr int Rw = bufN + N * N;           // body->world rotations
    static constexpr int pw = Rw + NB * 9;
    static constexpr int aw = pw + NB * 3;    // world joint axes
    static constexpr int JC = aw + NB * 3;    // C x N
    static constexpr int JbT = JC + C * N;            // J̄_c^T  C x N
    static constexpr int Lam = JbT + C * N;           // C x C
    static constexpr int NwJw = Lam + C * C;          // M x K
    static constexpr int FNl = NwJw + M * K;          // A_rot * J̄[:,6:] * NwJw   C x K
    static constexpr int G = FNl + C * K;
    static constexpr int tg = G + N;                  // torque_grav_
    static constexpr int tt = tg + M;                 // torque_task_
    static constexpr int tc = tt + M;                 // torque_contact_
    static constexpr int PC = tc + M;                 // C
    static constexpr int q = PC + C;                  // N+1
    static constexpr int Rc = q + N + 1;              // contact rotations (active) 2 x 9
    static constexpr int Pc = Rc + kMaxActiveContacts * 9;  // contact points (world) 2 x 3
    static constexpr int Xl = Pc + kMaxActiveContacts * 3;  // per level X = J_kt Lambda (M x T)
    static constexpr int Yl = Xl + (kMaxLevels - 1) * M * T;  // per level Y = (J_t A^-1 N_c)[:,6:] (T x M)
    static constexpr int tmp = Yl + (kMaxLevels - 1) * T * M;  // phase-local scratch
    // --- scratch, kinematics phase
    static constexpr int k_Rl = tmp;                  // local transforms nb x 9
    static constexpr int k_Iw = k_Rl + NB * 9;  // nb x 10
    static constexpr int k_Ic = k_Iw + NB * 10;
    static constexpr int k_S = k_Ic + NB * 10;  // N x 6
    static constexpr int k_F = k_S + N * 6;
    static constexpr int k_end = k_F + N * 6;
    // --- scratch, contact phase
    static constexpr int c_Y = tmp;                   // J_C A^-1 (C x N)
    static constexpr int c_Vb = c_Y + C * N;          // M x K
    static constexpr int c_s1 = c_Vb + M * K;         // C x 2C gauss-jordan scratch
    static constexpr int c_s2 = c_s1 + C * 2 * C;     // C x C
    static constexpr int c_P = c_s2 + C * C;          // M x M projector
    static constexpr int c_W1 = c_P + M * M;          // M x M
    static constexpr int c_vec = c_W1 + M * M;        // N
    static constexpr int c_end = c_vec + N;
    // --- scratch, task / QP phase
    static constexpr int t_Jt = tmp;                  // T x N
    static constexpr int t_T1 = t_Jt + T * N;         // T x N
    static constexpr int t_Lt = t_T1 + T * N;         // T x T
    static constexpr int t_Q = t_Lt + T * T;          // T x M
    static constexpr int t_QW = t_Q + T * M;          // T x M
    static constexpr int t_Jkt = t_QW + T * M;        // M x T
    static constexpr int t_U = t_Jkt + M * T;         // M x T
    static constexpr int t_s1 = t_U + M * T;          // T x 2T
    static constexpr int t_s2 = t_s1 + C * (T + 1);   // T x T   (t_s1 doubles as the C x (T+1) wrench-map scratch)
    static constexpr int t_s3 = t_s2 + T * T;         // T x T
    static constexpr int t_base = t_s3 + T * T;       // M
    static constexpr int t_F = t_base + M;            // C x kQpLd
    static constexpr int t_fv = t_F + C * kQpLd;      // C
    static constexpr int qp_V = t_fv + C;             // householder vectors of the final least-norm solve (12 x 12)
    static constexpr int qp_x = qp_V + kQpLd * kQpLd + 32; // QP solution (uniform copy for the torque updates)
    static constexpr int t_end = qp_x + kQpLd;
    static constexpr int max2(int a, int b) { return a > b ? a : b; }
    static constexpr int total = max2(max2(k_end, c_end), t_end);
    static constexpr int total_bytes = total * (int)sizeof(real_t) + 64 * 4 + 64;  // + int scratch
};

// ----------------------------------------------------------------------------------------------
// tiny helpers
// ----------------------------------------------------------------------------------------------
struct Thr {
    int tid;
};

// idx / n for 0 <= idx < 2048, 1 <= n <= 64 with one multiply (a runtime integer division costs ~40 VALU instructions)
struct FastDiv {
    int n;
    unsigned inv;
    DWBC_DEV explicit FastDiv(int n_) : n(n_), inv((1u << 20) / (unsigned)(n_ > 0 ? n_ : 1) + 1u) {}
    DWBC_DEV int div(int idx) const { return (int)(((unsigned)idx * inv) >> 20); }
};

template <int NT>
DWBC_DEV void mm_nn(Thr th, real_t *Cm, int ldc, const real_t *A, int lda, const real_t *B, int ldb, int m, int k, int n) {
    const FastDiv fd(n);
    for (int idx = th.tid; idx < m * n; idx += NT) {
        int i = fd.div(idx), j = idx - i * n;
        real_t s = real_t(0.0);
        _Pragma("unroll 8")
        for (int p = 0; p < k; p++) s += A[i * lda + p] * B[p * ldb + j];
        Cm[i * ldc + j] = s;
    }
}
// C = A * B^T   (A m x k, B n x k)
template <int NT>
DWBC_DEV void mm_nt(Thr th, real_t *Cm, int ldc, const real_t *A, int lda, const real_t *B, int ldb, int m, int k, int n) {
    const FastDiv fd(n);
    for (int idx = th.tid; idx < m * n; idx += NT) {
        int i = fd.div(idx), j = idx - i * n;
        real_t s = real_t(0.0);
        _Pragma("unroll 8")
        for (int p = 0; p < k; p++) s += A[i * lda + p] * B[j * ldb + p];
        Cm[i * ldc + j] = s;
    }
}
// C = A^T * B   (A k x m, B k x n)
template <int NT>
DWBC_DEV void mm_tn(Thr th, real_t *Cm, int ldc, const real_t *A, int lda, const real_t *B, int ldb, int m, int k, int n) {
    const FastDiv fd(n);
    for (int idx = th.tid; idx < m * n; idx += NT) {
        int i = fd.div(idx), j = idx - i * n;
        real_t s = real_t(0.0);
        _Pragma("unroll 8")
        for (int p = 0; p < k; p++) s += A[p * lda + i] * B[p * ldb + j];
        Cm[i * ldc + j] = s;
    }
}
template <int NT>
DWBC_DEV void mv_n(Thr th, real_t *y, const real_t *A, int lda, const real_t *x, int m, int n) {
    for (int i = th.tid; i < m; i += NT) {
        real_t s = real_t(0.0);
        _Pragma("unroll 8")
        for (int j = 0; j < n; j++) s += A[i * lda + j] * x[j];
        y[i] = s;
    }
}

// In-place inverse of a small general matrix by Gauss-Jordan with partial pivoting (stands in for Eigen's
// MatrixXd::inverse(), reference src/wbd.cpp:115,128,210).  W is an n x 2n scratch.  Returns min|pivot|/max|pivot|.
template <int NT>
DWBC_DEVN real_t gj_inverse(Thr th, const real_t *A, int lda, int n, real_t *Ai, int ldi, real_t *W) {
    const int w = 2 * n;
    DWBC_SYNC();
    for (int idx = th.tid; idx < n * w; idx += NT) {
        int i = idx / w, j = idx - i * w;
        W[idx] = j < n ? A[i * lda + j] : (j - n == i ? real_t(1.0) : real_t(0.0));
    }
    real_t pmin = kF32 ? real_t(1e30) : real_t(1e300), pmax = real_t(0.0);
    for (int c = 0; c < n; c++) {
        DWBC_SYNC();
        int p = c;
        real_t best = fabs(W[c * w + c]);
        for (int i = c + 1; i < n; i++) {
            real_t v = fabs(W[i * w + c]);
            if (v > best) { best = v; p = i; }
        }
        pmin = best < pmin ? best : pmin;
        pmax = best > pmax ? best : pmax;
        DWBC_SYNC();
        if (p != c)
            for (int j = th.tid; j < w; j += NT) { real_t t = W[c * w + j]; W[c * w + j] = W[p * w + j]; W[p * w + j] = t; }
        DWBC_SYNC();
        real_t piv = W[c * w + c];
        real_t inv = piv != real_t(0.0) ? real_t(1.0) / piv : real_t(0.0);
        DWBC_SYNC();
        for (int j = th.tid; j < w; j += NT) W[c * w + j] *= inv;
        DWBC_SYNC();
        // eliminate: element (i,j) -= W[i][c] * W[c][j]; column c itself must be read before it is overwritten
        for (int idx = th.tid; idx < n * w; idx += NT) {
            int i = idx / w, j = idx - i * w;
            if (i == c || j == c) continue;
            W[idx] -= W[i * w + c] * W[c * w + j];
        }
        DWBC_SYNC();
        for (int i = th.tid; i < n; i += NT)
            if (i != c) W[i * w + c] = real_t(0.0);
    }
    DWBC_SYNC();
    for (int idx = th.tid; idx < n * n; idx += NT) {
        int i = idx / n, j = idx - i * n;
        Ai[i * ldi + j] = W[i * w + n + j];
    }
    DWBC_SYNC();
    return pmax > real_t(0.0) ? pmin / pmax : real_t(0.0);
}

// SPD inverse of an n x n matrix held in LDS:  S (destroyed, ld n) -> Out (ld n); Tmp is n x n scratch.
// Right-looking Cholesky, column-parallel forward substitution, then L^-T L^-1 -- the arithmetic of Eigen's
// llt().solve(Identity) (reference src/dwbc.cpp:307).  Returns 0 when a pivot is not positive.
template <int NT>
DWBC_DEV int spd_inverse(Thr th, real_t *S, int n, real_t *Tmp, real_t *Out) {
    int ok = 1;
    for (int k = 0; k < n; k++) {
        DWBC_SYNC();
        real_t d = S[k * n + k];
        if (!(d > real_t(0.0))) { ok = 0; d = real_t(1.0); }
        d = sqrt(d);
        real_t rd = real_t(1.0) / d;
        DWBC_SYNC();
        for (int i = k + th.tid; i < n; i += NT) {
            real_t v = (i == k) ? d : S[i * n + k] * rd;
            S[i * n + k] = v;
            S[k * n + i] = v;  // keep row k too: the update below reads L[j][k] as S[k][j]
        }
        DWBC_SYNC();
        const int r = n - k - 1;
        for (int idx = th.tid; idx < r * r; idx += NT) {
            int i = k + 1 + idx / r, j = k + 1 + idx % r;
            S[i * n + j] -= S[i * n + k] * S[k * n + j];
        }
    }
    DWBC_SYNC();
    // Tmp = L^-1 (lower), column c owned by one thread
    for (int idx = th.tid; idx < n * n; idx += NT) Tmp[idx] = real_t(0.0);
    DWBC_SYNC();
    for (int c = th.tid; c < n; c += NT) {
        for (int i = c; i < n; i++) {
            real_t s = (i == c) ? real_t(1.0) : real_t(0.0);
            for (int k = c; k < i; k++) s -= S[i * n + k] * Tmp[k * n + c];
            Tmp[i * n + c] = s / S[i * n + i];
        }
    }
    DWBC_SYNC();
    for (int idx = th.tid; idx < n * n; idx += NT) {
        int i = idx / n, j = idx - i * n;
        int k0 = i > j ? i : j;
        real_t s = real_t(0.0);
        for (int k = k0; k < n; k++) s += Tmp[k * n + i] * Tmp[k * n + j];
        Out[idx] = s;
    }
    DWBC_SYNC();
    return ok;
}

// SPD inverse with one matrix column per lane held in registers (symmetric Gauss-Jordan "sweep", no pivoting needed for
// SPD).  Per pivot k: lane k publishes its column to LDS, every lane j reads c_j = S[j][k] and the uniform column c_i,
// and applies  S[i][j] -= (c_i - delta_ik) * h_j  with h_j = c_j / d (h_k = 1 - 1/d): one FMA per element, the row-k
// and column-k special cases of the sweep fall out of the modified multiplier.  The diagonal is tracked separately
// (dg) so that no register is indexed dynamically.  Same arithmetic role as Eigen's llt().solve(I) (reference
// src/dwbc.cpp:307).  Sin: NN x NN row-major LDS (ld), Out: NN x NN (ldo), colbuf: NN doubles of LDS.
template <int NN>
DWBC_DEVN int spd_inverse_wave(const real_t *Sin, int ld, real_t *Out, int ldo, real_t *colbuf) {
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
    int ok = 1;
    for (int k = 0; k < NN; k++) {
        real_t d = BCAST(dg, k);
        if (!(d > real_t(0.0))) { ok = 0; d = real_t(1.0); }
        const real_t rp = real_t(1.0) / d;
        DWBC_SYNC();
        LANES {
            if (lane == k) {
#pragma unroll
                for (int i = 0; i < NN; i++) colbuf[i] = LV(s)[i];
            }
        }
        DWBC_SYNC();
        LANES {
            if (lane == k) colbuf[k] = d - real_t(1.0);
        }
        DWBC_SYNC();
        LANES {
            const real_t cj = colbuf[lane < NN ? lane : 0];
            const real_t h = (lane == k) ? (real_t(1.0) - rp) : cj * rp;
#pragma unroll
            for (int i = 0; i < NN; i++) LV(s)[i] -= colbuf[i] * h;
            LV(dg) = (lane == k) ? -rp : LV(dg) - cj * h;
        }
    }
    DWBC_SYNC();
    LANES {
        if (lane < NN) {
#pragma unroll
            for (int i = 0; i < NN; i++) Out[i * ldo + lane] = (i == lane) ? -LV(dg) : -LV(s)[i];
        }
    }
    DWBC_SYNC();
    return ok;
}

DWBC_DEV real_t cone_row(int r, real_t lx, real_t ly, real_t mu, real_t muz, const real_t *w) {
    // rows of [GetZMPConstMatrix; GetForceConstMatrix] (reference src/wbd.cpp:59-97) applied to a local wrench
    switch (r) {
        case 0: return -lx * w[2] - w[4];
        case 1: return -lx * w[2] + w[4];
        case 2: return -ly * w[2] - w[3];
        case 3: return -ly * w[2] + w[3];
        case 4: return w[0] - mu * w[2];
        case 5: return -w[0] - mu * w[2];
        case 6: return w[1] - mu * w[2];
        case 7: return -w[1] - mu * w[2];
        case 8: return w[5] - muz * w[2];
        default: return -w[5] - muz * w[2];
    }
}

// ----------------------------------------------------------------------------------------------
// point Jacobian (6 x N, rows [linear; angular]) of world point P fixed on body `link`
// (CalcPointJacobian6D + row swap: reference src/link.cpp:98-119, src/contact_constraint.cpp:59-61)
// ----------------------------------------------------------------------------------------------
template <int N, int NB, int NT>
DWBC_DEVN void point_jacobian(Thr th, const real_t *Rw, const real_t *pw, const real_t *aw, const int *topo, int nb, int link,
                             const real_t *P, real_t *J, int ld, int row0, int nrows, int rsel, int cs = 1) {
    // element (row, col) is stored at J[row * ld + col * cs]  (ld = N, cs = 1: row-major;  ld = 1, cs = rows: transposed)
    // rsel: 0 -> rows 0..5, 1 -> linear rows only (0..2), 2 -> angular rows only (3..5)
    for (int j = th.tid; j < N; j += NT) {
        real_t lin[3] = {0, 0, 0}, ang[3] = {0, 0, 0};
        if (j < 3) {
            lin[j] = real_t(1.0);
        } else {
            real_t w[3], o[3];
            bool on = true;
            if (j < 6) {
                for (int a = 0; a < 3; a++) { w[a] = Rw[a * 3 + (j - 3)]; o[a] = pw[a]; }
            } else {
                const int b = j - 5;
                on = (b <= link) && (link < b + topo[2 * nb + b]);
                for (int a = 0; a < 3; a++) { w[a] = aw[b * 3 + a]; o[a] = pw[b * 3 + a]; }
            }
            if (on) {
                const real_t d0 = P[0] - o[0], d1 = P[1] - o[1], d2 = P[2] - o[2];
                lin[0] = w[1] * d2 - w[2] * d1;
                lin[1] = w[2] * d0 - w[0] * d2;
                lin[2] = w[0] * d1 - w[1] * d0;
                ang[0] = w[0]; ang[1] = w[1]; ang[2] = w[2];
            }
        }
        if (rsel == 0) {
            for (int a = 0; a < 3; a++) { J[(row0 + a) * ld + j * cs] = lin[a]; J[(row0 + 3 + a) * ld + j * cs] = ang[a]; }
        } else if (rsel == 1) {
            for (int a = 0; a < 3; a++) J[(row0 + a) * ld + j * cs] = lin[a];
        } else {
            for (int a = 0; a < 3; a++) J[(row0 + a) * ld + j * cs] = ang[a];
        }
    }
    (void)nrows;
}

// ----------------------------------------------------------------------------------------------
// Centroidal quantities of RobotData::UpdateKinematics (reference src/dwbc.cpp:318-352), written to the dump record
// only: com_pos, CMM_ = cm_rot6 * A[0:6,:], COM inertia, jac_com_ = SI_body^-1 CMM_.  None of them feeds the OSF
// torque path (G_ is taken from A directly), so they are produced on request (dwbc_batch_enable_dump).
//   A: staged mass matrix (row stride lda), R0: pelvis rotation, q0: base position, mt: total mass
// ----------------------------------------------------------------------------------------------
template <int N, int NT>
DWBC_DEV void dump_centroidal(Thr th, const real_t *A, int lda, const real_t *R0, const real_t *q0, real_t mt, real_t *dump,
                              const DumpLayout &dl) {
    real_t c[3];
    {
        // skm = R0 * A[3:6,0:3] / mt ;  com_from_pelv = (skm(2,1), skm(0,2), skm(1,0))
        const int ra[3] = {2, 0, 1}, cb[3] = {1, 2, 0};
        for (int e = 0; e < 3; e++) {
            real_t acc = real_t(0.0);
            for (int b = 0; b < 3; b++) acc += R0[ra[e] * 3 + b] * A[(3 + b) * lda + cb[e]];
            c[e] = acc / mt;
        }
    }
    // COM inertia about the COM, world frame
    real_t I[9];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
            real_t acc = real_t(0.0);
            for (int u = 0; u < 3; u++)
                for (int v = 0; v < 3; v++) acc += R0[a * 3 + u] * A[(3 + u) * lda + 3 + v] * R0[b * 3 + v];
            const real_t cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
            I[a * 3 + b] = acc - mt * ((a == b ? cc : real_t(0.0)) - c[a] * c[b]);  // skew(c) skew(c)^T = |c|^2 I - c c^T
        }
    const real_t det = I[0] * (I[4] * I[8] - I[5] * I[7]) - I[1] * (I[3] * I[8] - I[5] * I[6]) + I[2] * (I[3] * I[7] - I[4] * I[6]);
    const real_t id = real_t(1.0) / det;
    const real_t Ii[9] = {(I[4] * I[8] - I[5] * I[7]) * id, (I[2] * I[7] - I[1] * I[8]) * id, (I[1] * I[5] - I[2] * I[4]) * id,
                          (I[5] * I[6] - I[3] * I[8]) * id, (I[0] * I[8] - I[2] * I[6]) * id, (I[2] * I[3] - I[0] * I[5]) * id,
                          (I[3] * I[7] - I[4] * I[6]) * id, (I[1] * I[6] - I[0] * I[7]) * id, (I[0] * I[4] - I[1] * I[3]) * id};
    if (th.tid == 0) {
        for (int a = 0; a < 3; a++) dump[dl.com + a] = c[a] + q0[a];
        for (int a = 0; a < 9; a++) dump[dl.com_inertia + a] = I[a];
    }
    // skew(c)^T rows: [0 c2 -c1; -c2 0 c0; c1 -c0 0]
    const real_t St[9] = {real_t(0.0), c[2], -c[1], -c[2], real_t(0.0), c[0], c[1], -c[0], real_t(0.0)};
    for (int j = th.tid; j < N; j += NT) {
        real_t h[3];
        for (int a = 0; a < 3; a++) {
            dump[dl.CMM + a * N + j] = A[a * lda + j];
            dump[dl.J_com + a * N + j] = A[a * lda + j] / mt;
            real_t acc = real_t(0.0);
            for (int b = 0; b < 3; b++) acc += St[a * 3 + b] * A[b * lda + j] + R0[a * 3 + b] * A[(3 + b) * lda + j];
            h[a] = acc;
            dump[dl.CMM + (3 + a) * N + j] = acc;
        }
        for (int a = 0; a < 3; a++) dump[dl.J_com + (3 + a) * N + j] = Ii[a * 3] * h[0] + Ii[a * 3 + 1] * h[1] + Ii[a * 3 + 2] * h[2];
    }
}

// RobotData::getZMP(getContactForce(tau_total)) (reference src/dwbc.cpp:898-939) + cc_[i].xc_pos / rotm / zmp_pos of the
// active contacts, into the dump record.  The reference indexes the packed wrench by REGISTRATION index (i * 6), which is only
// consistent when the active contacts are the first registered ones; the active order is used here.
DWBC_DEV void dump_contacts_zmp(Thr th, const real_t *Pc, const real_t *Rc, const io_t *wr, int nc, real_t *dump, const DumpLayout &dl) {
    if (th.tid != 0) return;
    real_t tot = real_t(0.0), z[3] = {0, 0, 0};
    for (int a = 0; a < nc; a++) tot += (real_t)wr[6 * a + 2];
    for (int a = 0; a < kMaxActiveContacts; a++) {
        real_t zp[3] = {0, 0, 0};
        if (a < nc) {
            const real_t fz = (real_t)wr[6 * a + 2];
            zp[0] = Pc[a * 3]; zp[1] = Pc[a * 3 + 1]; zp[2] = Pc[a * 3 + 2];
            if (!(fz > -real_t(1.0e-3))) { zp[0] += -(real_t)wr[6 * a + 4] / fz; zp[1] += (real_t)wr[6 * a + 3] / fz; }
            for (int x = 0; x < 3; x++) z[x] += zp[x] * fz / tot;
        }
        for (int x = 0; x < 3; x++) {
            dump[dl.zmp + 3 + a * 3 + x] = zp[x];
            dump[dl.contact_pos + a * 3 + x] = a < nc ? Pc[a * 3 + x] : real_t(0.0);
        }
        for (int x = 0; x < 9; x++) dump[dl.contact_rot + a * 9 + x] = a < nc ? Rc[a * 9 + x] : real_t(0.0);
    }
    for (int x = 0; x < 3; x++) dump[dl.zmp + x] = z[x];
}

// Jacobian of the synthetic "COM" link, jac_ = jac_com_ = SI_body^-1 CMM_ (reference src/dwbc.cpp:318-353), 6 x N row-major
// [linear; angular], followed by com_pos (3).  Same arithmetic as dump_centroidal, kept in LDS for COM task levels.
template <int N, int NT>
DWBC_DEV void com_jacobian(Thr th, const real_t *A, int lda, const real_t *R0, const real_t *q0, real_t *Jcm) {
    const real_t mt = A[0];
    real_t c[3];
    {
        const int ra[3] = {2, 0, 1}, cb[3] = {1, 2, 0};
        for (int e = 0; e < 3; e++) {
            real_t acc = real_t(0.0);
            for (int b = 0; b < 3; b++) acc += R0[ra[e] * 3 + b] * A[(3 + b) * lda + cb[e]];
            c[e] = acc / mt;
        }
    }
    real_t I[9];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
            real_t acc = real_t(0.0);
            for (int u = 0; u < 3; u++)
                for (int v = 0; v < 3; v++) acc += R0[a * 3 + u] * A[(3 + u) * lda + 3 + v] * R0[b * 3 + v];
            const real_t cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
            I[a * 3 + b] = acc - mt * ((a == b ? cc : real_t(0.0)) - c[a] * c[b]);
        }
    const real_t det = I[0] * (I[4] * I[8] - I[5] * I[7]) - I[1] * (I[3] * I[8] - I[5] * I[6]) + I[2] * (I[3] * I[7] - I[4] * I[6]);
    const real_t id = real_t(1.0) / det;
    const real_t Ii[9] = {(I[4] * I[8] - I[5] * I[7]) * id, (I[2] * I[7] - I[1] * I[8]) * id, (I[1] * I[5] - I[2] * I[4]) * id,
                          (I[5] * I[6] - I[3] * I[8]) * id, (I[0] * I[8] - I[2] * I[6]) * id, (I[2] * I[3] - I[0] * I[5]) * id,
                          (I[3] * I[7] - I[4] * I[6]) * id, (I[1] * I[6] - I[0] * I[7]) * id, (I[0] * I[4] - I[1] * I[3]) * id};
    const real_t St[9] = {real_t(0.0), c[2], -c[1], -c[2], real_t(0.0), c[0], c[1], -c[0], real_t(0.0)};
    // read every A entry this thread needs before any store: Jcm may overlay dead rows of the staged A
    for (int j = th.tid; j < N; j += NT) {
        real_t al[3], ab[3], h[3];
        for (int a = 0; a < 3; a++) { al[a] = A[a * lda + j]; ab[a] = A[(3 + a) * lda + j]; }
        for (int a = 0; a < 3; a++) {
            real_t acc = real_t(0.0);
            for (int b = 0; b < 3; b++) acc += St[a * 3 + b] * al[b] + R0[a * 3 + b] * ab[b];
            h[a] = acc;
        }
        for (int a = 0; a < 3; a++) {
            Jcm[a * N + j] = al[a] / mt;
            Jcm[(3 + a) * N + j] = Ii[a * 3] * h[0] + Ii[a * 3 + 1] * h[1] + Ii[a * 3 + 2] * h[2];
        }
    }
    if (th.tid == 0)
        for (int a = 0; a < 3; a++) Jcm[6 * N + a] = c[a] + q0[a];
}

// rows of a COM task level into the transposed task Jacobian Jtt (N x T): rsel 0 -> 6 rows, 1 -> linear, 2 -> angular
template <int N, int NT>
DWBC_DEV void com_task_rows(Thr th, const real_t *Jcm, real_t *Jtt, int row0, int rsel, int T) {
    for (int j = th.tid; j < N; j += NT) {
        if (rsel == 0) { for (int a = 0; a < 6; a++) Jtt[j * T + row0 + a] = Jcm[a * N + j]; }
        else if (rsel == 1) { for (int a = 0; a < 3; a++) Jtt[j * T + row0 + a] = Jcm[a * N + j]; }
        else { for (int a = 0; a < 3; a++) Jtt[j * T + row0 + a] = Jcm[(3 + a) * N + j]; }
    }
}

// ----------------------------------------------------------------------------------------------
// QP rows into lanes + solve.  Lane r < M owns torque-limit row r (two sided), lane M + rr owns cone row rr.
//   torque rows:  [P1 | s2 P2][r,:] x  in  [-(lim + base), lim - base]        (reference src/dwbc.cpp:1001-1016)
//   cone rows:    -cone(W1 | s2 W2)[rr,:] x <= cone(fv)[rr]                     (reference src/dwbc.cpp:1041-1053)
// W1/W2/fv are the contact wrench maps already rotated into the contact frames (A_rot applied).
// ----------------------------------------------------------------------------------------------
template <int N, int NB, int WS = 1>
DWBC_DEV void qp_rows_and_solve(const Setup &su, real_t *L, int nlim, int ncone, int ci0, int ci1, const real_t *P1, int ld1,
                                int t1, const real_t *P2, int ld2, int t2, real_t s2, const real_t *W1, int ldw1,
                                const real_t *W2, int ldw2, const real_t *fv, const real_t *base, int tvars, int max_iter,
                                QpResult &res, real_t *Vlds, real_t *xlds) {
    using S = Lds<N, NB>;
    constexpr int M = S::M;
    DWBC_LANE_DECL;
    QpRows R;
    const int nv = t1 + t2;
    LANES {
#pragma unroll
        for (int j = 0; j < kQpN; j++) LV(R.g)[j] = real_t(0.0);
        LV(R.hi) = DWBC_QP_INF;
        LV(R.lo) = DWBC_QP_INF;
        LV(R.id_hi) = -1;
        LV(R.id_lo) = -1;
        if (lane < M) {
            if (nlim) {
#pragma unroll
                for (int j = 0; j < kQpN; j++) {
                    real_t v = real_t(0.0);
                    if (j < t1) v = P1[lane * ld1 + j];
                    else if (j < nv) v = P2[lane * ld2 + (j - t1)] * s2;
                    LV(R.g)[j] = v;
                }
                LV(R.hi) = su.tau_lim[lane] - base[lane];
                LV(R.lo) = su.tau_lim[lane] + base[lane];
                LV(R.id_hi) = lane;
                LV(R.id_lo) = M + lane;
            }
        } else if (lane - M < ncone) {
            // cone row r10 of contact a acts on the local wrench w as  c2 * w[2] + sg * w[oi]   (reference src/wbd.cpp:59-97)
            const int rr = lane - M, a = rr / 10, r10 = rr - 10 * a;
            const int ci = a ? ci1 : ci0;
            const int pr = r10 >> 1;
            const real_t c2 = -(pr == 0 ? su.c_lx[ci] : pr == 1 ? su.c_ly[ci] : pr == 4 ? su.c_muz[ci] : su.c_mu[ci]);
            const int oi = pr == 0 ? 4 : pr == 1 ? 3 : pr == 2 ? 0 : pr == 3 ? 1 : 5;
            const real_t sg = (pr < 2) ? ((r10 & 1) ? real_t(1.0) : -real_t(1.0)) : ((r10 & 1) ? -real_t(1.0) : real_t(1.0));
            const int row2 = 6 * a + 2, rowo = 6 * a + oi;
#pragma unroll
            for (int j = 0; j < kQpN; j++) {
                real_t v = real_t(0.0);
                if (j < t1) v = c2 * W1[row2 * ldw1 + j] + sg * W1[rowo * ldw1 + j];
                else if (j < nv) v = (c2 * W2[row2 * ldw2 + (j - t1)] + sg * W2[rowo * ldw2 + (j - t1)]) * s2;
                LV(R.g)[j] = -v;
            }
            LV(R.hi) = c2 * fv[row2] + sg * fv[rowo];
            LV(R.id_hi) = nlim + rr;
        }
    }
    // the solver is instantiated for 12, 9 and 6 variables (6 + 6, 3 + 6 and the 6 contact-null variables of the redistribution)
    if (nv <= 6) qp_solve_wave<WS, 6>(R, nv, tvars, max_iter, res, Vlds);
    else if (nv <= 9) qp_solve_wave<WS, 9>(R, nv, tvars, max_iter, res, Vlds);
    else qp_solve_wave<WS, 12>(R, nv, tvars, max_iter, res, Vlds);
    LANES {
        if (lane < kQpN) xlds[lane] = pick12(res.x, lane);
    }
    DWBC_SYNC();
}

// ----------------------------------------------------------------------------------------------
// the fused cycle for one instance
// ----------------------------------------------------------------------------------------------
template <int N, int NB, int NT>
DWBC_DEV void cycle_instance(Thr th, const Setup &su, const BatchIO &io, int inst, real_t *L, int *iL) {
    using S = Lds<N, NB>;
    constexpr int M = S::M, C = S::C, T = S::T;
    const int nb = su.nb;
    const real_t *body = io.body;
    const int *topo = io.topo;  // parent[nb] depth[nb] subtree[nb]
    const io_t *qin = io.q + (size_t)inst * (N + 1);
    const DumpLayout dl = DumpLayout::make(N);
    real_t *dump = io.dump ? io.dump + (size_t)inst * dl.total : nullptr;
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
        for (int idx = th.tid; idx < nb * 10; idx += NT) {
            const int i = idx / 10, c = idx - i * 10;
            const int e = i + topo[2 * nb + i];
            real_t s = real_t(0.0);
            for (int j = i; j < e; j++) s += Iw[j * 10 + c];
            Icm[idx] = s;
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
        if (dump) {
            for (int idx = th.tid; idx < N * N; idx += NT) dump[dl.A + idx] = A[idx];
            dump_centroidal<N, NT>(th, A, N, Rw, L + S::q, A[0], dump, dl);  // A(0,0) = total mass
            for (int idx = th.tid; idx < nb * 9; idx += NT) dump[dl.link_R + idx] = Rw[idx];
            for (int idx = th.tid; idx < nb * 3; idx += NT) dump[dl.link_p + idx] = pw[idx];
        }
    }
    DWBC_STAMP(0);  // kinematics + CRBA done
    int st_contact = 1;
    // A_inv = llt(A).solve(I)  (dwbc.cpp:307): bufA (A) -> bufA (A_inv), bufN scratch
    {
        const int ok = spd_inverse_wave<N>(L + S::bufA, N, L + S::bufA, N, L + S::tmp);  // in place (columns live in registers)
        if (!ok) st_contact = 0;
        if (dump) {
            for (int idx = th.tid; idx < N * N; idx += NT) dump[dl.A_inv + idx] = L[S::bufA + idx];
            for (int j = th.tid; j < N; j += NT) dump[dl.G + j] = L[S::G + j];
        }
    }

    DWBC_STAMP(1);  // A_inv done
    // ================= stage 1: contacts (dwbc.h:432-474, dwbc.cpp:433-478, wbd.cpp:108-143) =================
    const unsigned char *fl = io.flags + (size_t)inst * su.n_contacts;
    int act_c[kMaxActiveContacts] = {0, 0};
    int nc = 0;
    for (int i = 0; i < su.n_contacts; i++)
        if (fl[i] && nc < kMaxActiveContacts) act_c[nc++] = i;
    const int cd = 6 * nc, k = cd > 6 ? cd - 6 : 0;
    DWBC_SYNC();
    for (int a = 0; a < nc; a++) {
        const int ci = act_c[a], link = su.c_link[ci];
        const real_t *R = L + S::Rw + link * 9;
        for (int r = th.tid; r < 12; r += NT) {
            if (r < 9) L[S::Rc + a * 9 + r] = R[r];
            else {
                const int x = r - 9;
                L[S::Pc + a * 3 + x] = L[S::pw + link * 3 + x] + R[x * 3] * su.c_point[ci][0] + R[x * 3 + 1] * su.c_point[ci][1] + R[x * 3 + 2] * su.c_point[ci][2];
            }
        }
    }
    DWBC_SYNC();
    for (int a = 0; a < nc; a++)
        point_jacobian<N, NB, NT>(th, L + S::Rw, L + S::pw, L + S::aw, topo, nb, su.c_link[act_c[a]], L + S::Pc + a * 3, L + S::JC, N, 6 * a, 6, 0);
    DWBC_SYNC();
    {
        real_t *Ai = L + S::bufA, *JC = L + S::JC, *Y = L + S::c_Y, *Lam = L + S::Lam, *JbT = L + S::JbT, *AiNc = L + S::bufN;
        mm_nn<NT>(th, Y, N, JC, N, Ai, N, cd, N, N);                 // Y = J_C A^-1
        DWBC_SYNC();
        mm_nt<NT>(th, L + S::c_s2, cd, Y, N, JC, N, cd, N, cd);      // J A^-1 J^T
        if (cd > 0) {
            real_t cond = gj_inverse<NT>(th, L + S::c_s2, cd, cd, Lam, cd, L + S::c_s1);  // Lambda_c (wbd.cpp:115)
            if (!(cond > real_t(1e-14))) st_contact = 0;
        }
        mm_nn<NT>(th, JbT, N, Lam, cd, Y, N, cd, cd, N);             // J̄^T = Lambda J A^-1 (wbd.cpp:116)
        DWBC_SYNC();
        // A^-1 N_c = A^-1 - Y^T J̄^T   (wbd.cpp:117-118 without materialising N_c)
        for (int idx = th.tid; idx < N * N; idx += NT) {
            const int i = idx / N, j = idx - i * N;
            real_t s = Ai[idx];
            _Pragma("unroll 8")
            for (int p = 0; p < cd; p++) s -= Y[p * N + i] * JbT[p * N + j];
            AiNc[idx] = s;
        }
        DWBC_SYNC();
        if (dump) {
            for (int idx = th.tid; idx < cd * N; idx += NT) { dump[dl.J_C + idx] = JC[idx]; dump[dl.J_C_INV_T + idx] = JbT[idx]; }
            for (int idx = th.tid; idx < cd * cd; idx += NT) dump[dl.Lambda_c + idx] = Lam[idx];
            for (int idx = th.tid; idx < N * N; idx += NT) dump[dl.A_inv_N_C + idx] = AiNc[idx];
        }
        DWBC_STAMP(2);  // J_C, Lambda_c, J̄, A^-1 N_c done
        // ---- W^+ and NwJw.  null(W) is known in closed form: W = S A^-1 N_c S^T vanishes exactly on
        //      { J_C[:,6:]^T lam : J_C[:,:6]^T lam = 0 } (internal wrenches), so V2's span needs no pivoted QR.
        real_t *Winv = L + S::bufA;  // A_inv is dead from here on
        real_t *W1 = L + S::c_W1, *P = L + S::c_P, *Vb = L + S::c_Vb;
        if (k > 0) {
            // basis of internal wrenches: (f_i, m_i) = e_a on contact i>=1, balanced on contact 0
            const real_t *Pc = L + S::Pc;
            for (int idx = th.tid; idx < M * k; idx += NT) {
                const int r = idx / k, a = idx - r * k;
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
            // NwJw = Vb (J̄[0:k,6:] Vb)^-1   (wbd.cpp:128; invariant to the choice of basis of span(V2^T))
            for (int idx = th.tid; idx < k * k; idx += NT) {
                const int i = idx / k, j = idx - i * k;
                real_t s = real_t(0.0);
                _Pragma("unroll 8")
                for (int c = 0; c < M; c++) s += JbT[i * N + 6 + c] * Vb[c * k + j];
                L[S::c_s2 + idx] = s;
            }
            real_t cond = gj_inverse<NT>(th, L + S::c_s2, k, k, L + S::c_s2, k, L + S::c_s1);
            if (!(cond > real_t(1e-13))) st_contact = 0;
            mm_nn<NT>(th, L + S::NwJw, k, Vb, k, L + S::c_s2, k, M, k, k);
            DWBC_SYNC();
            // projector on null(W):  P = Vb (Vb^T Vb)^-1 Vb^T
            mm_tn<NT>(th, L + S::c_s2, k, Vb, k, Vb, k, k, M, k);
            gj_inverse<NT>(th, L + S::c_s2, k, k, L + S::c_s2, k, L + S::c_s1);
            // W1 (M x k) temporarily = Vb * Gi  (stored in first M*k of W1)
            mm_nn<NT>(th, W1, k, Vb, k, L + S::c_s2, k, M, k, k);
            DWBC_SYNC();
            mm_nt<NT>(th, P, M, W1, k, Vb, k, M, k, M);
            DWBC_SYNC();
        }
        DWBC_STAMP(3);  // NwJw + projector done
        // alpha = trace(W)/M ;  W + alpha P is SPD ;  W^+ = (W + alpha P)^-1 - P/alpha
        real_t alpha = real_t(0.0);
        for (int i = 0; i < M; i++) alpha += AiNc[(6 + i) * N + 6 + i];
        alpha /= M;
        DWBC_SYNC();
        for (int idx = th.tid; idx < M * M; idx += NT) {
            const int i = idx / M, j = idx - i * M;
            // symmetrise: W is symmetric in exact arithmetic
            real_t w = real_t(0.5) * (AiNc[(6 + i) * N + 6 + j] + AiNc[(6 + j) * N + 6 + i]);
            W1[idx] = w + (k > 0 ? alpha * P[idx] : real_t(0.0));
        }
        DWBC_SYNC();
        {
            // spd_inverse scratch: reuse c_Y.. region? it is M*M <= C*N + M*K + ... : use c_Y (C*N=468 < M*M) -> not enough.
            // Use bufA itself as Tmp and write the result to W1's neighbour P afterwards.
            real_t *Out = W1;
            const int ok = spd_inverse_wave<M>(W1, M, W1, M, L + S::c_Y);
            if (!ok) st_contact = 0;
            DWBC_SYNC();
            const real_t ia = alpha != real_t(0.0) ? real_t(1.0) / alpha : real_t(0.0);
            for (int idx = th.tid; idx < M * M; idx += NT) Winv[idx] = Out[idx] - (k > 0 ? P[idx] * ia : real_t(0.0));
            DWBC_SYNC();
        }
        if (dump) {
            for (int idx = th.tid; idx < M * M; idx += NT) dump[dl.W_inv + idx] = Winv[idx];
            for (int idx = th.tid; idx < M * k; idx += NT) { dump[dl.NwJw + idx] = L[S::NwJw + idx]; }
        }
        DWBC_STAMP(4);  // W^+ done
        // FNl = A_rot * (J̄[:,6:] NwJw)   (cd x k), contact-local frame
        if (k > 0) {
            DWBC_SYNC();
            for (int idx = th.tid; idx < cd * k; idx += NT) {
                const int i = idx / k, j = idx - i * k;
                real_t s = real_t(0.0);
                _Pragma("unroll 8")
                for (int c = 0; c < M; c++) s += JbT[i * N + 6 + c] * L[S::NwJw + c * k + j];
                L[S::c_s1 + idx] = s;
            }
            DWBC_SYNC();
            for (int idx = th.tid; idx < cd * k; idx += NT) {
                const int i = idx / k, j = idx - i * k;
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
        if (dump)
            for (int i = th.tid; i < cd; i += NT) dump[dl.P_C + i] = L[S::PC + i];
    }

    DWBC_STAMP(5);  // gravity compensation done
    // ================= stage 3: task cascade (dwbc.cpp:685-873, 941-1127; wbd.cpp:207-261) =================
    const int nlim = su.has_tau_lim ? 2 * M : 0;
    const int ncone = 10 * nc;
    int st_task = 1, fail_level = -1;
    const io_t *fs_in = io.fstar + (size_t)inst * su.fstar_total;
    {
        real_t *Winv = L + S::bufA, *AiNc = L + S::bufN, *JbT = L + S::JbT;
        for (int lv = 0; lv < su.n_levels && st_task; lv++) {
            const int t = su.t_dof[lv], nv = t + k;
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
                    for (int a = 0; a < 3; a++) pl[a] = su.t_point[lv][li][a];
                const real_t *R = L + S::Rw + link * 9;
                real_t P[3];
                for (int a = 0; a < 3; a++) P[a] = L[S::pw + link * 3 + a] + R[a * 3] * pl[0] + R[a * 3 + 1] * pl[1] + R[a * 3 + 2] * pl[2];
                if (mode <= TASK_LINK_6D_CUSTOM_FRAME) { point_jacobian<N, NB, NT>(th, L + S::Rw, L + S::pw, L + S::aw, topo, nb, link, P, Jt, N, row, 6, 0); row += 6; }
                else if (mode <= TASK_LINK_POSITION_CUSTOM_FRAME) { point_jacobian<N, NB, NT>(th, L + S::Rw, L + S::pw, L + S::aw, topo, nb, link, P, Jt, N, row, 3, 1); row += 3; }
                else { point_jacobian<N, NB, NT>(th, L + S::Rw, L + S::pw, L + S::aw, topo, nb, link, P, Jt, N, row, 3, 2); row += 3; }
            }
            DWBC_SYNC();
            // --- CalculateJKT (wbd.cpp:207-213)
            mm_nn<NT>(th, T1, N, Jt, N, AiNc, N, t, N, N);           // J_t A^-1 N_c
            DWBC_SYNC();
            mm_nt<NT>(th, L + S::t_s2, t, T1, N, Jt, N, t, N, t);
            gj_inverse<NT>(th, L + S::t_s2, t, t, Lt, t, L + S::t_s1);  // Lambda_task
            for (int idx = th.tid; idx < t * M; idx += NT) {             // Q = (Lambda J A^-1 N_c)[:,6:]
                const int i = idx / M, j = idx - i * M;
                real_t s = real_t(0.0);
                _Pragma("unroll 8")
                for (int p = 0; p < t; p++) s += Lt[i * t + p] * T1[p * N + 6 + j];
                Q[idx] = s;
            }
            DWBC_SYNC();
            mm_nn<NT>(th, QW, M, Q, M, Winv, M, t, M, M);              // Q W^+
            DWBC_SYNC();
            mm_nt<NT>(th, L + S::t_s2, t, QW, M, Q, M, t, M, t);       // Q W^+ Q^T
            real_t cond = gj_inverse<NT>(th, L + S::t_s2, t, t, L + S::t_s3, t, L + S::t_s1);  // PinvCODWB (full rank case)
            if (!(cond > real_t(1e-6))) { st_task = 0; fail_level = lv; }
            for (int idx = th.tid; idx < M * t; idx += NT) {             // J_kt = W^+ Q^T pinv(.)
                const int i = idx / t, j = idx - i * t;
                real_t s = real_t(0.0);
                _Pragma("unroll 8")
                for (int p = 0; p < t; p++) s += QW[p * M + i] * L[S::t_s3 + p * t + j];
                Jkt[idx] = s;
            }
            DWBC_SYNC();
            // X = J_kt Lambda ;  Y = (J_t A^-1 N_c)[:,6:]   => Null_i = Null_{i-1} (I - X Y)   (wbd.cpp:257-261)
            real_t *X = (lv < kMaxLevels - 1) ? L + S::Xl + lv * M * T : L + S::t_QW;
            for (int idx = th.tid; idx < M * t; idx += NT) {
                const int i = idx / t, j = idx - i * t;
                real_t s = real_t(0.0);
                _Pragma("unroll 8")
                for (int p = 0; p < t; p++) s += Jkt[i * t + p] * Lt[p * t + j];
                X[i * T + j] = s;
                U[i * T + j] = s;
            }
            if (lv < kMaxLevels - 1)
                for (int idx = th.tid; idx < t * M; idx += NT) {
                    const int i = idx / M, j = idx - i * M;
                    L[S::Yl + lv * T * M + idx] = T1[i * N + 6 + j];
                }
            DWBC_SYNC();
            if (dump) {
                for (int idx = th.tid; idx < t * N; idx += NT) dump[dl.J_task + lv * T * N + idx] = Jt[idx];
                for (int idx = th.tid; idx < t * t; idx += NT) dump[dl.Lambda_task + lv * T * T + idx] = Lt[idx];
                for (int idx = th.tid; idx < M * t; idx += NT) dump[dl.J_kt + lv * M * T + idx] = Jkt[idx];
            }
            // U = Null_{lv-1} X = (I - X0 Y0)(I - X1 Y1)...(I - X_{lv-1} Y_{lv-1}) X   -- applied right to left
            for (int pl = lv - 1; pl >= 0; pl--) {
                const int tp = su.t_dof[pl];
                const real_t *Xp = L + S::Xl + pl * M * T, *Yp = L + S::Yl + pl * T * M;
                DWBC_SYNC();
                for (int idx = th.tid; idx < tp * t; idx += NT) {       // Z = Yp U  (tp x t)
                    const int i = idx / t, j = idx - i * t;
                    real_t s = real_t(0.0);
                    _Pragma("unroll 8")
                    for (int c = 0; c < M; c++) s += Yp[i * M + c] * U[c * T + j];
                    L[S::t_s2 + idx] = s;
                }
                DWBC_SYNC();
                for (int idx = th.tid; idx < M * t; idx += NT) {
                    const int i = idx / t, j = idx - i * t;
                    real_t s = U[i * T + j];
                    _Pragma("unroll 8")
                    for (int p = 0; p < tp; p++) s -= Xp[i * T + p] * L[S::t_s2 + p * t + j];
                    U[i * T + j] = s;
                }
            }
            DWBC_SYNC();
            DWBC_STAMP(6 + 3 * lv);  // level lv: J_kt, Lambda, null-space chain done
            // --- QP rows (dwbc.cpp:988-1053)
            const io_t *fs = fs_in + su.fstar_off[lv];
            real_t *base = L + S::t_base, *F = L + S::t_F, *fv = L + S::t_fv;
            for (int i = th.tid; i < M; i += NT) {
                real_t s = L[S::tg + i] + L[S::tt + i];
                _Pragma("unroll 8")
                for (int j = 0; j < t; j++) s += U[i * T + j] * fs[j];
                base[i] = s;
            }
            DWBC_SYNC();
            // contact wrench map in the contact-local frame: F (cd x t) = A_rot J̄[:,6:] U ; fv = A_rot (J̄[:,6:] base - P_C)
            for (int idx = th.tid; idx < cd * (t + 1); idx += NT) {
                const int i = idx / (t + 1), j = idx - i * (t + 1);
                real_t s = real_t(0.0);
                if (j < t) { for (int c = 0; c < M; c++) s += JbT[i * N + 6 + c] * U[c * T + j]; }
                else { for (int c = 0; c < M; c++) s += JbT[i * N + 6 + c] * base[c]; s -= L[S::PC + i]; }
                L[S::t_s1 + i * (T + 1) + j] = s;  // T+1 = 7 columns; C x 7 = 84 <= T*2T = 72?  -> use qp_Nm.. as scratch
            }
            DWBC_SYNC();
            for (int idx = th.tid; idx < cd * (t + 1); idx += NT) {
                const int i = idx / (t + 1), j = idx - i * (t + 1);
                const int a = i / 6, h = (i % 6) / 3, x = i % 3;
                const real_t *R = L + S::Rc + a * 9;
                const real_t *src = L + S::t_s1 + (6 * a + 3 * h) * (T + 1) + j;
                const real_t v = R[0 * 3 + x] * src[0] + R[1 * 3 + x] * src[T + 1] + R[2 * 3 + x] * src[2 * (T + 1)];
                if (j < t) F[i * kQpLd + j] = v; else fv[i] = v;
            }
            DWBC_SYNC();
            DWBC_STAMP(7 + 3 * lv);  // level lv: QP inputs assembled
            QpResult qres;
            qp_rows_and_solve<N, NB>(su, L, nlim, ncone, act_c[0], act_c[1], U, T, t, L + S::NwJw, k, k, kQpScaleGI, F, kQpLd,
                                     L + S::FNl, k, fv, base, t, su.qp_max_iter_task, qres, L + S::qp_V, L + S::qp_x);
            const int ok = qres.status;
            const real_t viol = qres.viol;
            if (diag && th.tid == 0) {
                diag[DG_QP_ITER + lv] = qres.iters;
                diag[DG_QP_NACT + lv] = qres.nact;
                for (int a = 0; a < kQpLd; a++) diag[DG_QP_ACT + lv * kQpLd + a] = qres.act[a];
            }
            if (dump && th.tid == 0) dump[dl.qp_viol + lv] = viol;
            DWBC_STAMP(8 + 3 * lv);  // level lv: QP solved
            if (!ok) { st_task = 0; fail_level = lv; break; }  // f_star_qp_, contact_qp_ zero; cascade aborts (dwbc.cpp:836,1119)
            const real_t *x = L + S::qp_x;
            // torque_task_ += Null_{i-1} J_kt Lambda (f* + f*_qp) ; torque_contact_ = NwJw contact_qp_ (dwbc.cpp:839-851)
            for (int i = th.tid; i < M; i += NT) {
                real_t s = real_t(0.0);
                _Pragma("unroll 8")
                for (int j = 0; j < t; j++) s += U[i * T + j] * (fs[j] + x[j]);
                L[S::tt + i] += s;
                real_t c = real_t(0.0);
                _Pragma("unroll 8")
                for (int j = 0; j < k; j++) c += L[S::NwJw + i * k + j] * x[t + j];
                L[S::tc + i] = c;
            }
            if (dump) {
                for (int j = th.tid; j < t; j += NT) dump[dl.fstar_qp + lv * T + j] = x[j];
                for (int j = th.tid; j < k; j += NT) dump[dl.contact_qp + lv * (C - 6) + j] = x[t + j];
            }
            DWBC_SYNC();
        }
    }

    DWBC_STAMP(14);  // (levels 0..2 use stamps 6..14)
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
        QpResult qres;
        qp_rows_and_solve<N, NB>(su, L, nlim, ncone, act_c[0], act_c[1], L + S::NwJw, k, k, L + S::NwJw, k, 0, real_t(1.0), L + S::FNl, k,
                                 L + S::FNl, k, fv, base, k, su.qp_max_iter_contact, qres, L + S::qp_V, L + S::qp_x);
        const int ok = qres.status;
        const real_t viol = qres.viol;
        if (diag && th.tid == 0) {
            diag[DG_QP_ITER + kMaxLevels] = qres.iters;
            diag[DG_QP_NACT + kMaxLevels] = qres.nact;
            for (int a = 0; a < kQpLd; a++) diag[DG_QP_ACT + kMaxLevels * kQpLd + a] = qres.act[a];
        }
        if (dump && th.tid == 0) dump[dl.qp_viol + kMaxLevels] = viol;
        const real_t *x = L + S::qp_x;
        if (ok) {
            for (int i = th.tid; i < M; i += NT) {
                real_t c = real_t(0.0);
                _Pragma("unroll 8")
                for (int j = 0; j < k; j++) c += L[S::NwJw + i * k + j] * x[j];
                L[S::tc + i] += c;
            }
            if (dump)
                for (int j = th.tid; j < k; j += NT) dump[dl.cf_redis + j] = x[j];
        } else {
            st_redis = 0;
            for (int i = th.tid; i < M; i += NT) L[S::tc + i] = real_t(0.0);
        }
    } else {
        for (int i = th.tid; i < M; i += NT) L[S::tc + i] = real_t(0.0);  // dwbc.cpp:1562-1567
    }
    DWBC_SYNC();

    DWBC_STAMP(15);  // contact redistribution done
    // ================= outputs =================
    io_t *tau = io.tau + (size_t)inst * 3 * M;
    for (int i = th.tid; i < 3 * M; i += NT) tau[i] = L[S::tg + i];
    io_t *wr = io.wrench + (size_t)inst * 12;
    for (int i = th.tid; i < 12; i += NT) {
        real_t s = real_t(0.0);
        if (i < cd) {
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

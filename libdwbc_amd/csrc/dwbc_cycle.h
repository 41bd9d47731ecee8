// dwbc_cycle.h -- helpers shared by the fused cycle kernels (dwbc_cycle2.h: full model, dwbc_reduced.h: reduced dynamics,
// dwbc_nohqp.h, dwbc_hqp.h): synchronisation and stage-stamp macros, QP constants, small dense products on LDS operands, the
// contact-cone rows, point Jacobians, the centroidal outputs and the assembly of the QP rows into lanes.
//
// One workgroup (one 64-lane wavefront) owns ONE robot instance for the whole cycle; the kernels that include this keep every
// intermediate in registers / LDS.  The helpers are NT-generic (strided loops, explicit synchronisation points) so that the SAME
// source compiles with g++ for one host thread (tests/emu) to check the arithmetic without a GPU.  That build is a test
// harness only; the product library has no CPU path.
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
//   CQuadraticProgram::SolveQPoases      src/qp_wrapper.cpp:192-380 -> replaced by qp_solve_wave() (dwbc_qp_wave.h)
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
// every stamp is a fence plus a store and moves the register allocation: a build with all 40 fine stamps spills in the
// sweeps.  -DDWBC_FINE_MASK=0x...ull keeps only the stamps whose bit is set (a handful per build reads true)
#ifndef DWBC_FINE_MASK
#define DWBC_FINE_MASK 0ull
#endif
#define DWBC_FSTAMP(i)                                                                  \
    do {                                                                                \
        if ((DWBC_FINE_MASK >> (i)) & 1ull) {                                           \
            DWBC_SYNC();                                                                \
            if (diag && th.tid == 0) diag[DG_FTIME + (i)] = (int)(clock64() - t_start_); \
        }                                                                               \
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
constexpr real_t kCodThreshold = kF32 ? real_t(1.0e-4) : real_t(1.0e-6);  // COD_THRESHOLD of the reference (include/dwbc_wbd.h:10); fp32 cannot resolve it
constexpr real_t kCodCondFast = real_t(3.0e3);  // condition estimate of Lambda_task^-1 below which J_kt = (T1r W^+)^T is taken (dwbc_cycle2.h, stage 3a)
constexpr real_t kCodCheck = kF32 ? real_t(1.0e-3) : real_t(1.0e-4);      // Cholesky pivot ratio below which a t x t block takes the rank-revealing route
constexpr real_t kQpReorth = real_t(1.0e-2);  // |H n|^2 below which the new direction is projected a second time (dwbc_qp_wave.h)
#ifndef DWBC_QP_REFINE
#define DWBC_QP_REFINE 10
#endif
constexpr int kQpRefine = DWBC_QP_REFINE;   // most iterated-Tikhonov steps towards the lexicographic point (dwbc_qp_wave.h)
constexpr real_t kQpNullDir = kF32 ? real_t(1.0e-6) : real_t(1.0e-12);   // curvature / |p|^2 below which a CG direction of the lexicographic solve is a null direction (18-variable QPs)
constexpr real_t kQpNullRes = kF32 ? real_t(1.0e-4) : real_t(1.0e-9);    // relative residual accepted when the solve stopped at such a direction
constexpr real_t kQpRefineTol = kF32 ? real_t(1.0e-5) : real_t(1.0e-12);  // a step's change / |x| below which the sequence has settled
}  // namespace dwbc
#include "dwbc_qp_wave.h"
namespace dwbc {


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
// what a lane needs of the problem set-up to fill its QP row: fetched ONCE per cycle, ahead of the cascade (the set-up is a kernel
// argument, and a per-lane index into it is a vector memory load -- a trip to L2 in each of the three QPs when taken inside the fill)
struct QpLaneConst {
    PL(real_t, taul);       // torque limit of row `lane` (lane < M)
    PL(real_t, c2);         // cone lanes: the row is  c2 * w[2] + sg * w[oi]  on the local wrench of contact a (reference src/wbd.cpp:59-97)
    PL(real_t, sg);
    PL(int, row2);
    PL(int, rowo);
};
template <int N>
DWBC_DEV void qp_lane_consts(const Setup &su, int ci0, int ci1, QpLaneConst &qc) {
    constexpr int M = N - 6;
    DWBC_LANE_DECL;
    LANES {
        LV(qc.taul) = lane < M ? (real_t)su.tau_lim[lane] : real_t(0.0);
        const int rr = lane >= M ? lane - M : 0, a = rr >= 10 ? 1 : 0, r10 = rr - 10 * a;
        const int ci = a ? ci1 : ci0;
        const int pr = r10 >> 1;
        LV(qc.c2) = -(real_t)(pr == 0 ? su.c_lx[ci] : pr == 1 ? su.c_ly[ci] : pr == 4 ? su.c_muz[ci] : su.c_mu[ci]);
        const int oi = pr == 0 ? 4 : pr == 1 ? 3 : pr == 2 ? 0 : pr == 3 ? 1 : 5;
        LV(qc.sg) = (pr < 2) ? ((r10 & 1) ? real_t(1.0) : -real_t(1.0)) : ((r10 & 1) ? -real_t(1.0) : real_t(1.0));
        LV(qc.row2) = 6 * a + 2;
        LV(qc.rowo) = 6 * a + oi;
    }
}

// norm of the lane's row of the contact redistribution QP (torque rows: NwJw[lane, :]; cone rows: c2 FN[row2, :] + sg FN[rowo, :]), with
// the zero-row rule of the canon (a row below kQpZeroRow is the constraint 0 <= hi: its slack is taken as it is)
template <int N, class S>
DWBC_DEV void redis_row_norms(const real_t *L, int nlim, int ncone, int k, const QpLaneConst &qc, const real_t *FN, PL_REF(real_t, grn)) {
    constexpr int M = N - 6, WLD = S::WLD;
    DWBC_LANE_DECL;
    LANES {
        const bool tq = lane < M && nlim != 0, cn = lane >= M && lane - M < ncone;
        const real_t *pa = tq ? L + S::NwJw + lane * 6 : FN + (cn ? LV(qc.row2) : 0) * WLD;
        const real_t *pb = FN + (cn ? LV(qc.rowo) : 0) * WLD;
        const real_t ca = tq ? real_t(1.0) : LV(qc.c2), cb = tq ? real_t(0.0) : LV(qc.sg);
        real_t s2 = real_t(0.0);
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const real_t v = j < k ? ca * pa[j] + cb * pb[j] : real_t(0.0);
            s2 += v * v;
        }
        LV(grn) = s2 < kQpZeroRow * kQpZeroRow ? real_t(1.0) : sqrt(s2);
    }
}

template <int N, int NB, int WS = 1>
DWBC_DEV void qp_rows_and_solve(const Setup &su, real_t *L, int nlim, int ncone, int ci0, int ci1, const real_t *P1, int ld1,
                                int t1, const real_t *P2, int ld2, int t2, real_t s2, const real_t *W1, int ldw1,
                                const real_t *W2, int ldw2, const real_t *fv, const real_t *base, int tvars, int max_iter,
                                QpResult &res, real_t *Vlds, real_t *xlds, const int *warm, const QpLaneConst *qcp,
                                real_t vtol, PL_REF(real_t, sfin)) {
    constexpr int M = N - 6;
    DWBC_LANE_DECL;
    QpRows R;
    const int nv = t1 + t2;
#if defined(DWBC_STAGE_TIMERS) && !defined(DWBC_HOST_EMU)
    const long long t_fill0_ = clock64();
#endif
    QpLaneConst qloc;
    if (!qcp) qp_lane_consts<N>(su, ci0, ci1, qloc);  // (callers that solve one QP only)
    const QpLaneConst &qc = qcp ? *qcp : qloc;
    // One straight-line path for every lane: entry j of a row is  ca * A_j + cb * B_j  with A / B taken from the left block (j < t1) or the
    // right block, by UNCONDITIONAL loads (clamped addresses, the value masked afterwards) -- 24 independent LDS reads the scheduler can
    // keep in flight together.  (Rounds 1-2 filled torque and cone lanes in two divergent branches with a load behind every `if (j < t1)`:
    // each read waited out its own LDS round trip, 3.3 k cycles per QP in the stage table.)
    //   torque row r (lane < M):  A = [P1 | P2][r, :], ca = 1, cb = 0                         (reference src/dwbc.cpp:1001-1016)
    //   cone row (lane >= M):     A = [W1 | W2][row2, :], B = [W1 | W2][rowo, :], ca = -c2, cb = -sg  (reference src/dwbc.cpp:1041-1053, wbd.cpp:59-97)
    LANES {
        const bool tq = lane < M && nlim != 0;
        const bool cn = lane >= M && lane - M < ncone;
        const int row2 = LV(qc.row2), rowo = LV(qc.rowo);
        const real_t *pa1 = tq ? P1 + lane * ld1 : W1 + (cn ? row2 : 0) * ldw1;
        const real_t *pa2 = tq ? P2 + lane * ld2 : W2 + (cn ? row2 : 0) * ldw2;
        const real_t *pb1 = W1 + (cn ? rowo : 0) * ldw1;
        const real_t *pb2 = W2 + (cn ? rowo : 0) * ldw2;
        const real_t ca = tq ? real_t(1.0) : -LV(qc.c2), cb = tq ? real_t(0.0) : -LV(qc.sg);
        real_t va[kQpN], vb[kQpN];
#pragma unroll
        for (int j = 0; j < kQpN; j++) {
            const bool f1 = j < t1;
            const int j2 = j - t1 > 0 ? (j - t1 < (t2 > 0 ? t2 : 1) ? j - t1 : 0) : 0;  // clamped column of the right block
            const int j1 = f1 ? j : 0;
            va[j] = f1 ? pa1[j1] : pa2[j2];
            vb[j] = f1 ? pb1[j1] : pb2[j2];
        }
#pragma unroll
        for (int j = 0; j < kQpN; j++) {
            const real_t v = (ca * va[j] + cb * vb[j]) * (j < t1 ? real_t(1.0) : s2);
            LV(R.g)[j] = ((tq || cn) && j < nv) ? v : real_t(0.0);
        }
        LV(R.hi) = tq ? LV(qc.taul) - base[tq ? lane : 0] : (cn ? -(ca * fv[row2] + cb * fv[rowo]) : DWBC_QP_INF);
        LV(R.lo) = tq ? LV(qc.taul) + base[tq ? lane : 0] : DWBC_QP_INF;
        LV(R.id_hi) = tq ? lane : (cn ? nlim + (lane - M) : -1);
        LV(R.id_lo) = tq ? M + lane : -1;
    }
#if defined(DWBC_STAGE_TIMERS) && !defined(DWBC_HOST_EMU)
    const long long t_fill1_ = clock64();
#endif
    // the solver is instantiated for 12, 9 and 6 variables (6 + 6, 3 + 6 and the 6 contact-null variables of the redistribution)
    if (nv <= 6) qp_solve_wave<WS, 6>(R, nv, tvars, max_iter, res, Vlds, warm, vtol, sfin);
    else if (nv <= 9) qp_solve_wave<WS, 9>(R, nv, tvars, max_iter, res, Vlds, warm, vtol, sfin);
    else qp_solve_wave<WS, 12>(R, nv, tvars, max_iter, res, Vlds, warm, vtol, sfin);
    LANES {
        if (lane < kQpN) xlds[lane] = pick12(res.x, lane);
    }
    DWBC_SYNC();
#if defined(DWBC_STAGE_TIMERS) && !defined(DWBC_HOST_EMU)
    res.tm[8] = t_fill1_ - t_fill0_;
#endif
}

// (callers that do not look at the final slacks: the reduced-dynamics cycle)
template <int N, int NB, int WS = 1>
DWBC_DEV void qp_rows_and_solve(const Setup &su, real_t *L, int nlim, int ncone, int ci0, int ci1, const real_t *P1, int ld1,
                                int t1, const real_t *P2, int ld2, int t2, real_t s2, const real_t *W1, int ldw1,
                                const real_t *W2, int ldw2, const real_t *fv, const real_t *base, int tvars, int max_iter,
                                QpResult &res, real_t *Vlds, real_t *xlds, const int *warm = nullptr) {
    PL(real_t, sfin_unused);
    qp_rows_and_solve<N, NB, WS>(su, L, nlim, ncone, ci0, ci1, P1, ld1, t1, P2, ld2, t2, s2, W1, ldw1, W2, ldw2, fv, base, tvars, max_iter, res, Vlds,
                                 xlds, warm, nullptr, kQpTol, sfin_unused);
}

}  // namespace dwbc

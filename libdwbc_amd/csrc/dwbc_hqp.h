// dwbc_hqp.h -- the reference's generic hierarchical-QP class (HQP / HQP_Hierarch, include/dwbc_hqp.h, src/dwbc_hqp.cpp) for a
// batch of instances, and the LQP configurator that fills it from a solved control cycle (RobotData::ConfigureLQP,
// src/dwbc.cpp:4304-4430).  SURVEY 8 rows a16 / f3.  One workgroup per instance (256 threads on the device, dwbc_hqp_capi.hip).
//
//   level i:   min_{u, v}  1/2 |B_i Z u + (B_i y_prev + b_i)|^2  [+ 1/2 u^T Z^T H Z u + (Z^T H y_prev)^T u]  + 1/2 |v|^2
//              s.t.        A_i Z u - v <= -(A_i y_prev) - a_i                      own inequalities, slack v   (dwbc_hqp.cpp:320-336)
//                          A_j Z u     <= -(A_j y_prev) + v_ans_j - a_j   (j < i)  earlier levels, slack frozen (dwbc_hqp.cpp:366-384)
//              y_i = y_prev + Z u,  Z = Z_{i-1},  Z_i = Z_{i-1} null(B_i Z_{i-1})    (dwbc_hqp.cpp:23-85, 388; math.cpp:349-360)
//
// The reference hands each level to OSQP (OsqpEigen, not vendored; ADMM at default tolerances on sparseView(1e-5) copies,
// dwbc_hqp.cpp:583-631).  Here every level is solved EXACTLY by a dual active-set method in range-space form on the augmented
// variable (u, v): the working set holds own rows whose slack is positive and earlier rows at their bound; with T_a = H^-1 c_a
// kept per working-set row, M = C_W H^-1 C_W^T (+1 on the diagonal of own rows) is q x q and is refactorised every step.
// The Hessian in u is only positive semi-definite in general (the LQP's internal-wrench directions carry no cost); canon, as
// in oracle/hqp_np.py: Tikhonov term eps |u|^2 / 2.  PARITY UNPINNED in the reference (no fixture, no assertion).
//
// Storage: the level matrices, answers and the scratch (Z, H Z, C Z) live in HBM -- this is not the headline path; LDS holds
// the solver state (H^-1, T, M, the vectors).  Code is NT-generic (strided loops, every hand-over between threads behind a
// workgroup barrier) so that tests/emu runs it with one host thread and the device with several wavefronts.
#pragma once
#include <cstdlib>

#include "dwbc_cycle.h"

// this path keeps matrices in HBM that lanes of the wave hand to each other: a full workgroup barrier (waits for outstanding
// vector-memory and LDS operations), not the wavefront-scope compiler fence of the cycle kernel
#ifdef DWBC_HOST_EMU
#define HQP_SYNC() ((void)0)
#else
#define HQP_SYNC() __syncthreads()
#endif
// ordering point inside ONE wavefront (its lanes run in lockstep and its LDS operations execute in order: only the compiler has to be kept
// from moving LDS accesses across it) -- for the steps that wave 0 runs alone between two workgroup barriers
#if defined(DWBC_HOST_EMU)
#define HQP_WSYNC() ((void)0)
#else
#define HQP_WSYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
#endif

namespace dwbc {

constexpr int kHqpMaxLevels = 8;
constexpr int kHqpMaxQ = 32;   // working-set capacity of one level's QP
constexpr int kHqpW0 = 64;     // lanes of the one wave that refreshes the Cholesky factor of the working set's matrix
constexpr int kArgMax = 32;    // groups of 16 rows in the two-stage arg-min: up to 512 constraint rows of one level's QP
constexpr int kHqpMaxEq = 32;  // equality rows of one level
constexpr int kHqpMaxVar = 64; // y size (acceleration + torque + contact)

struct HqpDesc {
    int nv, n_levels, solve_first, max_iter;
    double eps, tol;
    int m[kHqpMaxLevels], e[kHqpMaxLevels], has_cost[kHqpMaxLevels];
    int exact[kHqpMaxLevels];  // 1: the level's equalities hold EXACTLY (least-norm step in the current null space), its inequalities are
                               // only declared (slack 0) and become hard rows of the later levels -- the JACC formulation's constraints
    // offsets (doubles) inside one instance's record
    int oA[kHqpMaxLevels], oa[kHqpMaxLevels], oB[kHqpMaxLevels], ob[kHqpMaxLevels], oH[kHqpMaxLevels];
    int oy[kHqpMaxLevels], ov[kHqpMaxLevels], ow[kHqpMaxLevels];
    int rec;       // doubles per instance record
    int max_rows;  // most inequality rows (own + earlier) any level sees
    int scratch;   // doubles per instance of scratch: Z (nv x nv) | HZ (nv x nv) | CZ (max_rows x nv)
    int lds;       // doubles of LDS
    int scratch_in_lds;  // bits: 1 = Z, 2 = H Z, 4 = C Z sit behind the solver state in LDS (as far as the 160 KB of a CU reach); else HBM (io.scratch)
};
constexpr int kHqpLdsBudget = 160 * 1024 / 8 - 64;  // doubles of LDS one workgroup may take
enum HqpStat { HQS_STATUS = 0, HQS_ITER = kHqpMaxLevels, HQS_NULL = 2 * kHqpMaxLevels, HQS_COUNT = 3 * kHqpMaxLevels };
struct HqpIO {
    int B;
    double *rec;
    double *scratch;
    int *stat;  // B x HQS_COUNT
};

// LDS map of the solver (doubles)
struct HqpLds {
    int Hinv, T, Mm, Lc, Bz, u, g, zu, tp, yp, cp, hy, rr, r, lam, rhs, vv, dd, sl, inw, wl, am, total;
    __host__ __device__ static HqpLds make(int nv, int max_rows) {
        HqpLds l;
        int o = 0;
        auto ev = [](int a) { return (a + 1) & ~1; };
        l.Hinv = o; o += ev(nv * nv);
        l.T = o; o += ev(kHqpMaxQ * nv);
        l.Mm = o; o += kHqpMaxQ * kHqpMaxQ;
        l.Lc = o; o += kHqpMaxQ * kHqpMaxQ;
        l.Bz = o; o += ev(kHqpMaxEq * nv);
        l.u = o; o += ev(nv);
        l.g = o; o += ev(nv);
        l.zu = o; o += ev(nv);
        l.tp = o; o += ev(nv);
        l.yp = o; o += ev(nv);
        l.cp = o; o += ev(nv);
        l.hy = o; o += ev(nv);
        l.rr = o; o += kHqpMaxEq;
        l.r = o; o += kHqpMaxQ;
        l.lam = o; o += kHqpMaxQ;
        l.rhs = o; o += kHqpMaxQ;
        l.vv = o; o += ev(max_rows);
        l.dd = o; o += ev(max_rows);
        l.sl = o; o += ev(max_rows);
        l.inw = o; o += ev(max_rows);
        l.wl = o; o += kHqpMaxQ;
        l.am = o; o += 2 * kArgMax;   // group minima of the arg-min and their rows
        l.total = o + 8;
        return l;
    }
};

// record layout from the level sizes (host).  share_cost: every level with a cost reads ONE H block (the LQP gives all its
// levels the same cost matrix, dwbc.cpp:4338-4342,4408,4424)
inline void hqp_layout(HqpDesc &d, bool share_cost) {
    int o = 0, shared = -1, rows = 0, max_rows = 0;
    const int nv = d.nv;
    for (int i = 0; i < d.n_levels; i++) {
        d.oA[i] = o; o += d.m[i] * nv;
        d.oa[i] = o; o += d.m[i];
        d.oB[i] = o; o += d.e[i] * nv;
        d.ob[i] = o; o += d.e[i];
        if (d.has_cost[i]) {
            if (share_cost && shared >= 0) d.oH[i] = shared;
            else { d.oH[i] = o; shared = o; o += nv * nv; }
        } else d.oH[i] = -1;
        d.oy[i] = o; o += nv;
        d.ov[i] = o; o += d.m[i];
        d.ow[i] = o; o += d.e[i];
        rows += d.m[i];
        max_rows = rows > max_rows ? rows : max_rows;
    }
    d.rec = (o + 1) & ~1;
    d.max_rows = max_rows > 0 ? max_rows : 1;
    d.scratch = 2 * nv * nv + d.max_rows * nv;
    d.lds = HqpLds::make(nv, d.max_rows).total;
    // Z, H Z and C Z are read and re-read inside a level -- C Z by every active-set step, Z by every product that forms the level,
    // H Z once.  VERDICT r2's "obvious first move" -- keep them in LDS -- was built and measured (DWBC_HQP_SCRATCH_LDS=1: they go to LDS
    // in that order of preference as far as the CU's 160 KB reach behind the solver state; the TOCABI LQP, 51 variables and 152 rows:
    // C Z 62 KB + Z 21 KB next to 72 KB of state, H Z stays in HBM) and does NOT pay: the map then allows one workgroup per CU instead
    // of two, and the cascade goes from 4.4 to 5.4 ms per 1024 instances (profiles/r03_lqp_summary.txt).  The default stays HBM.
    d.scratch_in_lds = 0;
    if (getenv("DWBC_HQP_SCRATCH_LDS")) {
        const int sz[3] = {d.max_rows * nv, nv * nv, nv * nv}, bit[3] = {4, 1, 2};
        for (int i = 0; i < 3; i++)
            if (d.lds + sz[i] <= kHqpLdsBudget) { d.lds += sz[i]; d.scratch_in_lds |= bit[i]; }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Z_view <- Z_view * Q, Q from the column-pivoted Householder QR of (Bm Z_view)^T; the view drops its first `rank` columns
// (getNullSpace, src/math.cpp:349-360: Eigen COD with its default threshold [ext]; restated as oracle/hqp_np.py::get_null_space)
// Z: nv x nv row-major (HBM), view = columns [off, nv).  Bm: e x nv.  Returns the rank.
// ------------------------------------------------------------------------------------------------------------------
template <int NT>
DWBC_DEV int hqp_null_update(Thr th, double *Z, int nv, int off, const double *Bm, int e, double *Bz, double *vh, double *rr = nullptr) {
    const int k = nv - off;
    if (e == 0 || k == 0) return 0;
    HQP_SYNC();
    for (int idx = th.tid; idx < e * k; idx += NT) {
        const int i = idx / k, j = idx - i * k;
        double acc = 0.0;
        for (int c = 0; c < nv; c++) acc += Bm[i * nv + c] * Z[c * nv + off + j];
        Bz[i * k + j] = acc;
    }
    HQP_SYNC();
    const int steps = e < k ? e : k;
    double maxp = 0.0;
    int rank = 0;
    const double thr = 2.220446049250313e-16 * steps;
    for (int s = 0; s < steps; s++) {
        // pivot: the remaining row of Bz (= column of (B Z)^T) with the largest norm over entries [s, k)
        int jp = s;
        double best = -1.0;
        for (int i = s; i < e; i++) {
            double n2 = 0.0;
            for (int c = s; c < k; c++) n2 += Bz[i * k + c] * Bz[i * k + c];
            if (n2 > best) { best = n2; jp = i; }
        }
        HQP_SYNC();
        if (jp != s) {
            for (int c = th.tid; c < k; c += NT) { const double t = Bz[s * k + c]; Bz[s * k + c] = Bz[jp * k + c]; Bz[jp * k + c] = t; }
            if (rr && th.tid == 0) { const double t = rr[s]; rr[s] = rr[jp]; rr[jp] = t; }  // the right-hand side follows its row
        }
        HQP_SYNC();
        const double nx = sqrt(best > 0.0 ? best : 0.0);
        if (nx == 0.0) break;
        const double x0 = Bz[s * k + s];
        const double alpha = x0 > 0.0 ? -nx : nx;
        double vn2 = 0.0;
        for (int c = s; c < k; c++) {
            const double vc = Bz[s * k + c] - (c == s ? alpha : 0.0);
            vn2 += vc * vc;
        }
        HQP_SYNC();
        for (int c = th.tid; c < k; c += NT) vh[c] = c < s ? 0.0 : Bz[s * k + c] - (c == s ? alpha : 0.0);
        HQP_SYNC();
        if (vn2 > 0.0) {
            const double beta = 2.0 / vn2;
            for (int i = s + 1 + th.tid; i < e; i += NT) {  // remaining rows of Bz
                double dd = 0.0;
                for (int c = s; c < k; c++) dd += Bz[i * k + c] * vh[c];
                dd *= beta;
                for (int c = s; c < k; c++) Bz[i * k + c] -= dd * vh[c];
            }
            for (int i = th.tid; i < nv; i += NT) {  // rows of the view of Z
                double dd = 0.0;
                for (int c = s; c < k; c++) dd += Z[i * nv + off + c] * vh[c];
                dd *= beta;
                for (int c = s; c < k; c++) Z[i * nv + off + c] -= dd * vh[c];
            }
        }
        HQP_SYNC();
        if (th.tid == 0) Bz[s * k + s] = alpha;  // Bz now holds L of (B Z) = Pi L Q^T: rows 0..s final in columns 0..s
        const double ap = fabs(alpha);
        if (s == 0) maxp = ap;
        // pivots come out in non-increasing magnitude: the first one below the threshold ends the rank
        if (ap > thr * maxp) rank++;
        else break;
        HQP_SYNC();
    }
    HQP_SYNC();
    return rank;
}

// in-place inverse of the SPD k x k matrix S (LDS, row stride k) by Gauss-Jordan without pivoting; col / row: k doubles of
// scratch each.  Returns 0 on a non-positive pivot.
template <int NT>
DWBC_DEV int hqp_spd_inverse(Thr th, double *S, int k, double *col, double *row) {
    int ok = 1;
    for (int p = 0; p < k; p++) {
        HQP_SYNC();
        double d = S[p * k + p];
        if (!(d > 0.0)) { ok = 0; d = 1.0; }
        const double rp = 1.0 / d;
        for (int i = th.tid; i < k; i += NT) { col[i] = S[i * k + p]; row[i] = S[p * k + i]; }
        HQP_SYNC();
        for (int idx = th.tid; idx < k * k; idx += NT) {
            const int i = idx / k, j = idx - i * k;
            double v;
            if (i == p && j == p) v = rp;
            else if (i == p) v = row[j] * rp;
            else if (j == p) v = -col[i] * rp;
            else v = S[idx] - col[i] * row[j] * rp;
            S[idx] = v;
        }
    }
    HQP_SYNC();
    return ok;
}

// ------------------------------------------------------------------------------------------------------------------
// one instance: the cascade over the levels
// ------------------------------------------------------------------------------------------------------------------
template <int NT>
DWBC_DEV void hqp_instance(Thr th, const HqpDesc &d, const HqpIO &io, int inst, double *L) {
    const int nv = d.nv;
    const HqpLds l = HqpLds::make(nv, d.max_rows);
    double *rec = io.rec + (size_t)inst * d.rec;
    double *Z = io.scratch + (size_t)inst * d.scratch, *HZ = Z + nv * nv, *CZ = HZ + nv * nv;
    {
        int o = l.total;  // LDS placement in the order hqp_layout() granted it: C Z, Z, H Z
        if (d.scratch_in_lds & 4) { CZ = L + o; o += d.max_rows * nv; }
        if (d.scratch_in_lds & 1) { Z = L + o; o += nv * nv; }
        if (d.scratch_in_lds & 2) { HZ = L + o; o += nv * nv; }
    }
    int *stat = io.stat + (size_t)inst * HQS_COUNT;
    for (int idx = th.tid; idx < nv * nv; idx += NT) Z[idx] = (idx / nv == idx % nv) ? 1.0 : 0.0;
    for (int i = th.tid; i < HQS_COUNT; i += NT) stat[i] = i < kHqpMaxLevels ? 1 : 0;
    HQP_SYNC();
    int off = 0;
    for (int lv = 0; lv < d.n_levels; lv++) {
        const int k = nv - off;
        const bool solve = (lv > 0 || d.solve_first) && !d.exact[lv];
        const double *Bm = rec + d.oB[lv], *bv = rec + d.ob[lv];
        const int e = d.e[lv], mo = d.m[lv];
        if (d.exact[lv]) {
            // ---- hard equalities: y = y_prev + Z_new[:, :rank] w with L w = -Pi^T (B y_prev + b), (B Z) = Pi L Q^T from the same
            //      Householder steps that extend the null-space chain (Z_new = Z Q); the level's own inequality rows keep slack 0
            double *yp = L + l.yp, *rr = L + l.rr, *u = L + l.u, *Bz = L + l.Bz;
            HQP_SYNC();
            for (int i = th.tid; i < nv; i += NT) yp[i] = lv > 0 ? rec[d.oy[lv - 1] + i] : 0.0;
            HQP_SYNC();
            for (int i = th.tid; i < e; i += NT) {
                double acc = bv[i];
                for (int c = 0; c < nv; c++) acc += Bm[i * nv + c] * yp[c];
                rr[i] = acc;
            }
            HQP_SYNC();
            const int rank = hqp_null_update<NT>(th, Z, nv, off, Bm, e, Bz, L + l.zu, rr);
            int status = 1;
            if (th.tid == 0)
                for (int i = 0; i < rank; i++) {  // forward substitution, serial (rank <= 32)
                    double acc = -rr[i];
                    for (int c = 0; c < i; c++) acc -= Bz[i * k + c] * u[c];
                    u[i] = acc / Bz[i * k + i];
                }
            HQP_SYNC();
            // rows beyond the rank must be consistent with the rows kept: |residual| of the dependent rows
            for (int i = rank; i < e; i++) {
                double acc = rr[i];
                for (int c = 0; c < rank; c++) acc += Bz[i * k + c] * u[c];
                if (fabs(acc) > 1.0e-6) status = 0;
            }
            for (int i = th.tid; i < nv; i += NT) {
                double acc = yp[i];
                for (int c = 0; c < rank; c++) acc += Z[i * nv + off + c] * u[c];
                rec[d.oy[lv] + i] = acc;
            }
            for (int i = th.tid; i < mo; i += NT) rec[d.ov[lv] + i] = 0.0;
            HQP_SYNC();
            for (int i = th.tid; i < e; i += NT) {
                double acc = bv[i];
                for (int c = 0; c < nv; c++) acc += Bm[i * nv + c] * rec[d.oy[lv] + c];
                rec[d.ow[lv] + i] = acc;
            }
            off += rank;
            if (th.tid == 0) { stat[HQS_STATUS + lv] = status; stat[HQS_ITER + lv] = 0; stat[HQS_NULL + lv] = nv - off; }
            HQP_SYNC();
            continue;
        }
        if (solve) {
            double *Hinv = L + l.Hinv, *T = L + l.T, *Mm = L + l.Mm, *Lc = L + l.Lc, *Bz = L + l.Bz;
            double *u = L + l.u, *g = L + l.g, *zu = L + l.zu, *tp = L + l.tp, *yp = L + l.yp, *cp = L + l.cp, *hy = L + l.hy;
            double *rr = L + l.rr, *r = L + l.r, *lam = L + l.lam, *rhs = L + l.rhs, *vv = L + l.vv, *dd = L + l.dd, *sl = L + l.sl;
            double *inw = L + l.inw, *wl = L + l.wl, *rhs2 = L + l.am;
            int status = 1, iters = 0;
            HQP_SYNC();
            for (int i = th.tid; i < nv; i += NT) yp[i] = lv > 0 ? rec[d.oy[lv - 1] + i] : 0.0;
            HQP_SYNC();
            // ---- Bz = B Z, rr = B y_prev + b
            for (int idx = th.tid; idx < e * k; idx += NT) {
                const int i = idx / k, j = idx - i * k;
                double acc = 0.0;
                for (int c = 0; c < nv; c++) acc += Bm[i * nv + c] * Z[c * nv + off + j];
                Bz[i * k + j] = acc;
            }
            for (int i = th.tid; i < e; i += NT) {
                double acc = bv[i];
                for (int c = 0; c < nv; c++) acc += Bm[i * nv + c] * yp[c];
                rr[i] = acc;
            }
            HQP_SYNC();
            // ---- H = Bz^T Bz + eps I (+ Z^T Hc Z), g = Bz^T rr (+ Z^T Hc y_prev)
            const double *Hc = d.has_cost[lv] ? rec + d.oH[lv] : nullptr;
            if (Hc) {
                for (int idx = th.tid; idx < nv * k; idx += NT) {
                    const int i = idx / k, j = idx - i * k;
                    double acc = 0.0;
                    for (int c = 0; c < nv; c++) acc += Hc[i * nv + c] * Z[c * nv + off + j];
                    HZ[i * nv + j] = acc;
                }
                for (int i = th.tid; i < nv; i += NT) {
                    double acc = 0.0;
                    for (int c = 0; c < nv; c++) acc += Hc[i * nv + c] * yp[c];
                    hy[i] = acc;
                }
            }
            HQP_SYNC();
            for (int idx = th.tid; idx < k * k; idx += NT) {
                const int i = idx / k, j = idx - i * k;
                double acc = (i == j) ? d.eps : 0.0;
                for (int c = 0; c < e; c++) acc += Bz[c * k + i] * Bz[c * k + j];
                if (Hc)
                    for (int c = 0; c < nv; c++) acc += Z[c * nv + off + i] * HZ[c * nv + j];
                Hinv[idx] = acc;
            }
            for (int i = th.tid; i < k; i += NT) {
                double acc = 0.0;
                for (int c = 0; c < e; c++) acc += Bz[c * k + i] * rr[c];
                if (Hc)
                    for (int c = 0; c < nv; c++) acc += Z[c * nv + off + i] * hy[c];
                g[i] = acc;
            }
            HQP_SYNC();
            if (!hqp_spd_inverse<NT>(th, Hinv, k, zu, tp)) status = 0;
            // ---- rows: own (soft) first, then the earlier levels' (hard, slack frozen)
            int nrows = 0;
            for (int jj = 0; jj <= lv; jj++) {
                const int j = jj == 0 ? lv : jj - 1;  // order: lv, 0, 1, .., lv - 1
                const int mj = d.m[j];
                if (mj == 0) continue;
                const double *Aj = rec + d.oA[j], *aj = rec + d.oa[j], *vj = rec + d.ov[j];
                for (int idx = th.tid; idx < mj * k; idx += NT) {
                    const int i = idx / k, c2 = idx - i * k;
                    double acc = 0.0;
                    for (int c = 0; c < nv; c++) acc += Aj[i * nv + c] * Z[c * nv + off + c2];
                    CZ[(nrows + i) * nv + c2] = acc;
                }
                for (int i = th.tid; i < mj; i += NT) {
                    double acc = -aj[i];
                    for (int c = 0; c < nv; c++) acc -= Aj[i * nv + c] * yp[c];
                    if (j != lv) acc += vj[i];
                    dd[nrows + i] = acc;
                    vv[nrows + i] = 0.0;
                    inw[nrows + i] = 0.0;
                }
                nrows += mj;
            }
            HQP_SYNC();
            // ---- u = -H^-1 g
            for (int i = th.tid; i < k; i += NT) {
                double acc = 0.0;
                for (int c = 0; c < k; c++) acc -= Hinv[i * k + c] * g[c];
                u[i] = acc;
            }
            HQP_SYNC();
            int nw = 0;
            while (status) {
                // most violated row outside the working set
                for (int row = th.tid; row < nrows; row += NT) {
                    double s = dd[row];
                    for (int c = 0; c < k; c++) s -= CZ[row * nv + c] * u[c];
                    if (row < mo) s += vv[row];
                    sl[row] = inw[row] != 0.0 ? 1.0e300 : s;
                }
                HQP_SYNC();
                // arg-min (first row of the smallest slack) in two stages: groups of kArgG rows by one thread each, then the group minima
                // by every thread -- 16 + ceil(nrows / 16) reads per thread instead of nrows by every thread (152 for the TOCABI LQP)
                const int kArgG = nrows <= 16 * kArgMax ? 16 : (nrows + kArgMax - 1) / kArgMax;
                const int ngrp = (nrows + kArgG - 1) / kArgG;
                for (int gi = th.tid; gi < ngrp; gi += NT) {
                    int bp = -1;
                    double bw = 1.0e300;
                    for (int row = gi * kArgG; row < nrows && row < (gi + 1) * kArgG; row++)
                        if (sl[row] < bw) { bw = sl[row]; bp = row; }
                    rhs2[gi] = bw;
                    rhs2[kArgMax + gi] = (double)bp;
                }
                HQP_SYNC();
                int p = -1;
                double worst = 1.0e300;
                for (int gi = 0; gi < ngrp; gi++)
                    if (rhs2[gi] < worst) { worst = rhs2[gi]; p = (int)rhs2[kArgMax + gi]; }
                if (p < 0 || !(worst < -d.tol)) break;
                const bool psoft = p < mo;
                double lam_p = 0.0;
                HQP_SYNC();
                for (int c = th.tid; c < k; c += NT) cp[c] = CZ[p * nv + c];
                HQP_SYNC();
                for (int i = th.tid; i < k; i += NT) {
                    double acc = 0.0;
                    for (int c = 0; c < k; c++) acc += Hinv[i * k + c] * cp[c];
                    tp[i] = acc;
                }
                HQP_SYNC();
                for (;;) {  // steps on row p until it enters the working set
                    if (++iters > d.max_iter) { status = 0; break; }
                    for (int a = th.tid; a < nw; a += NT) {
                        double acc = 0.0;
                        for (int c = 0; c < k; c++) acc += T[a * nv + c] * cp[c];
                        rhs[a] = acc;
                    }
                    HQP_SYNC();
                    if (nw > 0 && th.tid < kHqpW0) {
                        // r = M^-1 rhs by a fresh Cholesky factor (q <= 32), on wave 0 alone: row i of the factor per lane, column by column
                        // (left-looking: L[i][j] = (M[i][j] - sum_{c<j} L[i][c] L[j][c]) / L[j][j]), then the two triangular solves in their
                        // column-oriented form.  Between the columns only this wave has to see its own LDS writes: no workgroup barrier.
                        // (One thread doing all of it serially was O(q^3 / 6) dependent LDS round trips per active-set step.)
                        constexpr int WS_ = NT < kHqpW0 ? NT : kHqpW0;
                        for (int j = 0; j < nw; j++) {
                            for (int i = j + th.tid; i < nw; i += WS_) {
                                double s_ = Mm[i * kHqpMaxQ + j];
                                for (int c = 0; c < j; c++) s_ -= Lc[i * kHqpMaxQ + c] * Lc[j * kHqpMaxQ + c];
                                Lc[i * kHqpMaxQ + j] = s_;  // (row j: the squared pivot; the rows below it are scaled once it is known)
                            }
                            HQP_WSYNC();
                            const double pj = Lc[j * kHqpMaxQ + j];
                            const double dj_ = sqrt(pj > 1e-300 ? pj : 1e-300);
                            HQP_WSYNC();
                            for (int i = j + th.tid; i < nw; i += WS_) Lc[i * kHqpMaxQ + j] = (i == j) ? dj_ : Lc[i * kHqpMaxQ + j] / dj_;
                            HQP_WSYNC();
                        }
                        for (int i = th.tid; i < nw; i += WS_) r[i] = rhs[i];
                        HQP_WSYNC();
                        for (int c = 0; c < nw; c++) {  // L y = rhs
                            const double yc = r[c] / Lc[c * kHqpMaxQ + c];
                            HQP_WSYNC();
                            for (int i = c + th.tid; i < nw; i += WS_) r[i] = (i == c) ? yc : r[i] - Lc[i * kHqpMaxQ + c] * yc;
                            HQP_WSYNC();
                        }
                        for (int c = nw - 1; c >= 0; c--) {  // L^T r = y
                            const double xc = r[c] / Lc[c * kHqpMaxQ + c];
                            HQP_WSYNC();
                            for (int i = th.tid; i <= c; i += WS_) r[i] = (i == c) ? xc : r[i] - Lc[c * kHqpMaxQ + i] * xc;
                            HQP_WSYNC();
                        }
                    }
                    HQP_SYNC();
                    for (int i = th.tid; i < k; i += NT) {
                        double acc = tp[i];
                        for (int a = 0; a < nw; a++) acc -= r[a] * T[a * nv + i];
                        zu[i] = acc;
                    }
                    HQP_SYNC();
                    double nz = psoft ? 1.0 : 0.0, sp = dd[p] + (psoft ? vv[p] : 0.0), full = psoft ? 1.0 : 0.0;
                    for (int c = 0; c < k; c++) { nz += cp[c] * zu[c]; sp -= cp[c] * u[c]; full += cp[c] * tp[c]; }
                    // independent of the working set only if more than round-off (1e-12) of its H^-1-norm survives the projection
                    if (!(sp < 0.0)) break;  // a dual-only step (a row left the working set) has already satisfied it
                    const double t2 = nz > 1.0e-12 * full ? -sp / nz : 1.0e300;  // HQP_DEP
                    double t1 = 1.0e300;
                    int drop = -1;
                    for (int a = 0; a < nw; a++)
                        if (r[a] > 1e-14) {
                            const double ta = lam[a] / r[a];
                            if (ta < t1) { t1 = ta; drop = a; }
                        }
                    const double t = t1 < t2 ? t1 : t2;
#ifdef DWBC_HQP_TRACE
                    printf("lv %d it %d p %d soft %d nw %d nz %.3e full %.3e sp %.3e t1 %.3e t2 %.3e drop %d\n", lv, iters, p, (int)psoft, nw, nz, full, sp, t1, t2, drop);
#endif
                    if (!(t < 1.0e299)) {
                        // no step exists: the row depends on the working set and nothing can be traded against it.  Round-off
                        // on a redundant row (|slack| < 1e-6) is not a violation: the row is set aside; otherwise the level fails
                        if (-sp < 10.0 * d.tol) {  // HQP_TOL_DEP
                            HQP_SYNC();
                            if (th.tid == 0) inw[p] = 2.0;
                            HQP_SYNC();
                        } else status = 0;
                        break;
                    }
                    HQP_SYNC();
                    for (int i = th.tid; i < k; i += NT) u[i] -= t * zu[i];
                    if (th.tid == 0) {
                        if (psoft) vv[p] += t;
                        for (int a = 0; a < nw; a++) {
                            const int ja = (int)wl[a];
                            if (ja < mo) vv[ja] -= t * r[a];
                            lam[a] -= t * r[a];
                        }
                    }
                    lam_p += t;
                    HQP_SYNC();
                    if (t2 <= t1) {  // full step: row p enters
                        if (nw >= kHqpMaxQ) { status = 0; break; }
                        for (int c = th.tid; c < k; c += NT) T[nw * nv + c] = tp[c];
                        double mpp = psoft ? 1.0 : 0.0;
                        for (int c = 0; c < k; c++) mpp += cp[c] * tp[c];
                        if (th.tid == 0) {
                            for (int a = 0; a < nw; a++) { Mm[a * kHqpMaxQ + nw] = rhs[a]; Mm[nw * kHqpMaxQ + a] = rhs[a]; }
                            Mm[nw * kHqpMaxQ + nw] = mpp;
                            wl[nw] = (double)p;
                            lam[nw] = lam_p;
                            inw[p] = 1.0;
                        }
                        nw++;
                        HQP_SYNC();
                        break;
                    }
                    // partial step: the blocking row leaves the working set, row p is tried again
                    HQP_SYNC();
                    if (th.tid == 0) {
                        inw[(int)wl[drop]] = 0.0;
                        for (int a = drop; a + 1 < nw; a++) {
                            wl[a] = wl[a + 1];
                            lam[a] = lam[a + 1];
                        }
                        for (int a = 0; a < nw; a++)
                            for (int b2 = drop; b2 + 1 < nw; b2++) Mm[a * kHqpMaxQ + b2] = Mm[a * kHqpMaxQ + b2 + 1];
                        for (int a = drop; a + 1 < nw; a++)
                            for (int b2 = 0; b2 < nw; b2++) Mm[a * kHqpMaxQ + b2] = Mm[(a + 1) * kHqpMaxQ + b2];
                    }
                    HQP_SYNC();
                    for (int a = drop; a + 1 < nw; a++) {
                        for (int c = th.tid; c < k; c += NT) T[a * nv + c] = T[(a + 1) * nv + c];
                        HQP_SYNC();
                    }
                    nw--;
                    HQP_SYNC();
                }
            }
            HQP_SYNC();
            // ---- answers: y = y_prev + Z u, v_ans, w_ans = B y + b  (dwbc_hqp.cpp:388-394)
            for (int i = th.tid; i < nv; i += NT) {
                double acc = yp[i];
                if (status)
                    for (int c = 0; c < k; c++) acc += Z[i * nv + off + c] * u[c];
                rec[d.oy[lv] + i] = acc;
                hy[i] = acc;
            }
            for (int i = th.tid; i < mo; i += NT) rec[d.ov[lv] + i] = status ? vv[i] : 0.0;
            HQP_SYNC();
            for (int i = th.tid; i < e; i += NT) {
                double acc = bv[i];
                for (int c = 0; c < nv; c++) acc += Bm[i * nv + c] * hy[c];
                rec[d.ow[lv] + i] = acc;
            }
            if (th.tid == 0) { stat[HQS_STATUS + lv] = status; stat[HQS_ITER + lv] = iters; }
            HQP_SYNC();
        }
        // ---- null-space chain: Z_lv = Z_{lv-1} null(B_lv Z_{lv-1})
        off += hqp_null_update<NT>(th, Z, nv, off, Bm, e, L + l.Bz, L + l.zu);
        if (th.tid == 0) stat[HQS_NULL + lv] = nv - off;
        HQP_SYNC();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// RobotData::ConfigureLQP (src/dwbc.cpp:4304-4430) from the dump record of a solved cycle: y = [qddot (n); f_c (cd)],
//   level 0: torque limit (tlim = 200, hard-coded in the reference) / floating-base dynamics, "solved" as y = [-A^-1 B_; 0]
//   level 1: contact cones + joint acceleration limit (alim = 5) / J_C qddot = 0, cost 5 A / |A|_F on qddot
//   level 2 + i: J_task_i qddot = f*_i, same cost
// Rows are normalised (normalizeConstraintMatrix, dwbc_hqp.cpp:555-581).  Bn: RobotData::B_ of the instance.
// ------------------------------------------------------------------------------------------------------------------
struct LqpCfg {
    int n, cd, nc, n_tasks;   // system dof, contact dof of the (uniform) contact state, active contacts, task levels
    int t_dof[kMaxLevels];
    int act[kMaxActiveContacts];
    double lx[kMaxActiveContacts], ly[kMaxActiveContacts], mu[kMaxActiveContacts], muz[kMaxActiveContacts];
    int oBn;  // offset of B_ inside the dump record (DumpLayout::B, or ::G when no qdot was supplied: B_(q, 0) = G_)
    int fstar_off[kMaxLevels], fstar_total;
    double tlim, alim;
    // what the reduced variants change (ConfigureLQP_R src/dwbc.cpp:4504-4632, JACC_QP_R :3946-4122); the full model leaves the defaults
    int oNorm;         // >= 0: offset of the Frobenius norm that scales the cost (the FULL A_'s, dwbc.cpp:4533); -1: |A| of this system
    int tlim_idx;      // >= 0: this torque row is bounded by tlim_special (`tlim(tlim_size - 4) = 600`, dwbc.cpp:4552)
    double tlim_special;
    int jacc_mt;       // JACC: torque rows that carry the +-200 bound (JACC_QP_R: all but the six centroidal ones, dwbc.cpp:4096)
};

template <int NT>
DWBC_DEV void hqp_normalize_rows(Thr th, double *Mx, double *vx, int rows, int nv) {
    for (int i = th.tid; i < rows; i += NT) {
        double n2 = 0.0;
        for (int c = 0; c < nv; c++) n2 += Mx[i * nv + c] * Mx[i * nv + c];
        const double nrm = sqrt(n2);
        if (nrm > 0.0) {
            const double rn = 1.0 / nrm;
            for (int c = 0; c < nv; c++) Mx[i * nv + c] *= rn;
            vx[i] *= rn;
        }
    }
}

template <int NT>
DWBC_DEV void lqp_configure_instance(Thr th, const LqpCfg &cfg, const HqpDesc &d, const HqpIO &io, const double *dump, const io_t *fstar, int inst) {
    const int n = cfg.n, m = n - 6, cd = cfg.cd, nv = d.nv;
    const DumpLayout dl = DumpLayout::make(n);
    const double *dm = dump + (size_t)inst * dl.total;
    const double *A = dm + dl.A, *Ai = dm + dl.A_inv, *JC = dm + dl.J_C, *Bn = dm + cfg.oBn;
    const io_t *fs = fstar + (size_t)inst * cfg.fstar_total;
    double *rec = io.rec + (size_t)inst * d.rec;
    for (int idx = th.tid; idx < d.rec; idx += NT) rec[idx] = 0.0;
    HQP_SYNC();
    // ---- level 0
    {
        double *A0 = rec + d.oA[0], *a0 = rec + d.oa[0], *B0 = rec + d.oB[0], *b0 = rec + d.ob[0];
        for (int idx = th.tid; idx < m * nv; idx += NT) {
            const int i = idx / nv, c = idx - i * nv;
            const double v = c < n ? A[(6 + i) * n + c] : JC[(c - n) * n + 6 + i];  // [A_j | J_C^T_j]
            A0[i * nv + c] = v;
            A0[(m + i) * nv + c] = -v;
        }
        for (int i = th.tid; i < m; i += NT) {
            const double tl = i == cfg.tlim_idx ? cfg.tlim_special : cfg.tlim;
            a0[i] = -tl + Bn[6 + i];
            a0[m + i] = -tl - Bn[6 + i];
        }
        for (int idx = th.tid; idx < 6 * nv; idx += NT) {
            const int i = idx / nv, c = idx - i * nv;
            B0[idx] = c < n ? A[i * n + c] : JC[(c - n) * n + i];
        }
        for (int i = th.tid; i < 6; i += NT) b0[i] = Bn[i];
        HQP_SYNC();
        hqp_normalize_rows<NT>(th, A0, a0, 2 * m, nv);
        hqp_normalize_rows<NT>(th, B0, b0, 6, nv);
        double *y0 = rec + d.oy[0];
        for (int i = th.tid; i < n; i += NT) {
            double acc = 0.0;
            for (int c = 0; c < n; c++) acc -= Ai[i * n + c] * Bn[c];
            y0[i] = acc;
        }
    }
    // ---- shared cost 5 A / |A|_F on the acceleration block
    {
        double f2 = 0.0;
        if (cfg.oNorm >= 0) f2 = dm[cfg.oNorm] * dm[cfg.oNorm];
        else
            for (int idx = 0; idx < n * n; idx++) f2 += A[idx] * A[idx];
        const double sc = 5.0 / sqrt(f2);
        double *Hc = rec + d.oH[1];
        for (int idx = th.tid; idx < n * n; idx += NT) Hc[(idx / n) * nv + idx % n] = A[idx] * sc;
    }
    // ---- level 1
    {
        const int ncc = 10 * cfg.nc;
        double *A1 = rec + d.oA[1], *a1 = rec + d.oa[1], *B1 = rec + d.oB[1], *b1 = rec + d.ob[1];
        const double *Rc = dm + dl.contact_rot;
        // getContactConstraintMatrix(): C = -A_const_a A_rot; row r of contact a acts on (R_a^T f, R_a^T m)
        for (int idx = th.tid; idx < ncc * 6; idx += NT) {
            const int rr_ = idx / 6, j = idx - rr_ * 6, a = rr_ / 10, r10 = rr_ - 10 * a;
            const int ci = cfg.act[a];
            (void)ci;
            real_t w[6] = {0, 0, 0, 0, 0, 0};
            // world unit wrench e_j -> local wrench (R^T applied to the force or the moment half)
            const double *R = Rc + a * 9;
            const int h = j / 3, x = j % 3;
            for (int y = 0; y < 3; y++) w[3 * h + y] = (real_t)R[x * 3 + y];  // (R^T e_x)_y = R[x][y]
            const double v = (double)cone_row(r10, (real_t)cfg.lx[a], (real_t)cfg.ly[a], (real_t)cfg.mu[a], (real_t)cfg.muz[a], w);
            A1[rr_ * nv + n + 6 * a + j] = -v;
        }
        for (int i = th.tid; i < m; i += NT) {
            A1[(ncc + i) * nv + 6 + i] = 1.0;
            A1[(ncc + m + i) * nv + 6 + i] = -1.0;
            a1[ncc + i] = -cfg.alim;
            a1[ncc + m + i] = -cfg.alim;
        }
        for (int idx = th.tid; idx < cd * n; idx += NT) B1[(idx / n) * nv + idx % n] = JC[idx];
        (void)b1;
        HQP_SYNC();
        hqp_normalize_rows<NT>(th, A1, a1, ncc + 2 * m, nv);
        hqp_normalize_rows<NT>(th, B1, b1, cd, nv);
    }
    // ---- task levels
    for (int i = 0; i < cfg.n_tasks; i++) {
        const int t = cfg.t_dof[i];
        double *Bt = rec + d.oB[2 + i], *bt = rec + d.ob[2 + i];
        const double *Jt = dm + dl.J_task + i * kMaxTaskDof * n;
        for (int idx = th.tid; idx < t * n; idx += NT) Bt[(idx / n) * nv + idx % n] = Jt[idx];
        for (int j = th.tid; j < t; j += NT) bt[j] = -(double)fs[cfg.fstar_off[i] + j];
        HQP_SYNC();
        hqp_normalize_rows<NT>(th, Bt, bt, t, nv);
    }
    HQP_SYNC();
}

// ------------------------------------------------------------------------------------------------------------------
// RobotData::CalcSingleTaskTorqueWithJACC_QP (src/dwbc.cpp:3772-3945) for task level `level`, on the same solver.  The reference
// poses ONE QP over x = [qddot (n); tau (m); f_c (cd); s (t)]:
//     min 1/2 qddot^T A qddot + 50 |s|^2
//     s.t. A qddot - S^T tau + J_C^T f_c = -G ;  J_C qddot = 0 ;  J_i qddot = f*_i + f*_qp,i (i < level) ;  J qddot - s = f*
//          C f_c <= 0 (getContactConstraintMatrix) ;  |qddot_joint| <= 10 ;  |tau| <= 200
// tau and s are eliminated (tau = (A qddot + J_C^T f_c + G)[6:], s = J qddot - f*), leaving y = [qddot; f_c] with
//   level 0 (exact): floating-base rows of the dynamics, contact constraint, earlier tasks; its inequality rows (cones,
//                    acceleration bounds, torque bounds through the dynamics) are hard rows of level 1
//   level 1:         1/2 |10 (J qddot - f*)|^2 + 1/2 qddot^T A qddot
// Level-0 rows are normalised (scaling a hard row changes nothing); level 1 keeps the reference's weights.
// fqp_prev: per earlier level i, B x JACC_REC records ([acc n | torque m | contact 12 | f*_qp 6]) of its JACC solve.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kJaccAccLim = 10, kJaccTauLim = 200;
struct JaccPrev { const double *rec[kMaxLevels]; };
__host__ __device__ inline int jacc_rec_size(int n) { return n + (n - 6) + 12 + kMaxTaskDof; }

template <int NT>
DWBC_DEV void jacc_configure_instance(Thr th, const LqpCfg &cfg, int level, const JaccPrev &prev, const HqpDesc &d, const HqpIO &io, const double *dump,
                                      const io_t *fstar, int inst) {
    const int n = cfg.n, m = n - 6, cd = cfg.cd, nv = d.nv, ncc = 10 * cfg.nc, mt = cfg.jacc_mt;
    const DumpLayout dl = DumpLayout::make(n);
    const double *dm = dump + (size_t)inst * dl.total;
    const double *A = dm + dl.A, *JC = dm + dl.J_C, *G = dm + dl.G, *Rc = dm + dl.contact_rot;
    const io_t *fs = fstar + (size_t)inst * cfg.fstar_total;
    double *rec = io.rec + (size_t)inst * d.rec;
    for (int idx = th.tid; idx < d.rec; idx += NT) rec[idx] = 0.0;
    HQP_SYNC();
    double *A0 = rec + d.oA[0], *a0 = rec + d.oa[0], *B0 = rec + d.oB[0], *b0 = rec + d.ob[0];
    // ---- level-0 inequality rows: cones | +-qddot_joint <= 10 | +-tau <= 200
    for (int idx = th.tid; idx < ncc * 6; idx += NT) {
        const int rr_ = idx / 6, j = idx - rr_ * 6, a = rr_ / 10, r10 = rr_ - 10 * a;
        real_t w[6] = {0, 0, 0, 0, 0, 0};
        const double *R = Rc + a * 9;
        const int h = j / 3, x = j % 3;
        for (int y = 0; y < 3; y++) w[3 * h + y] = (real_t)R[x * 3 + y];
        A0[rr_ * nv + n + 6 * a + j] = -(double)cone_row(r10, (real_t)cfg.lx[a], (real_t)cfg.ly[a], (real_t)cfg.mu[a], (real_t)cfg.muz[a], w);
    }
    for (int i = th.tid; i < m; i += NT) {
        A0[(ncc + i) * nv + 6 + i] = 1.0;
        A0[(ncc + m + i) * nv + 6 + i] = -1.0;
        a0[ncc + i] = -(double)kJaccAccLim;
        a0[ncc + m + i] = -(double)kJaccAccLim;
        if (i < mt) {
            a0[ncc + 2 * m + i] = -(double)kJaccTauLim + G[6 + i];
            a0[ncc + 2 * m + mt + i] = -(double)kJaccTauLim - G[6 + i];
        }
    }
    for (int idx = th.tid; idx < mt * nv; idx += NT) {
        const int i = idx / nv, c = idx - i * nv;
        const double v = c < n ? A[(6 + i) * n + c] : JC[(c - n) * n + 6 + i];
        A0[(ncc + 2 * m + i) * nv + c] = v;
        A0[(ncc + 2 * m + mt + i) * nv + c] = -v;
    }
    // ---- level-0 equalities: floating-base dynamics | contact | earlier tasks
    for (int idx = th.tid; idx < 6 * nv; idx += NT) {
        const int i = idx / nv, c = idx - i * nv;
        B0[idx] = c < n ? A[i * n + c] : JC[(c - n) * n + i];
    }
    for (int i = th.tid; i < 6; i += NT) b0[i] = G[i];
    for (int idx = th.tid; idx < cd * n; idx += NT) B0[(6 + idx / n) * nv + idx % n] = JC[idx];
    int row = 6 + cd;
    for (int i = 0; i < level; i++) {
        const int t = cfg.t_dof[i];
        const double *Jt = dm + dl.J_task + i * kMaxTaskDof * n;
        const double *pr = prev.rec[i] + (size_t)inst * jacc_rec_size(n) + n + m + 12;
        for (int idx = th.tid; idx < t * n; idx += NT) B0[(row + idx / n) * nv + idx % n] = Jt[idx];
        for (int j = th.tid; j < t; j += NT) b0[row + j] = -((double)fs[cfg.fstar_off[i] + j] + pr[j]);
        row += t;
    }
    HQP_SYNC();
    hqp_normalize_rows<NT>(th, A0, a0, d.m[0], nv);
    hqp_normalize_rows<NT>(th, B0, b0, d.e[0], nv);
    // ---- level 1: 10 (J qddot - f*), cost A on qddot
    {
        const int t = cfg.t_dof[level];
        double *B1 = rec + d.oB[1], *b1 = rec + d.ob[1], *Hc = rec + d.oH[1];
        const double *Jt = dm + dl.J_task + level * kMaxTaskDof * n;
        for (int idx = th.tid; idx < t * n; idx += NT) B1[(idx / n) * nv + idx % n] = 10.0 * Jt[idx];
        for (int j = th.tid; j < t; j += NT) b1[j] = -10.0 * (double)fs[cfg.fstar_off[level] + j];
        for (int idx = th.tid; idx < n * n; idx += NT) Hc[(idx / n) * nv + idx % n] = A[idx];
    }
    HQP_SYNC();
}

// acc_qp_, torque_qp_, contact_qp_, f_star_qp_ (src/dwbc.cpp:3932-3942) from the level-1 answer
template <int NT>
DWBC_DEV void jacc_extract_instance(Thr th, const LqpCfg &cfg, int level, const HqpDesc &d, const HqpIO &io, const double *dump, const io_t *fstar,
                                    double *out, int *status, int inst) {
    const int n = cfg.n, m = n - 6, cd = cfg.cd;
    const DumpLayout dl = DumpLayout::make(n);
    const double *dm = dump + (size_t)inst * dl.total;
    const double *A = dm + dl.A, *JC = dm + dl.J_C, *G = dm + dl.G;
    const double *y = io.rec + (size_t)inst * d.rec + d.oy[1];
    const int *st = io.stat + (size_t)inst * HQS_COUNT;
    const int ok = st[HQS_STATUS + 0] && st[HQS_STATUS + 1];
    double *o = out + (size_t)inst * jacc_rec_size(n);
    const io_t *fs = fstar + (size_t)inst * cfg.fstar_total;
    for (int i = th.tid; i < n; i += NT) o[i] = ok ? y[i] : 0.0;
    for (int i = th.tid; i < m; i += NT) {
        double acc = G[6 + i];
        for (int c = 0; c < n; c++) acc += A[(6 + i) * n + c] * y[c];
        for (int c = 0; c < cd; c++) acc += JC[c * n + 6 + i] * y[n + c];
        o[n + i] = ok ? acc : 0.0;
    }
    for (int i = th.tid; i < 12; i += NT) o[n + m + i] = (ok && i < cd) ? y[n + i] : 0.0;
    const int t = cfg.t_dof[level];
    const double *Jt = dm + dl.J_task + level * kMaxTaskDof * n;
    for (int j = th.tid; j < kMaxTaskDof; j += NT) {
        double acc = 0.0;
        if (j < t && ok) {
            acc = -(double)fs[cfg.fstar_off[level] + j];
            for (int c = 0; c < n; c++) acc += Jt[j * n + c] * y[c];
        }
        o[n + m + 12 + j] = acc;
    }
    if (th.tid == 0) status[inst] = ok;
}

// tau = A[6:] qddot + J_C^T[6:] f_c + B_[6:]   (tests/sp_test/jacc_compare.cpp:416-418) from the last level's answer
template <int NT>
DWBC_DEV void lqp_torque_instance(Thr th, const LqpCfg &cfg, const HqpDesc &d, const HqpIO &io, const double *dump, double *tau, int inst) {
    const int n = cfg.n, m = n - 6, cd = cfg.cd;
    const DumpLayout dl = DumpLayout::make(n);
    const double *dm = dump + (size_t)inst * dl.total;
    const double *A = dm + dl.A, *JC = dm + dl.J_C, *Bn = dm + cfg.oBn;
    const double *y = io.rec + (size_t)inst * d.rec + d.oy[d.n_levels - 1];
    for (int i = th.tid; i < m; i += NT) {
        double acc = Bn[6 + i];
        for (int c = 0; c < n; c++) acc += A[(6 + i) * n + c] * y[c];
        for (int c = 0; c < cd; c++) acc += JC[c * n + 6 + i] * y[n + c];
        tau[(size_t)inst * m + i] = acc;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The reduced variants (ConfigureLQP_R src/dwbc.cpp:4504-4632, CalcSingleTaskTorqueWithJACC_QP_R :3946-4122) are the full-model
// formulations with (A_, A_inv_, J_C, G_ / B_, J_task) replaced by (A_R, A_R_inv, J_CR, G_R, J_task J_R_INV_T^T).  After a REDUCED
// cycle with the dump record on, this builds, per instance, a record in the layout of DumpLayout::make(RS) that holds exactly
// those matrices, so that lqp_configure_instance / jacc_configure_instance / *_extract / *_torque run on it unchanged (cfg.n = RS).
//   J_CR = [J_C[:, :vc], 0] (dwbc.cpp:3081-3083);  J_R_INV_T = [I_vc 0; 0 J_I_nc_inv_T] (dwbc.cpp:2976-2980)
// The record's `com` slot carries |A_|_F of the full model (the cost scale of ConfigureLQP_R).
// src[i]: task level whose Jacobian becomes slot i (the contact-chain levels, in order).
// ------------------------------------------------------------------------------------------------------------------
struct ReducedRecCfg {
    int n, vcd, cd, n_src;
    int src[kMaxLevels], t_dof[kMaxLevels];
};

template <int NT>
DWBC_DEV void reduced_record_instance(Thr th, const ReducedRecCfg &rc, const double *dump, double *rrec, int inst) {
    const int n = rc.n, vcd = rc.vcd, RS = vcd + 6, ncd = n - vcd, NCX = n - 12, RSX = kMaxReducedDof, T = kMaxTaskDof;
    const DumpLayout dl = DumpLayout::make(n), dr = DumpLayout::make(RS);
    const double *dm = dump + (size_t)inst * dl.total;
    double *o = rrec + (size_t)inst * dr.total;
    for (int idx = th.tid; idx < RS * RS; idx += NT) {
        const int i = idx / RS, j = idx - i * RS;
        o[dr.A + idx] = dm[dl.A_R + i * RSX + j];
        o[dr.A_inv + idx] = dm[dl.A_R_inv + i * RSX + j];
    }
    for (int idx = th.tid; idx < 12 * RS; idx += NT) {
        const int p = idx / RS, a = idx - p * RS;
        o[dr.J_C + idx] = (p < rc.cd && a < vcd) ? dm[dl.J_C + p * n + a] : 0.0;
    }
    for (int a = th.tid; a < RS; a += NT) { o[dr.G + a] = dm[dl.G_R + a]; o[dr.B + a] = dm[dl.G_R + a]; }
    for (int idx = th.tid; idx < kMaxActiveContacts * 9; idx += NT) o[dr.contact_rot + idx] = dm[dl.contact_rot + idx];
    for (int i = 0; i < rc.n_src; i++) {
        const double *J = dm + dl.J_task + rc.src[i] * T * n;
        for (int idx = th.tid; idx < rc.t_dof[i] * RS; idx += NT) {
            const int r = idx / RS, a = idx - r * RS;
            double v;
            if (a < vcd) v = J[r * n + a];
            else {
                v = 0.0;
                for (int c = 0; c < ncd; c++) v += J[r * n + vcd + c] * dm[dl.J_I_nc_inv_T + (a - vcd) * NCX + c];
            }
            o[dr.J_task + i * T * RS + idx] = v;
        }
    }
    if (th.tid == 0) {
        double f2 = 0.0;
        for (int idx = 0; idx < n * n; idx++) f2 += dm[dl.A + idx] * dm[dl.A + idx];
        o[dr.com] = sqrt(f2);
    }
    HQP_SYNC();
}

// ------------------------------------------------------------------------------------------------------------------
// The non-contact halves: ConfigureLQP_R_NC (src/dwbc.cpp:4634-4760) and CalcSingleTaskTorqueWithJACC_QP_R_NC (:4124-4302), for
// ONE 6-D task level on a non-contact link (the reference hard-codes ts_[1]).  Unknown: the nc_dof joint accelerations.
//   fstar_local = Ja (f* - base acceleration of the reduced answer), Ja = [[I, skew(x_link - x_pelvis)], [0, I]]
//   LQP level 0:  J_I_nc a = centroidal acceleration of the reduced answer ; |A_nn a + G_n| <= 200 ; cost 5 A_nn / |A_nn|_F
//   LQP level 1:  J_task[:, nc] a = fstar_local ; |a| <= 5 ; same cost              (rows NOT normalised, as in the reference)
//   JACC:         min 1/2 |J_I_nc a - gacc_prev|^2 + 5/2 |J_task[:, nc] a - fstar_local|^2 -- the bounds the reference prepares are
//                 deleted again before its solve (`DeleteSubjectToX`, dwbc.cpp:4277); one equality-only level, least-norm canon
// prev: per instance, the reduced answer [base 6 | chain joints | centroidal 6] (first RS doubles of a record of stride prev_stride).
// ------------------------------------------------------------------------------------------------------------------
struct NcCfg {
    int n, vcd, level, link, t, fstar_off, fstar_total;
    int prev_stride, prev_off;
};

template <int NT>
DWBC_DEV void nc_task_local(Thr th, const NcCfg &c, const double *dm, const DumpLayout &dl, const io_t *fs, const double *prev, double *out6) {
    // uniform 6-vector, computed by every thread
    const double *p0 = dm + dl.link_p, *pl = dm + dl.link_p + c.link * 3;
    const double dx = pl[0] - p0[0], dy = pl[1] - p0[1], dz = pl[2] - p0[2];
    double e[6];
    for (int j = 0; j < 6; j++) e[j] = (j < c.t ? (double)fs[c.fstar_off + j] : 0.0) - prev[j];
    out6[0] = e[0] + (-dz * e[4] + dy * e[5]);
    out6[1] = e[1] + (dz * e[3] - dx * e[5]);
    out6[2] = e[2] + (-dy * e[3] + dx * e[4]);
    out6[3] = e[3]; out6[4] = e[4]; out6[5] = e[5];
}

template <int NT>
DWBC_DEV void lqp_nc_configure_instance(Thr th, const NcCfg &c, const HqpDesc &d, const HqpIO &io, const double *dump, const io_t *fstar,
                                        const double *prev_all, int inst) {
    const int n = c.n, vcd = c.vcd, ncd = n - vcd, NCX = n - 12, nv = d.nv, T = kMaxTaskDof, RS = vcd + 6;
    const DumpLayout dl = DumpLayout::make(n);
    const double *dm = dump + (size_t)inst * dl.total;
    const double *A = dm + dl.A, *G = dm + dl.G, *JI = dm + dl.J_I_nc, *Jt = dm + dl.J_task + c.level * T * n;
    const io_t *fs = fstar + (size_t)inst * c.fstar_total;
    const double *prev = prev_all + (size_t)inst * c.prev_stride + c.prev_off;
    double *rec = io.rec + (size_t)inst * d.rec;
    for (int idx = th.tid; idx < d.rec; idx += NT) rec[idx] = 0.0;
    HQP_SYNC();
    double f2 = 0.0;
    for (int i = 0; i < ncd; i++)
        for (int j = 0; j < ncd; j++) f2 += A[(vcd + i) * n + vcd + j] * A[(vcd + i) * n + vcd + j];
    const double sc = 5.0 / sqrt(f2);
    double *A0 = rec + d.oA[0], *a0 = rec + d.oa[0], *B0 = rec + d.oB[0], *b0 = rec + d.ob[0], *Hc = rec + d.oH[0];
    double *A1 = rec + d.oA[1], *a1 = rec + d.oa[1], *B1 = rec + d.oB[1], *b1 = rec + d.ob[1];
    for (int idx = th.tid; idx < ncd * ncd; idx += NT) {
        const int i = idx / ncd, j = idx - i * ncd;
        const double v = A[(vcd + i) * n + vcd + j];
        A0[i * nv + j] = v;
        A0[(ncd + i) * nv + j] = -v;
        Hc[i * nv + j] = v * sc;
    }
    for (int i = th.tid; i < ncd; i += NT) {
        a0[i] = -200.0 + G[vcd + i];
        a0[ncd + i] = -200.0 - G[vcd + i];
        A1[i * nv + i] = 1.0;
        A1[(ncd + i) * nv + i] = -1.0;
        a1[i] = -5.0;
        a1[ncd + i] = -5.0;
    }
    for (int idx = th.tid; idx < 6 * ncd; idx += NT) {
        const int r = idx / ncd, j = idx - r * ncd;
        B0[r * nv + j] = JI[r * NCX + j];
        B1[r * nv + j] = r < c.t ? Jt[r * n + vcd + j] : 0.0;
    }
    double fl[6];
    nc_task_local<NT>(th, c, dm, dl, fs, prev, fl);
    for (int r = th.tid; r < 6; r += NT) { b0[r] = -prev[RS - 6 + r]; b1[r] = -fl[r]; }
    HQP_SYNC();
}

__host__ __device__ inline int jacc_nc_rec_size(int ncd) { return 2 * ncd + 6 + kMaxTaskDof; }

template <int NT>
DWBC_DEV void jacc_nc_configure_instance(Thr th, const NcCfg &c, const HqpDesc &d, const HqpIO &io, const double *dump, const io_t *fstar,
                                         const double *prev_all, int inst) {
    const int n = c.n, vcd = c.vcd, ncd = n - vcd, NCX = n - 12, nv = d.nv, T = kMaxTaskDof, RS = vcd + 6;
    const DumpLayout dl = DumpLayout::make(n);
    const double *dm = dump + (size_t)inst * dl.total;
    const double *JI = dm + dl.J_I_nc, *Jt = dm + dl.J_task + c.level * T * n;
    const io_t *fs = fstar + (size_t)inst * c.fstar_total;
    const double *prev = prev_all + (size_t)inst * c.prev_stride + c.prev_off;
    double *rec = io.rec + (size_t)inst * d.rec;
    for (int idx = th.tid; idx < d.rec; idx += NT) rec[idx] = 0.0;
    HQP_SYNC();
    const double w = sqrt(5.0);  // `5 * Identity` on the task slack (dwbc.cpp:4163)
    double *B0 = rec + d.oB[0], *b0 = rec + d.ob[0];
    for (int idx = th.tid; idx < (6 + c.t) * ncd; idx += NT) {
        const int r = idx / ncd, j = idx - r * ncd;
        B0[r * nv + j] = r < 6 ? JI[r * NCX + j] : w * Jt[(r - 6) * n + vcd + j];
    }
    double fl[6];
    nc_task_local<NT>(th, c, dm, dl, fs, prev, fl);
    for (int r = th.tid; r < 6 + c.t; r += NT) b0[r] = r < 6 ? -prev[RS - 6 + r] : -w * fl[r - 6];
    HQP_SYNC();
}

// acc_qp_ (nc_dof), torque_qp_ = A_nn a + G_n, gacc_qp_ = J_I_nc a - gacc_prev, f_star_qp_ = J_task[:, nc] a - fstar_local (dwbc.cpp:4296-4299)
template <int NT>
DWBC_DEV void jacc_nc_extract_instance(Thr th, const NcCfg &c, const HqpDesc &d, const HqpIO &io, const double *dump, const io_t *fstar,
                                       const double *prev_all, double *out, int *status, int inst) {
    const int n = c.n, vcd = c.vcd, ncd = n - vcd, NCX = n - 12, T = kMaxTaskDof, RS = vcd + 6;
    const DumpLayout dl = DumpLayout::make(n);
    const double *dm = dump + (size_t)inst * dl.total;
    const double *A = dm + dl.A, *G = dm + dl.G, *JI = dm + dl.J_I_nc, *Jt = dm + dl.J_task + c.level * T * n;
    const io_t *fs = fstar + (size_t)inst * c.fstar_total;
    const double *prev = prev_all + (size_t)inst * c.prev_stride + c.prev_off;
    const double *y = io.rec + (size_t)inst * d.rec + d.oy[0];
    const int ok = io.stat[(size_t)inst * HQS_COUNT + HQS_STATUS];
    double *o = out + (size_t)inst * jacc_nc_rec_size(ncd);
    double fl[6];
    nc_task_local<NT>(th, c, dm, dl, fs, prev, fl);
    for (int i = th.tid; i < ncd; i += NT) {
        o[i] = ok ? y[i] : 0.0;
        double acc = G[vcd + i];
        for (int j = 0; j < ncd; j++) acc += A[(vcd + i) * n + vcd + j] * y[j];
        o[ncd + i] = ok ? acc : 0.0;
    }
    for (int r = th.tid; r < 6; r += NT) {
        double g = -prev[RS - 6 + r], f = -fl[r];
        for (int j = 0; j < ncd; j++) { g += JI[r * NCX + j] * y[j]; if (r < c.t) f += Jt[r * n + vcd + j] * y[j]; }
        o[2 * ncd + r] = ok ? g : 0.0;
        o[2 * ncd + 6 + r] = (ok && r < c.t) ? f : 0.0;
    }
    if (th.tid == 0) status[inst] = ok;
}

}  // namespace dwbc

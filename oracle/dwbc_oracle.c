/*
 * dwbc_oracle.c -- plain-C CPU restatement of libdwbc's per-cycle OSF/HQP torque solve.
 *
 * TEST INFRASTRUCTURE ONLY (see dwbc_oracle.h).  Every function cites the reference file:line it
 * follows (paths relative to /root/reference).  Third-party arithmetic that is not vendored in the
 * reference (RBDL, Eigen, qpOASES: all unpinned forks, see oracle/README.md) is restated from its
 * published algorithm and marked [ext].
 *
 * Dense algebra is written "as the reference writes it" (N_C and A_inv*N_C are materialised, J*A^-1*N_c
 * is recomputed per task, ...) because this file is also the timed CPU baseline.
 */
#include "dwbc_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define COD_THRESHOLD 1.0e-6 /* include/dwbc_wbd.h:10 */
#define GRAV 9.81
#define QP_SCALE 1.0e4
#define QP_TOL 1.0e-9
#define QP_ZERO_ROW 1.0e-9 /* rows with a smaller norm are treated as 0 . x <= b */
#define QP_FEAS_TOL 1.0e-7

int orc_sizeof_model(void) { return (int)sizeof(orc_model); }
int orc_sizeof_setup(void) { return (int)sizeof(orc_setup); }
int orc_sizeof_out(void) { return (int)sizeof(orc_out); }
int orc_sizeof_debug(void) { return (int)sizeof(orc_debug); }

/* ------------------------------------------------------------------------------------------ */
/* small dense helpers (row-major)                                                            */
/* ------------------------------------------------------------------------------------------ */
static void mm(double *C, int ldc, const double *A, int lda, const double *B, int ldb, int m, int k, int n) {
    for (int i = 0; i < m; i++) {
        double *c = C + i * ldc;
        for (int j = 0; j < n; j++) c[j] = 0.0;
        for (int p = 0; p < k; p++) {
            double a = A[i * lda + p];
            const double *b = B + p * ldb;
            for (int j = 0; j < n; j++) c[j] += a * b[j];
        }
    }
}
/* C = A * B^T */
static void mmt(double *C, int ldc, const double *A, int lda, const double *B, int ldb, int m, int k, int n) {
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            double s = 0.0;
            for (int p = 0; p < k; p++) s += A[i * lda + p] * B[j * ldb + p];
            C[i * ldc + j] = s;
        }
}
/* C = A^T * B, A is k x m */
static void mtm(double *C, int ldc, const double *A, int lda, const double *B, int ldb, int m, int k, int n) {
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) C[i * ldc + j] = 0.0;
    for (int p = 0; p < k; p++)
        for (int i = 0; i < m; i++) {
            double a = A[p * lda + i];
            for (int j = 0; j < n; j++) C[i * ldc + j] += a * B[p * ldb + j];
        }
}
static void mv(double *y, const double *A, int lda, const double *x, int m, int n) {
    for (int i = 0; i < m; i++) {
        double s = 0.0;
        for (int j = 0; j < n; j++) s += A[i * lda + j] * x[j];
        y[i] = s;
    }
}

/* A^-1 via LL^T, the way Eigen's llt().solve(I) does it (src/dwbc.cpp:307) [ext].  returns 0 if not SPD */
static int chol_inverse(const double *A, int lda, int n, double *Ai, int ldi) {
    double L[ORC_MAXN * ORC_MAXN];
    for (int j = 0; j < n; j++) {
        double d = A[j * lda + j];
        for (int k = 0; k < j; k++) d -= L[j * n + k] * L[j * n + k];
        if (!(d > 0.0)) return 0;
        d = sqrt(d);
        L[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * lda + j];
            for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = s / d;
        }
    }
    for (int c = 0; c < n; c++) {
        double y[ORC_MAXN];
        for (int i = 0; i < n; i++) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = 0; k < i; k++) s -= L[i * n + k] * y[k];
            y[i] = s / L[i * n + i];
        }
        for (int i = n - 1; i >= 0; i--) {
            double s = y[i];
            for (int k = i + 1; k < n; k++) s -= L[k * n + i] * y[k];
            y[i] = s / L[i * n + i];
        }
        for (int i = 0; i < n; i++) Ai[i * ldi + c] = y[i];
    }
    return 1;
}

/* general inverse by LU with partial pivoting (Eigen MatrixXd::inverse(), src/wbd.cpp:115,128,210) [ext] */
static int lu_inverse(const double *A, int lda, int n, double *Ai, int ldi) {
    double M[ORC_MAXC * 2 * ORC_MAXC];
    int w = 2 * n;
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) M[i * w + j] = A[i * lda + j];
        for (int j = 0; j < n; j++) M[i * w + n + j] = (i == j) ? 1.0 : 0.0;
    }
    for (int c = 0; c < n; c++) {
        int p = c;
        double best = fabs(M[c * w + c]);
        for (int i = c + 1; i < n; i++)
            if (fabs(M[i * w + c]) > best) { best = fabs(M[i * w + c]); p = i; }
        if (best == 0.0) return 0;
        if (p != c)
            for (int j = 0; j < w; j++) { double t = M[c * w + j]; M[c * w + j] = M[p * w + j]; M[p * w + j] = t; }
        double inv = 1.0 / M[c * w + c];
        for (int j = 0; j < w; j++) M[c * w + j] *= inv;
        for (int i = 0; i < n; i++) {
            if (i == c) continue;
            double f = M[i * w + c];
            if (f == 0.0) continue;
            for (int j = 0; j < w; j++) M[i * w + j] -= f * M[c * w + j];
        }
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Ai[i * ldi + j] = M[i * w + n + j];
    return 1;
}

/* Householder QR with column pivoting.  A (m x n, ld n) is overwritten by R; Q (m x m) explicit.
 * piv[j] = original column now in position j.  returns min(m,n). */
static int qrcp(double *A, int m, int n, double *Q, int *piv, int pivoting) {
    double cn[ORC_MAXR];
    for (int i = 0; i < m; i++)
        for (int j = 0; j < m; j++) Q[i * m + j] = (i == j) ? 1.0 : 0.0;
    for (int j = 0; j < n; j++) piv[j] = j;
    int steps = m < n ? m : n;
    for (int s = 0; s < steps; s++) {
        if (pivoting) {
            int best = s;
            double bn = -1.0;
            for (int j = s; j < n; j++) {
                double t = 0.0;
                for (int i = s; i < m; i++) t += A[i * n + j] * A[i * n + j];
                cn[j] = t;
                if (t > bn) { bn = t; best = j; }
            }
            if (best != s) {
                for (int i = 0; i < m; i++) { double t = A[i * n + s]; A[i * n + s] = A[i * n + best]; A[i * n + best] = t; }
                int t = piv[s]; piv[s] = piv[best]; piv[best] = t;
            }
        }
        double nrm = 0.0;
        for (int i = s; i < m; i++) nrm += A[i * n + s] * A[i * n + s];
        nrm = sqrt(nrm);
        if (nrm == 0.0) continue;
        double alpha = A[s * n + s] > 0 ? -nrm : nrm;
        double v[ORC_MAXR];
        double vn = 0.0;
        for (int i = s; i < m; i++) { v[i] = A[i * n + s]; }
        v[s] -= alpha;
        for (int i = s; i < m; i++) vn += v[i] * v[i];
        if (vn == 0.0) continue;
        double beta = 2.0 / vn;
        for (int j = s; j < n; j++) {
            double d = 0.0;
            for (int i = s; i < m; i++) d += v[i] * A[i * n + j];
            d *= beta;
            for (int i = s; i < m; i++) A[i * n + j] -= d * v[i];
        }
        for (int i = 0; i < m; i++) { /* Q = Q * H */
            double d = 0.0;
            for (int k = s; k < m; k++) d += Q[i * m + k] * v[k];
            d *= beta;
            for (int k = s; k < m; k++) Q[i * m + k] -= d * v[k];
        }
        for (int i = s + 1; i < m; i++) A[i * n + s] = 0.0;
    }
    return steps;
}

/* Moore-Penrose pseudo-inverse through a complete orthogonal decomposition with Eigen's threshold rule
 * (rank = #{|R_ii| > thr*max|R_ii|}) and V2 = rows [rank:] of Q^T  -- src/wbd.cpp:5-53 [ext: Eigen COD].
 * M rows x cols (ld cols); pinv cols x rows (ld rows); V2 (rows-rank) x rows?  NOTE: the reference takes
 * V2 from householderQ() of a square matrix, so V2 is (rows-rank) x rows (ld rows). */
int orc_pinv_cod(const double *M, int rows, int cols, double thr, double *pinv, double *V2, int *rank_out) {
    double R[ORC_MAXM * ORC_MAXM], Q[ORC_MAXM * ORC_MAXM];
    int piv[ORC_MAXM];
    memcpy(R, M, sizeof(double) * rows * cols);
    int steps = qrcp(R, rows, cols, Q, piv, 1);
    double maxp = 0.0;
    for (int i = 0; i < steps; i++)
        if (fabs(R[i * cols + i]) > maxp) maxp = fabs(R[i * cols + i]);
    int rank = 0;
    for (int i = 0; i < steps; i++)
        if (fabs(R[i * cols + i]) > thr * maxp) rank++;
    if (rank_out) *rank_out = rank;
    for (int i = 0; i < cols * rows; i++) pinv[i] = 0.0;
    if (rank > 0) {
        /* R1 = R[:rank,:] (rank x cols).  QR of R1^T (cols x rank) = Qz Tz ; R1^+ = Qz Tz^-T */
        double R1t[ORC_MAXM * ORC_MAXM], Qz[ORC_MAXM * ORC_MAXM];
        int pz[ORC_MAXM];
        for (int i = 0; i < cols; i++)
            for (int j = 0; j < rank; j++) R1t[i * rank + j] = R[j * cols + i];
        qrcp(R1t, cols, rank, Qz, pz, 0);
        /* X = Tz^-T (rank x rank): solve Tz^T X = I, Tz upper => Tz^T lower */
        double X[ORC_MAXM * ORC_MAXM];
        for (int c = 0; c < rank; c++)
            for (int i = 0; i < rank; i++) {
                double s = (i == c) ? 1.0 : 0.0;
                for (int k = 0; k < i; k++) s -= R1t[k * rank + i] * X[k * rank + c];
                X[i * rank + c] = s / R1t[i * rank + i];
            }
        /* R1p = Qz[:, :rank] * X   (cols x rank) ; pinv = P * R1p * Q[:, :rank]^T */
        double R1p[ORC_MAXM * ORC_MAXM];
        for (int i = 0; i < cols; i++)
            for (int j = 0; j < rank; j++) {
                double s = 0.0;
                for (int k = 0; k < rank; k++) s += Qz[i * cols + k] * X[k * rank + j];
                R1p[i * rank + j] = s;
            }
        for (int i = 0; i < cols; i++)
            for (int j = 0; j < rows; j++) {
                double s = 0.0;
                for (int k = 0; k < rank; k++) s += R1p[i * rank + k] * Q[j * rows + k];
                pinv[piv[i] * rows + j] = s;
            }
    }
    if (V2)
        for (int i = rank; i < rows; i++)
            for (int j = 0; j < rows; j++) V2[(i - rank) * rows + j] = Q[j * rows + i];
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* QP  lexmin(1/2|x[:t]|^2, 1/2|x[t:]|^2) s.t. A x <= ub  (stands in for src/qp_wrapper.cpp:192-380)  */
/* ------------------------------------------------------------------------------------------ */
/* least squares r = argmin |N r - g|, z = g - N r, N is n x q given as q columns in Ncols[q][n] */
static void ls_project(const double *Ncols, int q, int n, const double *g, double *r, double *z) {
    if (q == 0) {
        for (int i = 0; i < n; i++) z[i] = g[i];
        return;
    }
    double M[ORC_MAXV * ORC_MAXV], Q[ORC_MAXV * ORC_MAXV];
    int piv[ORC_MAXV];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < q; j++) M[i * q + j] = Ncols[j * n + i];
    qrcp(M, n, q, Q, piv, 0);
    double y[ORC_MAXV];
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int k = 0; k < n; k++) s += Q[k * n + i] * g[k];
        y[i] = s;
    }
    for (int i = q - 1; i >= 0; i--) {
        double s = y[i];
        for (int k = i + 1; k < q; k++) s -= M[i * q + k] * r[k];
        r[i] = s / M[i * q + i];
    }
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int k = q; k < n; k++) s += Q[i * n + k] * y[k];
        z[i] = s;
    }
}

/* Goldfarb-Idnani dual active set on min 1/2|x|^2 s.t. G x <= b (normals n_i = -g_i) */
static int gi_least_distance(const double *G, const double *b, int m, int n, int max_iter, double *x, int *act,
                             int *nact_out, double *u, int *iters, double tol) {
    double gnorm[ORC_MAXR];
    double Ncols[ORC_MAXV * ORC_MAXV];
    int q = 0, it = 0;
    for (int i = 0; i < m; i++) {
        double s = 0.0;
        for (int j = 0; j < n; j++) s += G[i * n + j] * G[i * n + j];
        gnorm[i] = sqrt(s);
        if (gnorm[i] < QP_ZERO_ROW) gnorm[i] = 1.0; /* numerically zero row = the constraint 0 <= b: slack taken as is */
    }
    for (int j = 0; j < n; j++) x[j] = 0.0;
    for (;;) {
        int p = -1;
        double worst = -tol;
        for (int i = 0; i < m; i++) {
            int in = 0;
            for (int a = 0; a < q; a++) if (act[a] == i) in = 1;
            if (in) continue;
            double s = b[i];
            for (int j = 0; j < n; j++) s -= G[i * n + j] * x[j];
            s /= gnorm[i];
            if (s < worst) { worst = s; p = i; }
        }
        if (p < 0) { *nact_out = q; *iters = it; return 1; }
        double up = 0.0;
        for (;;) {
            if (++it > max_iter) { *nact_out = q; *iters = it; return 0; }
            double gp[ORC_MAXV], r[ORC_MAXV], z[ORC_MAXV];
            for (int j = 0; j < n; j++) gp[j] = -G[p * n + j];
            ls_project(Ncols, q, n, gp, r, z);
            double zn = 0.0, zg = 0.0, rmax = 1.0;
            for (int j = 0; j < n; j++) { zn += z[j] * z[j]; zg += z[j] * gp[j]; }
            zn = sqrt(zn);
            for (int a = 0; a < q; a++) if (fabs(r[a]) > rmax) rmax = fabs(r[a]);
            double t1 = INFINITY, t2 = INFINITY;
            int l = -1;
            for (int a = 0; a < q; a++)
                if (r[a] > 1e-13 * rmax) {
                    double tj = u[a] / r[a];
                    if (tj < t1) { t1 = tj; l = a; }
                }
            double sp = b[p];
            for (int j = 0; j < n; j++) sp -= G[p * n + j] * x[j];
            if (zn > 1e-10 * gnorm[p]) t2 = -sp / zg;
            double t = t1 < t2 ? t1 : t2;
            if (!isfinite(t)) { *nact_out = q; *iters = it; return 0; }
            int full = isfinite(t2) && t2 <= t1;
            if (isfinite(t2))
                for (int j = 0; j < n; j++) x[j] += t * z[j];
            for (int a = 0; a < q; a++) u[a] -= t * r[a];
            up += t;
            if (full) {
                for (int j = 0; j < n; j++) Ncols[q * n + j] = gp[j];
                act[q] = p;
                u[q] = up;
                q++;
                break;
            }
            for (int a = l; a < q - 1; a++) {
                act[a] = act[a + 1];
                u[a] = u[a + 1];
                memcpy(Ncols + a * n, Ncols + (a + 1) * n, sizeof(double) * n);
            }
            q--;
        }
    }
}

static void spd_solve_small(double *M, int n, double *rhs) { /* Cholesky, in place */
    for (int j = 0; j < n; j++) {
        double d = M[j * n + j];
        for (int k = 0; k < j; k++) d -= M[j * n + k] * M[j * n + k];
        d = sqrt(d);
        M[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = M[i * n + j];
            for (int k = 0; k < j; k++) s -= M[i * n + k] * M[j * n + k];
            M[i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; i++) {
        double s = rhs[i];
        for (int k = 0; k < i; k++) s -= M[i * n + k] * rhs[k];
        rhs[i] = s / M[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = rhs[i];
        for (int k = i + 1; k < n; k++) s -= M[k * n + i] * rhs[k];
        rhs[i] = s / M[i * n + i];
    }
}

/* exact lexicographic equality-constrained solve on the final working set (DESIGN.md "QP canon") */
static void lex_eqp(const double *Ad, const double *Ac, const double *b, int q, int t, int k, double *d, double *c) {
    for (int i = 0; i < t; i++) d[i] = 0.0;
    for (int i = 0; i < k; i++) c[i] = 0.0;
    if (q == 0) return;
    double R[ORC_MAXV * ORC_MAXV], Q[ORC_MAXV * ORC_MAXV];
    int piv[ORC_MAXV];
    memcpy(R, Ac, sizeof(double) * q * k);
    int steps = qrcp(R, q, k, Q, piv, 1);
    int rho = 0;
    double d0 = steps > 0 ? fabs(R[0]) : 0.0;
    for (int i = 0; i < steps; i++)
        if (d0 > 0 && fabs(R[i * k + i]) > 1e-9 * d0) rho++;
    double Ab[ORC_MAXV * ORC_MAXV], bb[ORC_MAXV];
    for (int i = 0; i < q; i++) {
        for (int j = 0; j < t; j++) {
            double s = 0.0;
            for (int r = 0; r < q; r++) s += Q[r * q + i] * Ad[r * t + j];
            Ab[i * t + j] = s;
        }
        double s = 0.0;
        for (int r = 0; r < q; r++) s += Q[r * q + i] * b[r];
        bb[i] = s;
    }
    int q2 = q - rho;
    if (q2 > 0) {
        double M[ORC_MAXV * ORC_MAXV], y[ORC_MAXV];
        for (int i = 0; i < q2; i++) {
            for (int j = 0; j < q2; j++) {
                double s = 0.0;
                for (int r = 0; r < t; r++) s += Ab[(rho + i) * t + r] * Ab[(rho + j) * t + r];
                M[i * q2 + j] = s;
            }
            y[i] = bb[rho + i];
        }
        spd_solve_small(M, q2, y);
        for (int j = 0; j < t; j++) {
            double s = 0.0;
            for (int i = 0; i < q2; i++) s += Ab[(rho + i) * t + j] * y[i];
            d[j] = s;
        }
    }
    if (rho > 0) {
        double f[ORC_MAXV], M[ORC_MAXV * ORC_MAXV];
        for (int i = 0; i < rho; i++) {
            double s = bb[i];
            for (int j = 0; j < t; j++) s -= Ab[i * t + j] * d[j];
            f[i] = s;
            for (int j = 0; j < rho; j++) {
                double a = 0.0;
                for (int r = 0; r < k; r++) a += R[i * k + r] * R[j * k + r];
                M[i * rho + j] = a;
            }
        }
        spd_solve_small(M, rho, f);
        for (int r = 0; r < k; r++) {
            double s = 0.0;
            for (int i = 0; i < rho; i++) s += R[i * k + r] * f[i];
            c[piv[r]] = s;
        }
    }
}

/* tol: violation (slack / |row|) below which a row counts as satisfied while the working set is searched.  QP_TOL for the task
 * QPs.  The contact redistribution QP starts from the point the last task QP handed over, and that point was ACCEPTED with the
 * tolerance QP_FEAS_TOL (rule 2 of the canon): re-examining its active rows at 1e-9 makes the redistribution chase the round-off
 * of the hand-over (and fail on a degenerate vertex), so it searches with QP_FEAS_TOL -- canon rule 5, DESIGN.md. */
int orc_solve_qp_tol(const double *A, const double *ub, int rows, int nv, int t, int max_iter, double *x, int *act,
                     int *nact, int *iters, double tol);
int orc_solve_qp(const double *A, const double *ub, int rows, int nv, int t, int max_iter, double *x, int *act,
                 int *nact, int *iters) {
    return orc_solve_qp_tol(A, ub, rows, nv, t, max_iter, x, act, nact, iters, QP_TOL);
}
int orc_solve_qp_tol(const double *A, const double *ub, int rows, int nv, int t, int max_iter, double *x, int *act,
                     int *nact, int *iters, double tol) {
    double G[ORC_MAXR * ORC_MAXV], u[ORC_MAXV], xh[ORC_MAXV];
    int k = nv - t, q = 0;
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < nv; j++) G[i * nv + j] = A[i * nv + j] * (j >= t ? QP_SCALE : 1.0);
    int st = gi_least_distance(G, ub, rows, nv, max_iter, xh, act, &q, u, iters, tol);
    *nact = q;
    if (!st) {
        for (int j = 0; j < nv; j++) x[j] = 0.0;
        return 0;
    }
    /* DESIGN.md "QP canon": lexicographic least-norm point on the working set if it is feasible, else the
     * Tikhonov point on the working set (near-singular contact blocks make the lexicographic point blow up). */
    int use_tikhonov = !(k > 0 && t > 0);
    if (!use_tikhonov) {
        double Ad[ORC_MAXV * ORC_MAXV], Ac[ORC_MAXV * ORC_MAXV], b[ORC_MAXV];
        for (int a = 0; a < q; a++) {
            for (int j = 0; j < t; j++) Ad[a * t + j] = A[act[a] * nv + j];
            for (int j = 0; j < k; j++) Ac[a * k + j] = A[act[a] * nv + t + j];
            b[a] = ub[act[a]];
        }
        lex_eqp(Ad, Ac, b, q, t, k, x, x + t);
        double worst = 0.0;
        for (int i = 0; i < rows; i++) {
            double sl = ub[i], nr = 0.0;
            for (int j = 0; j < nv; j++) { sl -= A[i * nv + j] * x[j]; nr += A[i * nv + j] * A[i * nv + j]; }
            nr = sqrt(nr);
            if (nr < QP_ZERO_ROW) nr = 1.0;
            if (sl / nr < worst) worst = sl / nr;
        }
        if (worst < -QP_FEAS_TOL) use_tikhonov = 1;
    }
    if (use_tikhonov) {
        if (q > 0) { /* least-norm point of N^T x = b on the final working set, by pivoted QR of N (nv x q) */
            double Nq[ORC_MAXV * ORC_MAXV], Q[ORC_MAXV * ORC_MAXV], y[ORC_MAXV];
            int piv[ORC_MAXV];
            for (int i = 0; i < nv; i++)
                for (int a = 0; a < q; a++) Nq[i * q + a] = G[act[a] * nv + i];
            qrcp(Nq, nv, q, Q, piv, 1);
            for (int a = 0; a < q; a++) { /* R^T y = b[piv] */
                double s = ub[act[piv[a]]];
                for (int c = 0; c < a; c++) s -= Nq[c * q + a] * y[c];
                y[a] = s / Nq[a * q + a];
            }
            for (int j = 0; j < nv; j++) {
                double s = 0.0;
                for (int a = 0; a < q; a++) s += Q[j * nv + a] * y[a];
                xh[j] = s;
            }
        }
        for (int j = 0; j < nv; j++) x[j] = xh[j] * (j >= t ? QP_SCALE : 1.0);
    }
    /* report active rows sorted */
    for (int a = 1; a < q; a++) {
        int v = act[a], b2 = a - 1;
        while (b2 >= 0 && act[b2] > v) { act[b2 + 1] = act[b2]; b2--; }
        act[b2 + 1] = v;
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* rigid-body kinematics / dynamics   [ext: RBDL]                                              */
/* ------------------------------------------------------------------------------------------ */
static void m3mul(double *C, const double *A, const double *B) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
static void m3v(double *y, const double *A, const double *x) {
    for (int i = 0; i < 3; i++) y[i] = A[i * 3] * x[0] + A[i * 3 + 1] * x[1] + A[i * 3 + 2] * x[2];
}
static void cross3(double *c, const double *a, const double *b) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
static void skew3(double *S, const double *v) {
    S[0] = 0; S[1] = -v[2]; S[2] = v[1];
    S[3] = v[2]; S[4] = 0; S[5] = -v[0];
    S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}
static void quat_to_R(double *R, double x, double y, double z, double w) {
    R[0] = 1 - 2 * y * y - 2 * z * z; R[1] = 2 * x * y - 2 * w * z; R[2] = 2 * x * z + 2 * w * y;
    R[3] = 2 * x * y + 2 * w * z; R[4] = 1 - 2 * x * x - 2 * z * z; R[5] = 2 * y * z - 2 * w * x;
    R[6] = 2 * x * z - 2 * w * y; R[7] = 2 * y * z + 2 * w * x; R[8] = 1 - 2 * x * x - 2 * y * y;
}
static void axis_angle_R(double *R, const double *a, double q) {
    double K[9], KK[9], s = sin(q), c = 1 - cos(q);
    skew3(K, a);
    m3mul(KK, K, K);
    for (int i = 0; i < 9; i++) R[i] = s * K[i] + c * KK[i];
    R[0] += 1; R[4] += 1; R[8] += 1;
}

/* UpdateKinematicsCustom + Link::UpdatePos (src/dwbc.cpp:304, src/link.cpp:76-96): body->world R, origin p */
static void forward_kinematics(const orc_model *m, const double *q, double (*R)[9], double (*p)[3]) {
    quat_to_R(R[0], q[3], q[4], q[5], q[m->ndof]);
    p[0][0] = q[0]; p[0][1] = q[1]; p[0][2] = q[2];
    for (int i = 1; i < m->nb; i++) {
        int par = m->parent[i];
        double Rj[9], T[9], t[3];
        axis_angle_R(Rj, m->axis[i], q[6 + i - 1]);
        m3mul(T, m->R_T[i], Rj);
        m3mul(R[i], R[par], T);
        m3v(t, R[par], m->p_T[i]);
        for (int k = 0; k < 3; k++) p[i][k] = p[par][k] + t[k];
    }
}

static void spatial_inertia(double *I6, double mass, const double *c, const double *I) {
    double C[9], CCt[9];
    skew3(C, c);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) CCt[i * 3 + j] = C[i * 3] * C[j * 3] + C[i * 3 + 1] * C[j * 3 + 1] + C[i * 3 + 2] * C[j * 3 + 2];
    memset(I6, 0, 36 * sizeof(double));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            I6[i * 6 + j] = I[i * 3 + j] + mass * CCt[i * 3 + j];
            I6[i * 6 + 3 + j] = mass * C[i * 3 + j];
            I6[(3 + i) * 6 + j] = mass * C[j * 3 + i];
        }
    I6[21] = I6[28] = I6[35] = mass;
}
static void xform(double *X, const double *E, const double *r) { /* [E 0; -E r~ E] */
    double S[9], ES[9];
    skew3(S, r);
    m3mul(ES, E, S);
    memset(X, 0, 36 * sizeof(double));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            X[i * 6 + j] = E[i * 3 + j];
            X[(3 + i) * 6 + 3 + j] = E[i * 3 + j];
            X[(3 + i) * 6 + j] = -ES[i * 3 + j];
        }
}

/* CompositeRigidBodyAlgorithm (src/dwbc.cpp:305) [ext: Featherstone CRBA in body coordinates] */
static void crba(const orc_model *m, const double *q, double *A, int n) {
    static const int NBMAX = ORC_MAXB;
    double Xl[ORC_MAXB][36], Ic[ORC_MAXB][36], S[ORC_MAXB][6];
    (void)NBMAX;
    int nb = m->nb;
    double Rb[9];
    quat_to_R(Rb, q[3], q[4], q[5], q[n]);
    for (int i = 0; i < nb; i++) {
        spatial_inertia(Ic[i], m->mass[i], m->com[i], m->inertia[i]);
        if (i == 0) continue;
        double Rj[9], T[9], E[9];
        axis_angle_R(Rj, m->axis[i], q[6 + i - 1]);
        m3mul(T, m->R_T[i], Rj);
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) E[a * 3 + b] = T[b * 3 + a];
        xform(Xl[i], E, m->p_T[i]);
        for (int a = 0; a < 3; a++) { S[i][a] = m->axis[i][a]; S[i][3 + a] = 0.0; }
    }
    for (int i = nb - 1; i > 0; i--) {
        double T[36], U[36];
        mm(T, 6, Ic[i], 6, Xl[i], 6, 6, 6, 6);
        mtm(U, 6, Xl[i], 6, T, 6, 6, 6, 6);
        int par = m->parent[i];
        for (int a = 0; a < 36; a++) Ic[par][a] += U[a];
    }
    for (int i = 0; i < n * n; i++) A[i] = 0.0;
    double S0[36];
    memset(S0, 0, sizeof(S0));
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) S0[(3 + a) * 6 + b] = Rb[b * 3 + a];
    S0[0 * 6 + 3] = S0[1 * 6 + 4] = S0[2 * 6 + 5] = 1.0;
    {
        double T[36], U[36];
        mm(T, 6, Ic[0], 6, S0, 6, 6, 6, 6);
        mtm(U, 6, S0, 6, T, 6, 6, 6, 6);
        for (int a = 0; a < 6; a++)
            for (int b = 0; b < 6; b++) A[a * n + b] = U[a * 6 + b];
    }
    for (int i = 1; i < nb; i++) {
        int di = 6 + i - 1;
        double F[6], F2[6];
        mv(F, Ic[i], 6, S[i], 6, 6);
        double s = 0.0;
        for (int a = 0; a < 6; a++) s += S[i][a] * F[a];
        A[di * n + di] = s;
        int j = i;
        while (m->parent[j] > 0) {
            for (int a = 0; a < 6; a++) {
                double t = 0.0;
                for (int b = 0; b < 6; b++) t += Xl[j][b * 6 + a] * F[b];
                F2[a] = t;
            }
            memcpy(F, F2, sizeof(F));
            j = m->parent[j];
            int dj = 6 + j - 1;
            s = 0.0;
            for (int a = 0; a < 6; a++) s += F[a] * S[j][a];
            A[di * n + dj] = A[dj * n + di] = s;
        }
        for (int a = 0; a < 6; a++) {
            double t = 0.0;
            for (int b = 0; b < 6; b++) t += Xl[j][b * 6 + a] * F[b];
            F2[a] = t;
        }
        for (int a = 0; a < 6; a++) {
            double t = 0.0;
            for (int b = 0; b < 6; b++) t += S0[b * 6 + a] * F2[b];
            A[di * n + a] = A[a * n + di] = t;
        }
    }
}

/* CalcPointJacobian6D + row swap to [linear; angular] (src/link.cpp:98-119, src/contact_constraint.cpp:51-77) */
static void point_jacobian(const orc_model *m, double (*R)[9], double (*p)[3], int body, const double *pl, double *J, int n) {
    double P[3], t[3];
    m3v(t, R[body], pl);
    for (int k = 0; k < 3; k++) P[k] = p[body][k] + t[k];
    for (int i = 0; i < 6 * n; i++) J[i] = 0.0;
    J[0 * n + 0] = J[1 * n + 1] = J[2 * n + 2] = 1.0;
    double d[3], w[3], c[3];
    for (int k = 0; k < 3; k++) d[k] = P[k] - p[0][k];
    for (int k = 0; k < 3; k++) {
        w[0] = R[0][k]; w[1] = R[0][3 + k]; w[2] = R[0][6 + k];
        cross3(c, w, d);
        for (int a = 0; a < 3; a++) { J[a * n + 3 + k] = c[a]; J[(3 + a) * n + 3 + k] = w[a]; }
    }
    int j = body;
    while (j > 0) {
        m3v(w, R[j], m->axis[j]);
        for (int k = 0; k < 3; k++) d[k] = P[k] - p[j][k];
        cross3(c, w, d);
        for (int a = 0; a < 3; a++) { J[a * n + 6 + j - 1] = c[a]; J[(3 + a) * n + 6 + j - 1] = w[a]; }
        j = m->parent[j];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* cone rows  (src/wbd.cpp:59-97) and C~ = A_const_a * A_rot (src/dwbc.cpp:1018-1039)           */
/* ------------------------------------------------------------------------------------------ */
static void cone_matrix(const orc_setup *su, const int *act_idx, int nc, double (*R)[9], double *Ct /* 10nc x 6nc */) {
    int cd = 6 * nc;
    for (int i = 0; i < 10 * nc * cd; i++) Ct[i] = 0.0;
    for (int a = 0; a < nc; a++) {
        int ci = act_idx[a];
        double L[60];
        memset(L, 0, sizeof(L));
        double lx = su->c_lx[ci], ly = su->c_ly[ci], mu = su->c_mu[ci], muz = su->c_muz[ci];
        L[0 * 6 + 2] = -lx; L[0 * 6 + 4] = -1;
        L[1 * 6 + 2] = -lx; L[1 * 6 + 4] = 1;
        L[2 * 6 + 2] = -ly; L[2 * 6 + 3] = -1;
        L[3 * 6 + 2] = -ly; L[3 * 6 + 3] = 1;
        L[4 * 6 + 0] = 1; L[4 * 6 + 2] = -mu;
        L[5 * 6 + 0] = -1; L[5 * 6 + 2] = -mu;
        L[6 * 6 + 1] = 1; L[6 * 6 + 2] = -mu;
        L[7 * 6 + 1] = -1; L[7 * 6 + 2] = -mu;
        L[8 * 6 + 5] = 1; L[8 * 6 + 2] = -muz;
        L[9 * 6 + 5] = -1; L[9 * 6 + 2] = -muz;
        const double *Rc = R[su->c_link[ci]];
        /* A_rot block = Rc^T ; (L * blkdiag(Rc^T,Rc^T))[r][3h+j] = sum_i L[r][3h+i] * Rc[j][i] */
        for (int r = 0; r < 10; r++)
            for (int h = 0; h < 2; h++)
                for (int j = 0; j < 3; j++) {
                    double s = 0.0;
                    for (int i = 0; i < 3; i++) s += L[r * 6 + 3 * h + i] * Rc[j * 3 + i];
                    Ct[(10 * a + r) * cd + 6 * a + 3 * h + j] = s;
                }
    }
}

static int task_link_dof(int mode) { return mode <= 2 ? 6 : 3; } /* src/task.cpp:14-31 */

/* ------------------------------------------------------------------------------------------ */
/* the control cycle                                                                            */
/* ------------------------------------------------------------------------------------------ */
void orc_cycle(const orc_model *mdl, const orc_setup *su, const double *q, const uint8_t *flags, const double *fstar,
               orc_out *out, orc_debug *dbg) {
    const int n = mdl->ndof, m = n - 6, nb = mdl->nb;
    double R[ORC_MAXB][9], p[ORC_MAXB][3];
    static const double zero3[3] = {0, 0, 0};
    memset(out, 0, sizeof(*out));

    /* ---- UpdateKinematics (src/dwbc.cpp:279-371) ---- */
    double A[ORC_MAXN * ORC_MAXN], Ai[ORC_MAXN * ORC_MAXN];
    forward_kinematics(mdl, q, R, p);
    crba(mdl, q, A, n);
    int spd = chol_inverse(A, n, n, Ai, n);
    double mtot = 0.0;
    for (int i = 0; i < nb; i++) mtot += mdl->mass[i];
    /* com (dwbc.cpp:320-324), CMM (331-340), J_com (346-356), G (358) */
    double skm[9], A30[9], cfp[3];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) A30[a * 3 + b] = A[(3 + a) * n + b] / mtot;
    m3mul(skm, R[0], A30);
    cfp[0] = skm[7]; cfp[1] = skm[2]; cfp[2] = skm[3];
    for (int k = 0; k < 3; k++) out->com[k] = cfp[k] + q[k];
    double cm[36], CMM[6 * ORC_MAXN];
    memset(cm, 0, sizeof(cm));
    for (int a = 0; a < 6; a++) cm[a * 6 + a] = 1.0;
    {
        double S[9];
        skew3(S, cfp);
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                cm[(3 + a) * 6 + 3 + b] = R[0][a * 3 + b];
                cm[(3 + a) * 6 + b] = S[b * 3 + a];
            }
    }
    mm(CMM, n, cm, 6, A, n, 6, 6, n);
    double G[ORC_MAXN];
    double Jcom[6 * ORC_MAXN]; /* link_.back().jac_ = jac_com_ (dwbc.cpp:352-353): the Jacobian of the "COM" link */
    {
        double A33[9], T[9], Icom[9], S[9], SSt[9], SI[36], SIi[36];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) A33[a * 3 + b] = A[(3 + a) * n + 3 + b];
        m3mul(T, R[0], A33);
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) Icom[a * 3 + b] = T[a * 3] * R[0][b * 3] + T[a * 3 + 1] * R[0][b * 3 + 1] + T[a * 3 + 2] * R[0][b * 3 + 2];
        skew3(S, cfp);
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) SSt[a * 3 + b] = S[a * 3] * S[b * 3] + S[a * 3 + 1] * S[b * 3 + 1] + S[a * 3 + 2] * S[b * 3 + 2];
        memset(SI, 0, sizeof(SI));
        for (int a = 0; a < 3; a++) {
            SI[a * 6 + a] = mtot;
            for (int b = 0; b < 3; b++) SI[(3 + a) * 6 + 3 + b] = Icom[a * 3 + b] - mtot * SSt[a * 3 + b];
        }
        lu_inverse(SI, 6, 6, SIi, 6);
        mm(Jcom, n, SIi, 6, CMM, n, 6, 6, n);
        for (int j = 0; j < n; j++) G[j] = -Jcom[2 * n + j] * mtot * (-GRAV);
    }
    memcpy(out->G, G, sizeof(double) * n);

    /* ---- SetContact / UpdateContactConstraint (include/dwbc.h:432-474, src/dwbc.cpp:433-454) ---- */
    int act_idx[ORC_MAXCON], nc = 0;
    double JC[ORC_MAXC * ORC_MAXN];
    for (int i = 0; i < su->n_contacts; i++)
        if (flags[i]) {
            point_jacobian(mdl, R, p, su->c_link[i], su->c_point[i], JC + 6 * nc * n, n);
            act_idx[nc++] = i;
        }
    const int cd = 6 * nc, k = cd > 6 ? cd - 6 : 0;
    out->cdof = cd;
    out->k = k;

    /* ---- CalcContactConstraint (src/wbd.cpp:108-143) ---- */
    double Lam[ORC_MAXC * ORC_MAXC], JCit[ORC_MAXC * ORC_MAXN], NC[ORC_MAXN * ORC_MAXN], AiNC[ORC_MAXN * ORC_MAXN];
    double W[ORC_MAXM * ORC_MAXM], Wi[ORC_MAXM * ORC_MAXM], V2[ORC_MAXC * ORC_MAXM], NwJw[ORC_MAXM * ORC_MAXC];
    int st_contact = spd;
    {
        double JA[ORC_MAXC * ORC_MAXN], JAJ[ORC_MAXC * ORC_MAXC];
        mm(JA, n, JC, n, Ai, n, cd, n, n);
        mmt(JAJ, cd, JA, n, JC, n, cd, n, cd);
        if (cd > 0 && !lu_inverse(JAJ, cd, cd, Lam, cd)) st_contact = 0;
        mm(JCit, n, Lam, cd, JA, n, cd, cd, n);
        mtm(NC, n, JC, n, JCit, n, n, cd, n);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) NC[i * n + j] = (i == j ? 1.0 : 0.0) - NC[i * n + j];
        mm(AiNC, n, Ai, n, NC, n, n, n, n);
        for (int i = 0; i < m; i++)
            for (int j = 0; j < m; j++) W[i * m + j] = AiNC[(6 + i) * n + 6 + j];
        int rank = 0;
        double V2full[ORC_MAXM * ORC_MAXM];
        orc_pinv_cod(W, m, m, COD_THRESHOLD, Wi, V2full, &rank);
        if (cd > 6 && m - rank <= ORC_MAXC) memcpy(V2, V2full, sizeof(double) * (m - rank) * m);
        if (cd > 6) {
            if (m - rank != k) st_contact = 0;
            int vr = m - rank; /* rows of V2 actually produced */
            if (vr == k) {
                double JV[ORC_MAXC * ORC_MAXC], JVi[ORC_MAXC * ORC_MAXC];
                for (int i = 0; i < k; i++)
                    for (int j = 0; j < k; j++) {
                        double s = 0.0;
                        for (int c = 0; c < m; c++) s += JCit[i * n + 6 + c] * V2[j * m + c];
                        JV[i * k + j] = s;
                    }
                if (!lu_inverse(JV, k, k, JVi, k)) st_contact = 0;
                for (int i = 0; i < m; i++)
                    for (int j = 0; j < k; j++) {
                        double s = 0.0;
                        for (int c = 0; c < k; c++) s += V2[c * m + i] * JVi[c * k + j];
                        NwJw[i * k + j] = s;
                    }
            } else {
                memset(NwJw, 0, sizeof(double) * m * (k > 0 ? k : 1));
            }
        }
    }
    out->st_contact = st_contact;

    /* ---- CalcGravCompensation (src/wbd.cpp:186-192) ---- */
    double tau_g[ORC_MAXM], PC[ORC_MAXC];
    {
        double NG[ORC_MAXN], t1[ORC_MAXM];
        mv(NG, NC, n, G, n, n);
        mv(t1, Ai + 6 * n, n, NG, m, n);
        mv(tau_g, Wi, m, t1, m, m);
        mv(PC, JCit, n, G, cd, n);
    }
    memcpy(out->tau_grav, tau_g, sizeof(double) * m);
    memcpy(out->P_C, PC, sizeof(double) * cd);

    /* cone data shared by every QP of the cycle */
    double Ct[10 * ORC_MAXCON * ORC_MAXC], Atemp[10 * ORC_MAXCON * ORC_MAXM], CtPC[10 * ORC_MAXCON];
    const int ncone = 10 * nc, nlim = su->has_tau_lim ? 2 * m : 0;
    cone_matrix(su, act_idx, nc, R, Ct);
    for (int r = 0; r < ncone; r++) {
        for (int j = 0; j < m; j++) {
            double s = 0.0;
            for (int c = 0; c < cd; c++) s += Ct[r * cd + c] * JCit[c * n + 6 + j];
            Atemp[r * m + j] = s;
        }
        double s = 0.0;
        for (int c = 0; c < cd; c++) s += Ct[r * cd + c] * PC[c];
        CtPC[r] = s;
    }

    /* ---- CalcTaskSpace (src/dwbc.cpp:685-816; src/wbd.cpp:207-261) ---- */
    const int L = su->n_levels;
    static const double I6[1] = {0};
    (void)I6;
    double Jt[ORC_MAXL][ORC_MAXT * ORC_MAXN], Lt[ORC_MAXL][ORC_MAXT * ORC_MAXT], Jkt[ORC_MAXL][ORC_MAXM * ORC_MAXT];
    double Null[ORC_MAXL][ORC_MAXM * ORC_MAXM];
    int tdof[ORC_MAXL];
    double AiNC2[ORC_MAXN * ORC_MAXN];
    mm(AiNC2, n, Ai, n, NC, n, n, n, n); /* wbd.cpp:210 recomputes A_inv * N_C */
    for (int lv = 0; lv < L; lv++) {
        int t = 0;
        for (int li = 0; li < su->t_nlinks[lv]; li++) {
            int mode = su->t_mode[lv][li], link = su->t_link[lv][li];
            double J6[6 * ORC_MAXN];
            const double *pt = zero3;
            if ((mode == 1 || mode == 4) && link < mdl->nb) pt = mdl->com[link];
            else if (mode == 2 || mode == 5) pt = su->t_point[lv][li];
            if (link == mdl->nb) memcpy(J6, Jcom, sizeof(double) * 6 * n); /* the COM link (dwbc.cpp:230-231,708-780) */
            else point_jacobian(mdl, R, p, link, pt, J6, n);
            if (mode <= 2) { memcpy(Jt[lv] + t * n, J6, sizeof(double) * 6 * n); t += 6; }
            else if (mode <= 5) { memcpy(Jt[lv] + t * n, J6, sizeof(double) * 3 * n); t += 3; }
            else { memcpy(Jt[lv] + t * n, J6 + 3 * n, sizeof(double) * 3 * n); t += 3; }
        }
        tdof[lv] = t;
        out->task_dof[lv] = t;
        double JA[ORC_MAXT * ORC_MAXN], JAJ[ORC_MAXT * ORC_MAXT], Qf[ORC_MAXT * ORC_MAXN], Q[ORC_MAXT * ORC_MAXM];
        double QW[ORC_MAXT * ORC_MAXM], QWQ[ORC_MAXT * ORC_MAXT], QWQi[ORC_MAXT * ORC_MAXT], WQt[ORC_MAXM * ORC_MAXT];
        mm(JA, n, Jt[lv], n, AiNC2, n, t, n, n);
        mmt(JAJ, t, JA, n, Jt[lv], n, t, n, t);
        lu_inverse(JAJ, t, t, Lt[lv], t);
        mm(Qf, n, Lt[lv], t, JA, n, t, t, n);
        for (int i = 0; i < t; i++)
            for (int j = 0; j < m; j++) Q[i * m + j] = Qf[i * n + 6 + j];
        mm(QW, m, Q, m, Wi, m, t, m, m);
        mmt(QWQ, t, QW, m, Q, m, t, m, t);
        orc_pinv_cod(QWQ, t, t, COD_THRESHOLD, QWQi, 0, 0);
        mmt(WQt, t, Wi, m, Q, m, m, m, t);
        mm(Jkt[lv], t, WQt, t, QWQi, t, m, t, t);
        if (lv != L - 1) { /* CalculateTaskNullSpace wbd.cpp:257-261 */
            double JL[ORC_MAXM * ORC_MAXT], JLJ[ORC_MAXM * ORC_MAXN], T[ORC_MAXM * ORC_MAXM];
            mm(JL, t, Jkt[lv], t, Lt[lv], t, m, t, t);
            mm(JLJ, n, JL, t, Jt[lv], n, m, t, n);
            for (int i = 0; i < m; i++)
                for (int j = 0; j < m; j++) {
                    double s = 0.0;
                    for (int c = 0; c < n; c++) s += JLJ[i * n + c] * AiNC[c * n + 6 + j];
                    T[i * m + j] = (i == j ? 1.0 : 0.0) - s;
                }
            if (lv == 0) memcpy(Null[lv], T, sizeof(double) * m * m);
            else mm(Null[lv], m, Null[lv - 1], m, T, m, m, m, m);
        }
    }

    /* ---- CalcTaskControlTorque(hqp=true) (src/dwbc.cpp:818-873, 941-1127) ---- */
    double tau_t[ORC_MAXM], tau_c[ORC_MAXM];
    memset(tau_t, 0, sizeof(tau_t));
    memset(tau_c, 0, sizeof(tau_c));
    int st_task = 1, foff = 0;
    for (int lv = 0; lv < L && st_task; lv++) {
        const int t = tdof[lv], nv = t + k, rows = nlim + ncone;
        double Nt[ORC_MAXM * ORC_MAXT], JL[ORC_MAXM * ORC_MAXT];
        mm(JL, t, Jkt[lv], t, Lt[lv], t, m, t, t);
        if (lv == 0) memcpy(Nt, JL, sizeof(double) * m * t);
        else mm(Nt, t, Null[lv - 1], m, JL, t, m, m, t);
        const double *fs = fstar + foff;
        double base[ORC_MAXM];
        for (int i = 0; i < m; i++) {
            double s = tau_g[i] + tau_t[i];
            for (int j = 0; j < t; j++) s += Nt[i * t + j] * fs[j];
            base[i] = s;
        }
        double QA[ORC_MAXR * ORC_MAXV], Qub[ORC_MAXR];
        memset(QA, 0, sizeof(double) * rows * nv);
        if (nlim) {
            for (int i = 0; i < m; i++) {
                for (int j = 0; j < t; j++) { QA[i * nv + j] = Nt[i * t + j]; QA[(m + i) * nv + j] = -Nt[i * t + j]; }
                for (int j = 0; j < k; j++) { QA[i * nv + t + j] = NwJw[i * k + j]; QA[(m + i) * nv + t + j] = -NwJw[i * k + j]; }
                Qub[i] = su->tau_lim[i] - base[i];
                Qub[m + i] = su->tau_lim[i] + base[i];
            }
        }
        for (int r = 0; r < ncone; r++) {
            for (int j = 0; j < t; j++) {
                double s = 0.0;
                for (int c = 0; c < m; c++) s += Atemp[r * m + c] * Nt[c * t + j];
                QA[(nlim + r) * nv + j] = -s;
            }
            for (int j = 0; j < k; j++) {
                double s = 0.0;
                for (int c = 0; c < m; c++) s += Atemp[r * m + c] * NwJw[c * k + j];
                QA[(nlim + r) * nv + t + j] = -s;
            }
            double s = CtPC[r];
            for (int c = 0; c < m; c++) s -= Atemp[r * m + c] * base[c];
            Qub[nlim + r] = -s;
        }
        double x[ORC_MAXV];
        int ok = orc_solve_qp(QA, Qub, rows, nv, t, 1000, x, out->qp_act[lv], &out->qp_nact[lv], &out->qp_iter[lv]);
        if (dbg) {
            memcpy(dbg->qpA[lv], QA, sizeof(double) * rows * nv);
            memcpy(dbg->qpub[lv], Qub, sizeof(double) * rows);
            dbg->qp_rows[lv] = rows;
            dbg->qp_cols[lv] = nv;
        }
        if (!ok) { st_task = 0; break; } /* f_star_qp_, contact_qp_ zeroed; cascade aborts (dwbc.cpp:836,1119) */
        memcpy(out->fstar_qp[lv], x, sizeof(double) * t);
        memcpy(out->contact_qp[lv], x + t, sizeof(double) * k);
        /* torque_h_ = J_kt Lambda (f* + f*_qp); torque_task_ += Null_{i-1} torque_h_ (dwbc.cpp:839-849) */
        for (int i = 0; i < m; i++) {
            double s = 0.0;
            for (int j = 0; j < t; j++) s += Nt[i * t + j] * (fs[j] + x[j]);
            tau_t[i] += s;
        }
        for (int i = 0; i < m; i++) { /* torque_contact_ = NwJw * contact_qp_ (dwbc.cpp:851); k==0 -> 0 */
            double s = 0.0;
            for (int j = 0; j < k; j++) s += NwJw[i * k + j] * x[t + j];
            tau_c[i] = s;
        }
        foff += t;
    }
    out->st_task = st_task;

    /* ---- CalcContactRedistribute(hqp=true) (src/dwbc.cpp:1372-1568) ---- */
    int st_redis = 1;
    if (k > 0) {
        const int rows = nlim + ncone;
        double tin[ORC_MAXM], QA[ORC_MAXR * ORC_MAXV], Qub[ORC_MAXR], x[ORC_MAXV];
        for (int i = 0; i < m; i++) tin[i] = tau_g[i] + tau_t[i] + tau_c[i];
        if (nlim)
            for (int i = 0; i < m; i++) {
                for (int j = 0; j < k; j++) { QA[i * k + j] = NwJw[i * k + j]; QA[(m + i) * k + j] = -NwJw[i * k + j]; }
                Qub[i] = su->tau_lim[i] - tin[i];
                Qub[m + i] = su->tau_lim[i] + tin[i];
            }
        for (int r = 0; r < ncone; r++) { /* CM = -C~ ; rows CM*J*NwJw c <= CM*P_C - CM*J*tau_in */
            for (int j = 0; j < k; j++) {
                double s = 0.0;
                for (int c = 0; c < m; c++) s += Atemp[r * m + c] * NwJw[c * k + j];
                QA[(nlim + r) * k + j] = -s;
            }
            double s = -CtPC[r];
            for (int c = 0; c < m; c++) s += Atemp[r * m + c] * tin[c];
            Qub[nlim + r] = s;
        }
        /* the reference pads contact_link_num_*rows zero rows (dwbc.cpp:1420): dropped */
        int ok = orc_solve_qp_tol(QA, Qub, rows, k, k, 300, x, out->qp_act[L], &out->qp_nact[L], &out->qp_iter[L], QP_FEAS_TOL);
        if (dbg) {
            memcpy(dbg->qpA[L], QA, sizeof(double) * rows * k);
            memcpy(dbg->qpub[L], Qub, sizeof(double) * rows);
            dbg->qp_rows[L] = rows;
            dbg->qp_cols[L] = k;
        }
        if (ok) {
            memcpy(out->cf_redis, x, sizeof(double) * k);
            for (int i = 0; i < m; i++) {
                double s = 0.0;
                for (int j = 0; j < k; j++) s += NwJw[i * k + j] * x[j];
                tau_c[i] += s;
            }
        } else {
            st_redis = 0;
            memset(tau_c, 0, sizeof(tau_c));
        }
    } else {
        memset(tau_c, 0, sizeof(tau_c));
    }
    out->st_redis = st_redis;
    memcpy(out->tau_task, tau_t, sizeof(double) * m);
    memcpy(out->tau_contact, tau_c, sizeof(double) * m);
    /* getContactForce(tau_total) (src/wbd.cpp:268-271) */
    for (int c = 0; c < cd; c++) {
        double s = -PC[c];
        for (int j = 0; j < m; j++) s += JCit[c * n + 6 + j] * (tau_g[j] + tau_t[j] + tau_c[j]);
        out->contact_force[c] = s;
    }
    out->status = st_contact && st_task && st_redis;

    if (dbg) {
        memcpy(dbg->A, A, sizeof(double) * n * n);
        memcpy(dbg->A_inv, Ai, sizeof(double) * n * n);
        memcpy(dbg->J_C, JC, sizeof(double) * cd * n);
        memcpy(dbg->Lambda_c, Lam, sizeof(double) * cd * cd);
        memcpy(dbg->J_C_INV_T, JCit, sizeof(double) * cd * n);
        memcpy(dbg->N_C, NC, sizeof(double) * n * n);
        memcpy(dbg->A_inv_N_C, AiNC, sizeof(double) * n * n);
        memcpy(dbg->W, W, sizeof(double) * m * m);
        memcpy(dbg->W_inv, Wi, sizeof(double) * m * m);
        memcpy(dbg->V2, V2, sizeof(double) * k * m);
        memcpy(dbg->NwJw, NwJw, sizeof(double) * m * k);
        memcpy(dbg->CMM, CMM, sizeof(double) * 6 * n);
        for (int lv = 0; lv < L; lv++) {
            memcpy(dbg->J_task[lv], Jt[lv], sizeof(double) * tdof[lv] * n);
            memcpy(dbg->Lambda_task[lv], Lt[lv], sizeof(double) * tdof[lv] * tdof[lv]);
            memcpy(dbg->J_kt[lv], Jkt[lv], sizeof(double) * m * tdof[lv]);
            if (lv != L - 1) memcpy(dbg->Null_task[lv], Null[lv], sizeof(double) * m * m);
        }
        memcpy(dbg->link_R, R, sizeof(double) * 9 * nb);
        memcpy(dbg->link_p, p, sizeof(double) * 3 * nb);
    }
}

int orc_cycle_batch(const orc_model *mdl, const orc_setup *su, int B, const double *q, const uint8_t *flags,
                    const double *fstar, int fstar_stride, double *tau_out, double *wrench_out, int32_t *status_out,
                    int nthreads) {
    const int n = mdl->ndof, m = n - 6, wc = 6 * su->n_contacts;
    int used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    used = nthreads > 0 ? nthreads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int b = 0; b < B; b++) {
        orc_out o;
        orc_cycle(mdl, su, q + (size_t)b * (n + 1), flags + (size_t)b * su->n_contacts, fstar + (size_t)b * fstar_stride, &o, 0);
        double *t = tau_out + (size_t)b * 3 * m;
        memcpy(t, o.tau_grav, sizeof(double) * m);
        memcpy(t + m, o.tau_task, sizeof(double) * m);
        memcpy(t + 2 * m, o.tau_contact, sizeof(double) * m);
        double *w = wrench_out + (size_t)b * wc;
        for (int c = 0; c < wc; c++) w[c] = c < o.cdof ? o.contact_force[c] : 0.0;
        status_out[b] = o.status;
    }
    return used;
}

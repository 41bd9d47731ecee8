// dwbc_qp_wave.h -- register-resident batched active-set QP, one wavefront per problem.
//
//   lexmin( 1/2|x[:t]|^2 , 1/2|x[t:]|^2 )  s.t.  -lo_r <= g_r . x <= hi_r      (H = diag(I_t, 0_k), g = 0)
//
// Replaces CQuadraticProgram::SolveQPoases (reference src/qp_wrapper.cpp:192-380) for the QPs assembled by
// CalcSingleTaskTorqueWithQP / CalcContactRedistribute (reference src/dwbc.cpp:988-1053, 1458-1517).
//
// Layout: lane r owns constraint row r in registers -- the 33 torque-limit rows are stored once as two-sided rows
// (+/-(row) <= tau_lim -/+ base), the 20 friction/CoP cone rows one-sided -- so 53 lanes hold the whole 86-row QP and
// a slack evaluation is 12 FMAs per lane against the uniform iterate.  The Goldfarb-Idnani working-set state lives in
// lanes 0..11: lane a holds row a of the pseudo-inverse N^+ of the active normals and lane i row i of N; every step is
// a 12-term dot product per lane plus v_readlane broadcasts -- no LDS round trips, no triangular solves.  Loops that
// would otherwise be unrolled over the working-set size run over a uniform counter with lane masks, so the whole solver
// stays a few KB of code (the kernel must fit the instruction cache).
// See DESIGN.md "QP canon" for the definition of the returned point.
#pragma once
#include "dwbc_wave.h"

namespace dwbc {

constexpr int kQpN = 12;  // max variables (6 task + 6 contact-null)

struct QpRows {
    PLA(double, g, kQpN);  // row coefficients, contact columns already scaled by kQpScaleGI, zero padded
    PL(double, hi);        // g.x <= hi
    PL(double, lo);        // -g.x <= lo   (lo = +inf: one-sided row)
    PL(int, id_hi);        // reference row index of the hi side (for diagnostics)
    PL(int, id_lo);
};

struct QpResult {
    int status, iters, nact;
    double viol;
    double x[kQpN];   // uniform, unscaled [delta (t); c (k)]
    int act[kQpN];    // reference row indices of the working set
};

#define DWBC_QP_INF 1.0e300

// uniform 12-array element with a uniform dynamic index
DWBC_WDEV double upick12(const double *a, int idx) {
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < kQpN; i++) v = (i == idx) ? a[i] : v;
    return v;
}

template <int DUMMY>
DWBC_WDEV void qp_solve_wave(QpRows &R, int nv, int t, int max_iter, QpResult &out, double *V /* LDS, 176 doubles */) {
    DWBC_LANE_DECL;
    const int k = nv - t;
    PLA(double, Np, kQpN);
    PLA(double, Nr, kQpN);
    PL(double, r);
    PL(double, z);
    PL(double, u);
    PL(double, rgn);   // 1 / |g|
    PL(double, gno);   // |g|
    PL(int, akey);     // lane a < q: (owner lane << 1) | side of the a-th working-set member
    PL(int, actf);     // bit0: hi side in the working set, bit1: lo side
    PL(double, val);
    PL(int, key);
    double xu[kQpN], n[kQpN], ru[kQpN], zu[kQpN];
    double *nbuf = V, *rbuf = V + 16, *zbuf = V + 32;  // LDS broadcast buffers (V is only needed by the final solve)
    PL(double, nl);    // lane i < nv: component i of the current normal
#pragma unroll
    for (int i = 0; i < kQpN; i++) { xu[i] = 0.0; n[i] = 0.0; }
    LANES {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < kQpN; j++) { s += LV(R.g)[j] * LV(R.g)[j]; LV(Np)[j] = 0.0; LV(Nr)[j] = 0.0; }
        s = sqrt(s);
        LV(gno) = s < 1e-300 ? 1e-300 : s;
        LV(rgn) = 1.0 / LV(gno);
        LV(u) = 0.0;
        LV(r) = 0.0;
        LV(z) = 0.0;
        LV(nl) = 0.0;
        LV(akey) = 0;
        LV(actf) = 0;
    }
    // ---- Goldfarb-Idnani.  mode 0: look for the most violated row; 1: primal/dual step for row (p, side);
    //      2: re-adding working-set member `ra` after a drop (N, N^+ are rebuilt column by column).
    //      Vectors that every lane needs (normal n, r = N^+ n, z = n - N r) are broadcast through 12-double LDS
    //      buffers: one store per lane + uniform reads, instead of 24 v_readlane and a 12-way select chain each.
    int q = 0, it = 0, status = 1, mode = 0, ra = 0, p = 0, side = 0, kmin = 0;
    double up = 0.0, bp = 0.0, gnp = 1.0, worst = 0.0;
    for (;;) {
        if (mode == 0) {
            LANES {
                double d = 0.0;
#pragma unroll
                for (int j = 0; j < kQpN; j++) d += LV(R.g)[j] * xu[j];
                const double sh = ((LV(actf) & 1) || LV(R.hi) >= DWBC_QP_INF) ? DWBC_QP_INF : (LV(R.hi) - d) * LV(rgn);
                const double sl = ((LV(actf) & 2) || LV(R.lo) >= DWBC_QP_INF) ? DWBC_QP_INF : (LV(R.lo) + d) * LV(rgn);
                const bool lo_side = sl < sh;
                LV(val) = lo_side ? sl : sh;
                LV(key) = (lane << 1) | (lo_side ? 1 : 0);
            }
            WAVE_ARGMIN(val, key, worst, kmin);
            if (!(worst < -kQpTol)) break;
            p = kmin >> 1;
            side = kmin & 1;
            up = 0.0;
            mode = 1;
        }
        // normal of the row being processed (GI is stated for n^T x >= b': hi side -> n = -g, lo side -> n = +g),
        // published by its owner lane
        int pe = p, se = side;
        if (mode == 2) {
            const int ka = BCASTI(akey, ra);
            pe = ka >> 1;
            se = ka & 1;
        }
        WSYNC();
        LANES {
            if (lane == pe) {
#pragma unroll
                for (int j = 0; j < kQpN; j++) nbuf[j] = se ? LV(R.g)[j] : -LV(R.g)[j];
                nbuf[12] = se ? LV(R.lo) : LV(R.hi);
                nbuf[13] = LV(gno);
            }
        }
        WSYNC();
#pragma unroll
        for (int j = 0; j < kQpN; j++) n[j] = nbuf[j];
        if (mode == 1) { bp = nbuf[12]; gnp = nbuf[13]; }
        const int qe = (mode == 2) ? ra : q;  // columns currently in N
        // Greville projection: r = N^+ n (lane a), z = n - N r (lane i)
        LANES {
            LV(nl) = nbuf[lane < kQpN ? lane : 0];
            double s_ = 0.0;
#pragma unroll
            for (int i = 0; i < kQpN; i++) s_ += LV(Np)[i] * n[i];
            LV(r) = (lane < qe) ? s_ : 0.0;
            if (lane < kQpN) rbuf[lane] = LV(r);
        }
        WSYNC();
#pragma unroll
        for (int a = 0; a < kQpN; a++) ru[a] = rbuf[a];
        LANES {
            double s_ = LV(nl);
#pragma unroll
            for (int a = 0; a < kQpN; a++) s_ -= LV(Nr)[a] * ru[a];
            LV(z) = (lane < nv) ? s_ : 0.0;
            if (lane < kQpN) zbuf[lane] = LV(z);
        }
        WSYNC();
        double zn2 = 0.0, zg = 0.0;
#pragma unroll
        for (int i = 0; i < kQpN; i++) {
            zu[i] = zbuf[i];
            zn2 += zu[i] * zu[i];
            zg += zu[i] * n[i];
        }
        bool commit = (mode == 2);
        if (mode == 1) {
            if (++it > max_iter) { status = 0; break; }
            double rmax = 1.0;
#pragma unroll
            for (int a = 0; a < kQpN; a++) rmax = fabs(ru[a]) > rmax ? fabs(ru[a]) : rmax;
            LANES {
                const bool ok = (lane < q) && (LV(r) > 1e-13 * rmax);
                LV(val) = ok ? LV(u) * fast_rcp(LV(r)) : DWBC_QP_INF;
                LV(key) = lane;
            }
            double t1;
            int l;
            WAVE_ARGMIN(val, key, t1, l);
            double gx = 0.0;
#pragma unroll
            for (int j = 0; j < kQpN; j++) gx += n[j] * xu[j];
            const double sp = bp + gx;  // slack of the violated side (negative): hi - g.x or lo + g.x
            const bool zok = zn2 > (1e-10 * gnp) * (1e-10 * gnp) && q < nv;
            const double t2 = zok ? -sp * fast_rcp(zg) : DWBC_QP_INF;
            const double tstep = t1 < t2 ? t1 : t2;
            if (!(tstep < DWBC_QP_INF)) { status = 0; break; }
            const bool full = zok && t2 <= t1;
            if (zok) {
#pragma unroll
                for (int i = 0; i < kQpN; i++) xu[i] += tstep * zu[i];
            }
            LANES { LV(u) -= tstep * LV(r); }
            up += tstep;
            if (full) {
                commit = true;
            } else {
                // drop working-set member l, compact (u, akey), restart N / N^+ from empty
                const int kl = BCASTI(akey, l);
                PL(double, un);
                PL(int, kn);
                LANES {
                    const int src = (lane >= l && lane < q - 1) ? lane + 1 : lane;
                    LV(un) = SHFL(u, src);
                    LV(kn) = SHFL(akey, src);
                }
                LANES {
                    LV(u) = LV(un);
                    LV(akey) = LV(kn);
                    if (lane == (kl >> 1)) LV(actf) &= ~((kl & 1) ? 2 : 1);
#pragma unroll
                    for (int j = 0; j < kQpN; j++) { LV(Np)[j] = 0.0; LV(Nr)[j] = 0.0; }
                }
                q--;
                ra = 0;
                mode = (q > 0) ? 2 : 1;
            }
        }
        if (commit) {
            // append n as column qe of N and update N^+ (Greville)
            const double inv_ = fast_rcp(zn2);
            LANES {
                const double ra_ = LV(r) * inv_;
#pragma unroll
                for (int i = 0; i < kQpN; i++) LV(Np)[i] -= ra_ * zu[i];  // lanes >= qe hold zeros and r = 0
                if (lane == qe) {
#pragma unroll
                    for (int i = 0; i < kQpN; i++) LV(Np)[i] = zu[i] * inv_;
                }
                switch (qe) {  // uniform
                    case 0: LV(Nr)[0] = LV(nl); break;
                    case 1: LV(Nr)[1] = LV(nl); break;
                    case 2: LV(Nr)[2] = LV(nl); break;
                    case 3: LV(Nr)[3] = LV(nl); break;
                    case 4: LV(Nr)[4] = LV(nl); break;
                    case 5: LV(Nr)[5] = LV(nl); break;
                    case 6: LV(Nr)[6] = LV(nl); break;
                    case 7: LV(Nr)[7] = LV(nl); break;
                    case 8: LV(Nr)[8] = LV(nl); break;
                    case 9: LV(Nr)[9] = LV(nl); break;
                    case 10: LV(Nr)[10] = LV(nl); break;
                    default: LV(Nr)[11] = LV(nl); break;
                }
            }
            if (mode == 2) {
                ra++;
                if (ra == q) mode = 1;
            } else {
                LANES {
                    if (lane == q) { LV(u) = up; LV(akey) = kmin; }
                    if (lane == p) LV(actf) |= (side ? 2 : 1);
                }
                q++;
                mode = 0;
            }
        }
    }
    out.iters = it;
    out.nact = q;
    out.status = status;
    out.viol = worst >= DWBC_QP_INF ? 0.0 : worst;
#pragma unroll
    for (int a = 0; a < kQpN; a++) {
        const int ka = BCASTI(akey, a);
        const int ow = ka >> 1;
        const int idh = BCASTI(R.id_hi, ow), idl = BCASTI(R.id_lo, ow);
        out.act[a] = a < q ? ((ka & 1) ? idl : idh) : -1;
    }
#pragma unroll
    for (int i = 0; i < kQpN; i++) out.x[i] = 0.0;
    if (!status || q == 0) return;  // x = 0: failure (caller zeroes the correction) or no active constraint
    // ---- final point from the working set alone: lexicographic least-norm point (contact block weighted) if it is
    //      feasible, else the Tikhonov point (DESIGN.md "QP canon").  Column a of the weighted normal matrix lives in
    //      lane a (registers indexed by VARIABLE); column-pivoted Householder QR whose elimination order visits the
    //      contact variables first -- pos(j) = position of variable j -- which is the row sorting that keeps the 1e9
    //      weight stable.  Pivot column, y and the reflector dot products travel through small LDS buffers.
    const bool lex = (k > 0 && t > 0);
    int pos[kQpN];
#pragma unroll
    for (int j = 0; j < kQpN; j++) pos[j] = (j < nv) ? ((j >= t) ? (j - t) : (k + j)) : (100 + j);  // padding never pivots
    double *cbuf = V + 144, *pbuf = V + 160;  // V[0..143]: reflectors; V is 176 doubles
    for (int attempt = 0; attempt < 2; attempt++) {
        const bool weighted = lex && attempt == 0;
        const double wsc = weighted ? kQpScalePolish / kQpScaleGI : 1.0;
        const double cscale = weighted ? kQpScalePolish : kQpScaleGI;
        PLA(double, c, kQpN);
        PL(double, bb);
        PL(int, done);
        PL(double, w);
        PL(double, beta);
        PL(int, ord);
        LANES {
            const int ka = LV(akey);
            const int ow = (lane < q) ? (ka >> 1) : lane;
            const double sg = (ka & 1) ? -1.0 : 1.0;  // row as  (sg*g).x = b
#pragma unroll
            for (int j = 0; j < kQpN; j++) {
                const double gj = SHFLA(R.g, j, ow);
                LV(c)[j] = (lane < q && j < nv) ? sg * gj * ((j >= t) ? wsc : 1.0) : 0.0;
            }
            const double bh = SHFL(R.hi, ow), bl = SHFL(R.lo, ow);
            LV(bb) = (lane < q) ? ((ka & 1) ? bl : bh) : 0.0;
            LV(done) = (lane < q) ? 0 : 1;
            LV(w) = 0.0;
            LV(beta) = 0.0;
            LV(ord) = 0;
        }
        double yv[kQpN];
#pragma unroll
        for (int i = 0; i < kQpN; i++) yv[i] = 0.0;
        for (int s = 0; s < q; s++) {
            LANES {
                double c2 = 0.0;
#pragma unroll
                for (int j = 0; j < kQpN; j++) c2 += (pos[j] >= s) ? LV(c)[j] * LV(c)[j] : 0.0;
                LV(val) = LV(done) ? DWBC_QP_INF : -c2;
                LV(key) = lane;
            }
            double bn;
            int jp;
            WAVE_ARGMIN(val, key, bn, jp);
            WSYNC();
            LANES {
                if (lane == jp) {
#pragma unroll
                    for (int j = 0; j < kQpN; j++) cbuf[j] = LV(c)[j];
                }
            }
            WSYNC();
            double v[kQpN];
            double nrm2 = 0.0, a0 = 0.0;
#pragma unroll
            for (int j = 0; j < kQpN; j++) {
                const double cj = cbuf[j];
                v[j] = (pos[j] >= s) ? cj : 0.0;
                a0 = (pos[j] == s) ? cj : a0;
                nrm2 += v[j] * v[j];
            }
            const double nrm = sqrt(nrm2);
            const double alpha = a0 > 0 ? -nrm : nrm;
            const double vs0 = a0 - alpha;
            const double vn2 = nrm2 - a0 * a0 + vs0 * vs0;
            const double bt = vn2 > 0.0 ? 2.0 * fast_rcp(vn2) : 0.0;
#pragma unroll
            for (int j = 0; j < kQpN; j++) v[j] = (pos[j] == s) ? vs0 : v[j];
            LANES {
                if (lane == s) { LV(beta) = bt; LV(ord) = jp; }
                if (lane == jp) {
                    LV(done) = 1;
#pragma unroll
                    for (int j = 0; j < kQpN; j++) {
                        V[s * kQpN + j] = v[j];
                        LV(c)[j] = (pos[j] == s) ? alpha : LV(c)[j];
                    }
                } else if (!LV(done)) {
                    double d = 0.0;
#pragma unroll
                    for (int j = 0; j < kQpN; j++) d += v[j] * LV(c)[j];
                    d *= bt;
#pragma unroll
                    for (int j = 0; j < kQpN; j++) LV(c)[j] -= d * v[j];
                }
            }
        }
        // R^T y = b in pivot order: the column of R for pivot s is lane ord[s]'s c at positions 0..s
        for (int s = 0; s < q; s++) {
            const int os = BCASTI(ord, s);
            LANES {
                double sacc = LV(bb), cs = 1.0;
#pragma unroll
                for (int j = 0; j < kQpN; j++) {
                    sacc -= (pos[j] < s) ? LV(c)[j] * yv[j] : 0.0;
                    cs = (pos[j] == s) ? LV(c)[j] : cs;
                }
                LV(val) = sacc * fast_rcp(cs);
            }
            const double ys = BCAST(val, os);
#pragma unroll
            for (int j = 0; j < kQpN; j++) yv[j] = (pos[j] == s) ? ys : yv[j];
        }
        LANES { LV(w) = pick12(yv, lane); }  // x~ = Q [y; 0], variable `lane` in lane `lane`
        WSYNC();
        for (int s = q - 1; s >= 0; s--) {
            PL(double, vs);
            LANES {
                LV(vs) = (lane < kQpN) ? V[s * kQpN + lane] : 0.0;
                if (lane < kQpN) pbuf[lane] = LV(vs) * LV(w);
            }
            WSYNC();
            double d = 0.0;
#pragma unroll
            for (int i = 0; i < kQpN; i++) d += pbuf[i];
            d *= BCAST(beta, s);
            LANES { LV(w) -= d * LV(vs); }
            WSYNC();
        }
        LANES {
            if (lane < kQpN) pbuf[lane] = (lane < nv) ? LV(w) * ((lane >= t) ? cscale : 1.0) : 0.0;
        }
        WSYNC();
#pragma unroll
        for (int j = 0; j < kQpN; j++) out.x[j] = pbuf[j];
        WSYNC();
        // worst slack of the returned point, normalised by the unscaled row norm
        LANES {
            double d = 0.0, nr = 0.0;
#pragma unroll
            for (int j = 0; j < kQpN; j++) {
                const double a = (j < t) ? LV(R.g)[j] : LV(R.g)[j] * (1.0 / kQpScaleGI);
                d += a * out.x[j];
                nr += a * a;
            }
            nr = sqrt(nr);
            const double rn = 1.0 / (nr < 1e-300 ? 1e-300 : nr);
            const double sh = LV(R.hi) >= DWBC_QP_INF ? DWBC_QP_INF : (LV(R.hi) - d) * rn;
            const double sl = LV(R.lo) >= DWBC_QP_INF ? DWBC_QP_INF : (LV(R.lo) + d) * rn;
            LV(val) = sl < sh ? sl : sh;
            LV(key) = lane;
        }
        double wv;
        int wi;
        WAVE_ARGMIN(val, key, wv, wi);
        out.viol = wv >= DWBC_QP_INF ? 0.0 : wv;
        if (!weighted || !(wv < -kQpFeasTol)) break;
    }
}

}  // namespace dwbc

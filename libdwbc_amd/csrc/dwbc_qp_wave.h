// dwbc_qp_wave.h -- register-resident batched active-set QP, one wavefront per problem.
//
//   lexmin( 1/2|x[:t]|^2 , 1/2|x[t:]|^2 )  s.t.  -lo_r <= g_r . x <= hi_r      (H = diag(I_t, 0_k), g = 0)
//
// Replaces CQuadraticProgram::SolveQPoases (reference src/qp_wrapper.cpp:192-380) for the QPs assembled by
// CalcSingleTaskTorqueWithQP / CalcContactRedistribute (reference src/dwbc.cpp:988-1053, 1458-1517).
//
// Layout: lane r owns constraint row r in registers -- the 33 torque-limit rows are stored once as two-sided rows
// (+/-(row) <= tau_lim -/+ base), the 20 friction/CoP cone rows one-sided -- so 53 lanes hold the whole 86-row QP.
// Rows are normalised once; each lane carries g.x of its row and updates it with ONE FMA per step (g.z is needed for
// the step length anyway).  The Goldfarb-Idnani operators live in one 12-register array per lane: lanes 0..11 hold the
// rows of H = I - N N^+ (projector on the null space of the active normals), lanes 16..27 the rows of N^+, one working-
// set slot each.  So z = H n and r = N^+ n are a single 12-FMA dot product per lane against the normal broadcast with
// v_readlane, adding a constraint is one rank-one update M -= coef * z^T shared by both operators, and dropping one is
// the inverse rank-one update -- no refactorisation, no LDS round trip, no triangular solve in the loop.
// The final point (DESIGN.md "QP canon") comes from a column-pivoted Householder QR of the weighted working-set
// normals done in place in the owner lanes (no gather), fully unrolled over the elimination step.
#pragma once
#include <type_traits>

#include "dwbc_wave.h"

namespace dwbc {

constexpr int kQpN = 12;  // max variables (6 task + 6 contact-null)

// QN: capacity in variables -- kQpN for the product kernels (6 task + 6 contact-null), 18 for the three-contact kernel (dwbc_cycle_gc.h)
template <int QN>
struct QpRowsT {
    PLA(real_t, g, QN);    // row coefficients, contact columns already scaled by kQpScaleGI, zero padded
    PL(real_t, hi);        // g.x <= hi
    PL(real_t, lo);        // -g.x <= lo   (lo = +inf: one-sided row)
    PL(int, id_hi);        // reference row index of the hi side (for diagnostics)
    PL(int, id_lo);
};
using QpRows = QpRowsT<kQpN>;

template <int QN>
struct QpResultT {
    int status, iters, nact;
    real_t viol;
    real_t x[QN];     // uniform, unscaled [delta (t); c (k)]
    int act[QN];      // reference row indices of the working set
#if defined(DWBC_STAGE_TIMERS) && !defined(DWBC_HOST_EMU)
    long long tm[10];  // diagnostic build: cycles per solver section; [8] row fill, [9] (free)
#endif
};
using QpResult = QpResultT<kQpN>;

#if defined(DWBC_STAGE_TIMERS) && !defined(DWBC_HOST_EMU)
#define DWBC_QPT_INIT() long long qpt_last_ = clock64(); for (int i_ = 0; i_ < 10; i_++) out.tm[i_] = 0
#define DWBC_QPT(i) do { const long long now_ = clock64(); out.tm[i] += now_ - qpt_last_; qpt_last_ = now_; } while (0)
#else
#define DWBC_QPT_INIT() ((void)0)
#define DWBC_QPT(i) ((void)0)
#endif

#ifdef DWBC_QP_TRACE
#ifdef DWBC_HOST_EMU
#define QP_TRACE(...) printf(__VA_ARGS__)
#else
#define QP_TRACE(...) do { if (threadIdx.x == 0) printf(__VA_ARGS__); } while (0)
#endif
#endif
#define DWBC_QP_INF (dwbc::kF32 ? dwbc::real_t(1.0e30) : dwbc::real_t(1.0e300))



// WS != 0: report the working set (reference row indices) in out.act -- diagnostics the lean kernel build leaves out.
// NV >= nv: compile-time bound of the variable count (12, 9 or 6); every dot product, rank-one update and QR step runs over NV
// entries instead of the maximum 12 (the padded entries are exact zeros, so the result does not depend on NV).
// KCV: contact-null variables of the fixed layout (QN - KCV task variables first): 6 in the product kernels; the general-contact kernel
// passes its own (12 with three contacts, whatever the task block is sized for)
template <int WS, int NV = kQpN, int QN = kQpN, int KCV = (QN > 12 ? QN - 6 : 6)>
// warm: reference row indices of the previous solve's working set (kQpN entries, -1 = empty) or nullptr.  Hot start in the sense of
// SolveQPoases(init = false): the search visits those rows first -- each is added as soon as it is violated -- before it falls
// back to the most-violated rule.  The Tikhonov problem is strictly convex, so the point the search ends at does not depend on
// the order of the picks; only the path (and the iteration count, when the cold path adds rows it later drops) does.
// vtol: violation (slack / |row|) below which a row counts as satisfied during the search: kQpTol for the task QPs, kQpFeasTol for the
// contact redistribution QP -- it starts from the point the last task QP handed over, which was accepted at kQpFeasTol (canon rule 5)
// sfin (per lane): the UNNORMALISED slack of the lane's row (its tighter side) at the point returned in out.x -- the caller uses it to see
// whether the contact redistribution QP that follows the last task level has anything to do (dwbc_cycle2.h)
DWBC_WDEV void qp_solve_wave(QpRowsT<QN> &R, int nv, int t, int max_iter, QpResultT<QN> &out, real_t *V /* LDS, QN doubles */, const int *warm,
                             real_t vtol, PL_REF(real_t, sfin)) {
    DWBC_LANE_DECL;
    static_assert(NV <= QN && NV <= 24, "variable count");
    constexpr int SB = NV > 16 ? 32 : 16;  // first working-set slot lane (the H rows take lanes 0..NV-1)
    const int k = nv - t;
    PLA(real_t, Mx, NV);  // lanes 0..11: row of H;  lanes 16..27: row of N^+ for working-set slot lane-16
    PL(real_t, d);          // g . x of the own row (normalised row, scaled variables)
    PL(real_t, fs);         // |g| / |a|: factor from the normalised slack to slack / (norm of the unscaled row)
    PL(real_t, gnv);        // |g| before the normalisation (1 for a zero row): normalised slack * gnv = slack of the row as given
    PL(real_t, u);          // slot lanes: multiplier
    PL(int, akey);          // slot lanes: (owner lane << 1) | side
    PL(int, actf);          // bit0: hi side in the working set, bit1: lo side
    PL(int, slotbit);       // 1 << slot for lanes 16..27, else 0
    PL(real_t, val);
    PL(int, key);
    PL(real_t, m);
    PL(real_t, dz);
    PL(real_t, xl);         // lanes 0..NV-1: the iterate, one variable per lane (z = H n lives in the same lanes: one FMA per step)
    DWBC_QPT_INIT();
    LANES {
        LV(xl) = real_t(0.0);
        DWBC_LANE_OPAQUE(lq);
        real_t s2 = real_t(0.0), a2 = real_t(0.0);
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const real_t g2 = LV(R.g)[j] * LV(R.g)[j];
            s2 += g2;
            a2 += (j < t) ? g2 : g2 * (real_t(1.0) / (kQpScaleGI * kQpScaleGI));
        }
        // a numerically zero row (|g| < kQpZeroRow) is the constraint 0 <= hi: its coefficients are dropped and its
        // slack is taken unnormalised, so noise never becomes a unit normal (oracle: QP_ZERO_ROW)
        const bool zrow = s2 < kQpZeroRow * kQpZeroRow;
        const real_t gn = zrow ? real_t(1.0) : sqrt(s2);
        const real_t rg = zrow ? real_t(0.0) : real_t(1.0) / gn;
        LV(fs) = zrow ? real_t(1.0) : gn / sqrt(a2);
        LV(gnv) = gn;
#pragma unroll
        for (int j = 0; j < NV; j++) {
            LV(R.g)[j] *= rg;
            LV(Mx)[j] = (lq == j) ? real_t(1.0) : real_t(0.0);
        }
        if (LV(R.hi) < DWBC_QP_INF && !zrow) LV(R.hi) *= rg;
        if (LV(R.lo) < DWBC_QP_INF && !zrow) LV(R.lo) *= rg;
        LV(d) = real_t(0.0);
        LV(u) = real_t(0.0);
        LV(akey) = 0;
        LV(actf) = 0;
        LV(slotbit) = (lane >= SB && lane < SB + NV) ? (1 << (lane - SB)) : 0;
        LV(m) = real_t(0.0);
        LV(dz) = real_t(0.0);
    }
    DWBC_QPT(6);  // rows normalised, operators initialised
    // ---- Goldfarb-Idnani dual active set on min 1/2 |x|^2 (scaled variables), x = 0 start
    int used = 0, q = 0, it = 0, status = 1, p = 0, side = 0, kmin = 0;
    bool pick = true;
    real_t up = real_t(0.0), worst = real_t(0.0);
    int wid[QN], wpos = QN;
    if (WS && warm) {
#pragma unroll
        for (int a = 0; a < QN; a++) wid[a] = warm[a];
        wpos = 0;
    }
    for (;;) {
        bool got = false;
        if (WS && pick) {
            while (wpos < QN && !got) {  // rows of the previous working set first
                int id = -1;
#pragma unroll
                for (int a = 0; a < QN; a++) id = (a == wpos) ? wid[a] : id;
                wpos++;
                if (id < 0) continue;
                LANES {
                    const bool mh = LV(R.id_hi) == id && !(LV(actf) & 1) && LV(R.hi) < DWBC_QP_INF;
                    const bool ml = LV(R.id_lo) == id && !(LV(actf) & 2) && LV(R.lo) < DWBC_QP_INF;
                    LV(val) = mh ? LV(R.hi) - LV(d) : (ml ? LV(R.lo) + LV(d) : DWBC_QP_INF);
                    LV(key) = (lane << 1) | (ml ? 1 : 0);
                }
                int pl;
                WAVE_ARGMIN_F32(val, pl);
                worst = BCAST(val, pl);
                kmin = BCASTI(key, pl);
                if (worst < -vtol) got = true;
            }
            if (got) {
                p = kmin >> 1;
                side = kmin & 1;
                up = real_t(0.0);
                pick = false;
            }
        }
        if (pick) {
            LANES {
                const real_t sh = ((LV(actf) & 1) || LV(R.hi) >= DWBC_QP_INF) ? DWBC_QP_INF : LV(R.hi) - LV(d);
                const real_t sl = ((LV(actf) & 2) || LV(R.lo) >= DWBC_QP_INF) ? DWBC_QP_INF : LV(R.lo) + LV(d);
                const bool lo_side = sl < sh;
                LV(val) = lo_side ? sl : sh;
                LV(key) = (lane << 1) | (lo_side ? 1 : 0);
            }
            int pl;
            WAVE_ARGMIN_F32(val, pl);
            worst = BCAST(val, pl);
            kmin = BCASTI(key, pl);
            DWBC_QPT(1);
            if (!(worst < -vtol)) break;
            p = kmin >> 1;
            side = kmin & 1;
            up = real_t(0.0);
            pick = false;
        }
        if (++it > max_iter) { status = 0; break; }
        // normal of the violated side (GI is stated for n^T x >= b': hi side -> n = -g, lo side -> n = +g)
        real_t gp[NV];
#pragma unroll
        for (int j = 0; j < NV; j++) gp[j] = BCASTA(R.g, j, p);
        const real_t sgn = side ? real_t(1.0) : -real_t(1.0);
        LANES {
            real_t s_ = real_t(0.0);
#pragma unroll
            for (int j = 0; j < NV; j++) s_ += LV(Mx)[j] * gp[j];
            LV(m) = sgn * s_;  // lanes 0..11: z = H n;  slot lanes: r = N^+ n
        }
        real_t zu[NV];
#pragma unroll
        for (int i = 0; i < NV; i++) zu[i] = BCAST(m, i);
        LANES {
            real_t s_ = real_t(0.0);
#pragma unroll
            for (int j = 0; j < NV; j++) s_ += LV(R.g)[j] * zu[j];
            LV(dz) = s_;  // change of g.x per unit step along z
            LV(val) = side ? LV(R.lo) + LV(d) : LV(R.hi) - LV(d);
        }
        real_t zg = sgn * BCAST(dz, p);  // n.z = |z|^2
        // z = H n loses digits when n lies close to the span of the working set (the rows are unit vectors, so |z|^2 is the squared
        // sine of that angle): H is kept by rank-one updates, i.e. classical Gram-Schmidt.  "Twice is enough": project once more.
        // Without it the final point carries ~1e-9 relative round-off on tilted feet (3e-7 Nm against the oracle instead of 2e-8);
        // about one step in five takes this branch.
        if (zg < kQpReorth) {
            LANES {
                real_t s_ = real_t(0.0);
#pragma unroll
                for (int j = 0; j < NV; j++) s_ += LV(Mx)[j] * zu[j];
                if (lane < NV) LV(m) = s_;  // (the slot lanes keep r = N^+ n)
            }
#pragma unroll
            for (int i = 0; i < NV; i++) zu[i] = BCAST(m, i);
            LANES {
                real_t s_ = real_t(0.0);
#pragma unroll
                for (int j = 0; j < NV; j++) s_ += LV(R.g)[j] * zu[j];
                LV(dz) = s_;
            }
            zg = sgn * BCAST(dz, p);
        }
        const real_t sp = BCAST(val, p);       // slack of the violated side (negative)
        const bool zok = zg > (kF32 ? real_t(1e-10) : real_t(1e-20)) && q < nv;
        DWBC_QPT(2);
        LANES {
            const bool ok = (LV(slotbit) & used) && LV(m) > (kF32 ? real_t(1e-6) : real_t(1e-12));
            LV(val) = ok ? LV(u) * fast_rcp(LV(m)) : DWBC_QP_INF;
            LV(key) = lane;
        }
        real_t t1;
        int l;
        if constexpr (SB == 16) {
            WAVE_ARGMIN_ROW1(val, key, t1, l);  // (the slot lanes are DPP row 1)
        } else {
            WAVE_ARGMIN(val, key, t1, l);       // every other lane holds +inf
        }
        const real_t t2 = zok ? -sp * fast_rcp(zg) : DWBC_QP_INF;
        const real_t tstep = t1 < t2 ? t1 : t2;
        if (!(tstep < DWBC_QP_INF)) { status = 0; break; }
        const bool full = zok && t2 <= t1;
        if (zok) {
            LANES {
                LV(d) += tstep * LV(dz);
                LV(xl) += tstep * LV(m);  // (lanes 0..NV-1: m = z; the value in the other lanes is not used)
            }
        }
        LANES { LV(u) -= tstep * LV(m); }
        up += tstep;
        DWBC_QPT(3);
        if (full) {
            // working set += (p, side): H -= z z^T / zg, N^+ rows -= r_a z^T / zg, new N^+ row = z^T / zg
            int slot = 0;
#pragma unroll
            for (int a = NV - 1; a >= 0; a--) slot = ((used >> a) & 1) ? slot : a;
            const real_t inv_ = fast_rcp(zg);
            LANES {
                real_t coef = LV(m) * inv_;
                if (lane == SB + slot) { coef = -inv_; LV(u) = up; LV(akey) = kmin; }
#pragma unroll
                for (int j = 0; j < NV; j++) LV(Mx)[j] -= coef * zu[j];
                if (lane == p) LV(actf) |= (side ? 2 : 1);
            }
            used |= 1 << slot;
            q++;
            pick = true;
        } else {
            // drop slot lane l: with rho = its N^+ row, H += rho rho^T / |rho|^2, other N^+ rows lose their rho part
            const int kl = BCASTI(akey, l);
            WSYNC();
            LANES {
                if (lane == l) {
#pragma unroll
                    for (int j = 0; j < NV; j++) V[j] = LV(Mx)[j];
                }
            }
            WSYNC();
            real_t rho[NV], rr = real_t(0.0);
#pragma unroll
            for (int j = 0; j < NV; j++) { rho[j] = V[j]; rr += rho[j] * rho[j]; }
            const real_t inv_ = fast_rcp(rr);
            LANES {
                real_t dd = real_t(0.0);
#pragma unroll
                for (int j = 0; j < NV; j++) dd += LV(Mx)[j] * rho[j];
                real_t coef = dd * inv_;
                if (lane < NV) coef = -V[lane] * inv_;
#pragma unroll
                for (int j = 0; j < NV; j++) LV(Mx)[j] = (lane == l) ? real_t(0.0) : LV(Mx)[j] - coef * rho[j];
                if (lane == l) LV(u) = real_t(0.0);
                if (lane == (kl >> 1)) LV(actf) &= ~((kl & 1) ? 2 : 1);
            }
            WSYNC();
            used &= ~(1 << (l - SB));
            q--;
        }
        DWBC_QPT(4);
    }
    DWBC_QPT(0);
    out.iters = it;
    out.nact = q;
    out.status = status;
    out.viol = worst >= DWBC_QP_INF ? real_t(0.0) : worst;
#pragma unroll
    for (int a = 0; a < QN; a++) {
        if (WS) {
            const int ka = BCASTI(akey, SB + a);
            const int ow = ka >> 1;
            const int idh = BCASTI(R.id_hi, ow), idl = BCASTI(R.id_lo, ow);
            out.act[a] = ((used >> a) & 1) ? ((ka & 1) ? idl : idh) : -1;
        } else {
            out.act[a] = -1;
        }
    }
#pragma unroll
    for (int i = 0; i < QN; i++) out.x[i] = real_t(0.0);
    // slack of every row at the returned point: gx = g . x of the lane's (normalised) row there
    auto final_slack = [&](int which) {  // 0: x = 0, 1: the GI iterate (d), 2: the lexicographic point (dz holds g . x of it)
        LANES {
            const real_t gx = which == 0 ? real_t(0.0) : (which == 1 ? LV(d) : LV(dz));
            const real_t sh = LV(R.hi) >= DWBC_QP_INF ? DWBC_QP_INF : (LV(R.hi) - gx) * LV(gnv);
            const real_t sl = LV(R.lo) >= DWBC_QP_INF ? DWBC_QP_INF : (LV(R.lo) + gx) * LV(gnv);
            LV(sfin) = sl < sh ? sl : sh;
        }
    };
    if (!status || q == 0) { final_slack(0); return; }  // x = 0: failure (caller zeroes the correction) or no active constraint
    real_t xu[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) xu[i] = BCAST(xl, i);
    // Tikhonov point on the working set = the GI iterate (the fallback of the canon, and the answer when one of the two
    // variable blocks is empty)
#pragma unroll
    for (int j = 0; j < QN; j++) out.x[j] = (j < nv) ? xu[j < NV ? j : 0] * ((j >= t) ? kQpScaleGI : real_t(1.0)) : real_t(0.0);
    if (!(k > 0 && t > 0)) { final_slack(1); return; }
    // ---- lexicographic least-norm point on the working set (min |delta| first, then min |c|) from the operators the search
    //      already holds.  With N the active normals (scaled variables) the GI iterate is x^ = N^+T b, the minimiser of
    //      1/2 |delta|^2 + 1/2 |c^|^2 on the working set; the minimiser of 1/2 |delta|^2 + 1/2 |c^ - c^_k|^2 on it is
    //      x_{k+1} = x^ + H [0; c^_k]  (H = I - N N^+, rows in lanes 0..NV-1).  From c^_0 = 0 this proximal-point sequence
    //      converges to the minimiser of |delta| whose c is closest to 0 -- the lexicographic point -- i.e. to the solution of
    //            (I - H_cc) c^ = c^_1        (H_cc: the contact block of the projector; symmetric, eigenvalues in [0, 1]),
    //      then delta = delta^ + H_dc c^.  On a well-posed contact block I - H_cc is the identity up to
    //      |g_delta|^2 / (s^2 |g_c|^2) ~ 1e-8 (s = kQpScaleGI); tilted feet bring eigenvalues down to 1e-2.  Conjugate gradients on
    //      the 6 x 6 system -- one 6-term dot product per lane and 6 broadcasts per step, the vector recurrences on uniform data --
    //      end in one or two steps in the first case and in at most k steps in any.  A residual that has not settled by then is
    //      the block on which the lexicographic point blows up (DESIGN.md "QP canon" rule 3): the Tikhonov point stands.
    //      (Replaces the column-pivoted Householder QR of the weighted normals of rounds 1-2, ~13 k cycles per QP.)
    {
        real_t xs[NV];
        constexpr int KC = KCV;  // contact-null variables: k <= 6 (one or two 6D contacts); 12 in the three-contact build
        constexpr int kCgMax = KC > kQpRefine ? 4 * KC : kQpRefine;
        static_assert(NV >= KC, "variable blocks");
        bool settled = false;
        // STD: the layout every lean launch has when it gets here (t = NV - 6 task variables, then 6 contact-null ones): the
        // contact block sits at compile-time positions.  Any other (t, k) -- TASK_CUSTOM levels of the full build -- picks its
        // entries with uniform selects.
        auto lex_point = [&](auto std_tag) {
            constexpr bool STD = decltype(std_tag)::value;
            real_t cx[KC], cr[KC], cp[KC], cb[KC];
#pragma unroll
            for (int i = 0; i < KC; i++) {
                real_t v = real_t(0.0);
                if constexpr (STD) {
                    v = xu[NV - KC + i];
                } else {
#pragma unroll
                    for (int j = 0; j < NV; j++) v = (j == t + i && i < k) ? xu[j] : v;
                }
                cb[i] = v;    // c^_1
                cx[i] = v;    // start at the Tikhonov point: I - H_cc ~ I
            }
            // one product with H: lanes 0..NV-1 get (H [0; v])_lane
            auto hmul = [&](const real_t (&v)[KC]) {
                LANES {
                    real_t s_ = real_t(0.0);
                    if constexpr (STD) {
#pragma unroll
                        for (int i = 0; i < KC; i++) s_ += LV(Mx)[NV - KC + i] * v[i];
                    } else {
#pragma unroll
                        for (int j = 0; j < NV; j++) {
                            real_t vj = real_t(0.0);
#pragma unroll
                            for (int i = 0; i < KC; i++) vj = (j == t + i && i < k) ? v[i] : vj;
                            s_ += LV(Mx)[j] * vj;
                        }
                    }
                    LV(m) = s_;
                }
            };
            auto hcc = [&](int i) -> real_t {  // entry i of the contact block of the last product
                real_t h_;
                if constexpr (STD) h_ = BCAST(m, NV - KC + i);
                else h_ = (i < k) ? BCAST(m, t + i) : real_t(0.0);
                return h_;
            };
            hmul(cx);
            real_t rs = real_t(0.0), bn = real_t(0.0);
#pragma unroll
            for (int i = 0; i < KC; i++) {
                cr[i] = cb[i] - (cx[i] - hcc(i));  // r = b - (I - H_cc) x
                cp[i] = cr[i];
                rs += cr[i] * cr[i];
                bn += cb[i] * cb[i];
            }
            const real_t rtol2 = kQpRefineTol * kQpRefineTol * bn;
            bool nulldir = false;
            const real_t atol2 = kQpNullRes * kQpNullRes * bn;  // residual (squared) at which a solve that cannot go on stands
            for (int r = 0; r < kCgMax && rs > rtol2; r++) {
#ifdef DWBC_QP_TRACE
                QP_TRACE("  cg %d rs %.3e\n", r, (double)rs);
#endif
                hmul(cp);
                real_t ap[KC], pap = real_t(0.0), pp = real_t(0.0);
#pragma unroll
                for (int i = 0; i < KC; i++) {
                    ap[i] = cp[i] - hcc(i);
                    pap += cp[i] * ap[i];
                    pp += cp[i] * cp[i];
                }
                // A search direction on which I - H_cc has (numerically) no curvature is a direction of c the working set does not
                // constrain -- a working set whose contact block has lost rank, seen as soon as a HAND is in contact (twelve active rows
                // of rank 11 on two feet and a hand, six of rank 5 on a foot and a hand): a step along it would divide round-off by
                // round-off (c of 1e7).  The lexicographic point has no component there (stage 2 is the least-norm c; the restatement
                // truncates the same direction by rank, oracle _lex_eqp), and neither has the start: stop here.  (pap <= 0: the same
                // case seen through round-off.)
                if (!(pap > kQpNullDir * pp)) { nulldir = true; break; }
                const real_t al = rs * fast_rcp(pap);
                // a step along a direction of small but not negligible curvature (1e-8 seen) can throw away ten digits of an iterate
                // that is already good enough: such a step is not taken
                real_t rs2 = real_t(0.0);
#pragma unroll
                for (int i = 0; i < KC; i++) {
                    const real_t rn = cr[i] - al * ap[i];
                    rs2 += rn * rn;
                }
                if (rs2 > rs && !(rs > atol2)) { nulldir = true; break; }
#pragma unroll
                for (int i = 0; i < KC; i++) {
                    cx[i] += al * cp[i];
                    cr[i] -= al * ap[i];
                }
                const real_t be = rs2 * fast_rcp(rs);
#pragma unroll
                for (int i = 0; i < KC; i++) cp[i] = cr[i] + be * cp[i];
                rs = rs2;
            }
            // a solve that stopped at such a direction (or ran out of steps around it) has reached what the arithmetic allows; it stands
            // if its relative residual is below kQpNullRes (seen: 2e-10 on a foot + a hand)
            settled = !(rs > rtol2) || !(rs > atol2);
            (void)nulldir;
#ifdef DWBC_QP_TRACE
            QP_TRACE("lex: NV %d KC %d t %d k %d q %d rs %.3e rtol2 %.3e bn %.3e settled %d\n", NV, KC, t, k, q, (double)rs, (double)rtol2, (double)bn, (int)settled);
#endif
            hmul(cx);  // x = x^ + H [0; c^]  (its contact block reproduces c^ when the residual is zero)
#ifdef DWBC_QP_TRACE
            {
                double tr_ = 0.0;
                for (int i = 0; i < KC; i++) { const double r_ = (double)(cb[i] - (cx[i] - hcc(i))); tr_ += r_ * r_; }
                QP_TRACE("  true residual^2 %.3e (recursive %.3e)\n", tr_, (double)rs);
            }
#endif
#pragma unroll
            for (int i = 0; i < NV; i++) xs[i] = xu[i] + BCAST(m, i);
        };
        if (!WS || (t == NV - KC && k == KC)) lex_point(std::true_type{}); else lex_point(std::false_type{});
        DWBC_QPT(5);
        // worst slack of the lexicographic point, normalised by the unscaled row norm
        LANES {
            real_t dd = real_t(0.0);
#pragma unroll
            for (int j = 0; j < NV; j++) dd += LV(R.g)[j] * xs[j];
            LV(dz) = dd;  // (dz is free after the search: g . x of the lexicographic point, for final_slack)
            const real_t sh = LV(R.hi) >= DWBC_QP_INF ? DWBC_QP_INF : (LV(R.hi) - dd) * LV(fs);
            const real_t sl = LV(R.lo) >= DWBC_QP_INF ? DWBC_QP_INF : (LV(R.lo) + dd) * LV(fs);
            LV(val) = sl < sh ? sl : sh;
        }
        int wi;
        WAVE_ARGMIN_F32(val, wi);
        const real_t wv = BCAST(val, wi);
        DWBC_QPT(7);
#ifdef DWBC_QP_TRACE
        QP_TRACE("  lex worst slack %.3e at lane %d -> %s;  xs:", (double)wv, wi, (settled && !(wv < -kQpFeasTol)) ? "lex" : "tikhonov");
        for (int j = 0; j < NV; j++) QP_TRACE(" %.6e", (double)(xs[j] * ((j >= t) ? kQpScaleGI : real_t(1.0))));
        QP_TRACE("\n  xu:");
        for (int j = 0; j < NV; j++) QP_TRACE(" %.6e", (double)(xu[j] * ((j >= t) ? kQpScaleGI : real_t(1.0))));
        QP_TRACE("\n  working set (owner lane, side):");
        for (int a = 0; a < NV; a++) if ((used >> a) & 1) { const int ka = BCASTI(akey, SB + a); QP_TRACE(" (%d,%d)", ka >> 1, ka & 1); }
        QP_TRACE("\n");
#endif
        if (settled && !(wv < -kQpFeasTol)) {
            out.viol = wv >= DWBC_QP_INF ? real_t(0.0) : wv;
#pragma unroll
            for (int j = 0; j < QN; j++) out.x[j] = (j < nv) ? xs[j < NV ? j : 0] * ((j >= t) ? kQpScaleGI : real_t(1.0)) : real_t(0.0);
            final_slack(2);
        } else {
            final_slack(1);
        }
    }
}

}  // namespace dwbc

// dwbc_nohqp.h -- CalcContactRedistribute(hqp = false): the closed-form two-contact redistribution
// (reference src/dwbc.cpp:1570-1619 -> ContactRedistributetwomod, src/wbd.cpp:273-404).  SURVEY.md §8 row f4.
//
// The resultant wrench of the two feet about the centre of mass is split with one scalar eta (share of contact 1), chosen
// from the quadratic CoP / yaw-friction bounds; the joint torque that shifts contact 2's wrench to its new share is
// V (J̄_2 V)^-1 dF_2 with V any basis of null(W) -- the reference takes Eigen's V2, here NwJw (same subspace, the expression
// does not depend on the basis).  link_[0].rpy(2) is the third angle of Eigen's eulerAngles(2, 1, 0) [ext], i.e. the rotation
// about X, which the reference feeds to rotateWithZ (dwbc.cpp:1580) -- kept as is.
// Runs after the task cascade (tg, tt filled); writes torque_contact_ into L[S::tc ..).  Returns the reference's int.
#pragma once
#include "dwbc_cycle.h"

namespace dwbc {

DWBC_DEV void eta_bound(real_t A, real_t B, real_t C, real_t &lb, real_t &ub) {
    const real_t a = A * A, b = real_t(2.0) * A * B, c = B * B - C * C;
    const real_t disc = sqrt(b * b - real_t(4.0) * a * c);
    const real_t s1 = (-b + disc) / real_t(2.0) / a, s2 = (-b - disc) / real_t(2.0) / a;
    const real_t hi = s1 > s2 ? s1 : s2, lo = s1 > s2 ? s2 : s1;
    if (hi < ub) ub = hi;
    if (lo > lb) lb = lo;
}

template <class S, int N, int NB, int NT>
DWBC_DEV int redistribute_closed_form(Thr th, real_t *L, const real_t *JbT, int cd, int k) {
    constexpr int M = S::M;
    if (cd != 12) {  // dwbc.cpp:1612-1617
        for (int i = th.tid; i < M; i += NT) L[S::tc + i] = real_t(0.0);
        DWBC_SYNC();
        return 0;
    }
    real_t *cf = L + S::t_fv;                   // ContactForce_ (12)
    real_t *X = L + S::t_F;                     // J̄_2 NwJw (6 x 6), then its inverse at X + 36, GJ scratch at X + 72
    real_t *des = L + S::t_base;                // desired_force[6:12]
    DWBC_SYNC();
    for (int i = th.tid; i < 12; i += NT) {
        real_t acc = -L[S::PC + i];
        for (int c = 0; c < M; c++) acc += JbT[i * N + 6 + c] * (L[S::tg + c] + L[S::tt + c]);
        cf[i] = acc;
    }
    // J̄[6:12, 6:] NwJw from FNl = A_rot (J̄[:,6:] NwJw): rotate contact 2's rows back to the world frame
    for (int idx = th.tid; idx < 36; idx += NT) {
        const int i = idx / 6, j = idx - i * 6, h = i / 3, y = i % 3;
        const real_t *R = L + S::Rc + 9;
        real_t acc = real_t(0.0);
        for (int x = 0; x < 3; x++) acc += R[y * 3 + x] * L[S::FNl + (6 + 3 * h + x) * k + j];
        X[idx] = acc;
    }
    DWBC_SYNC();
    if (th.tid == 0) {
        const real_t *R0 = L + S::Rw, *com = L + S::comp, *Pc = L + S::Pc;
        // eulerAngles(2, 1, 0)[2] (Eigen >= 3.3 [ext]); only the first angle's branch matters for the third one
        real_t r0 = atan2(R0[3], R0[0]);
        if (r0 < real_t(0.0)) r0 += real_t(3.14159265358979323846);
        const real_t s1 = sin(r0), c1 = cos(r0);
        const real_t roll = atan2(s1 * R0[2] - c1 * R0[5], c1 * R0[4] - s1 * R0[1]);
        const real_t cy = cos(-roll), sy = sin(-roll);  // Rotyaw = rotateWithZ(-rpy(2))
        real_t F12[12], P1[3], P2[3];
        for (int b = 0; b < 4; b++) {
            F12[3 * b] = cy * cf[3 * b] - sy * cf[3 * b + 1];
            F12[3 * b + 1] = sy * cf[3 * b] + cy * cf[3 * b + 1];
            F12[3 * b + 2] = cf[3 * b + 2];
        }
        {
            const real_t d1[3] = {Pc[0] - com[0], Pc[1] - com[1], Pc[2] - com[2]}, d2[3] = {Pc[3] - com[0], Pc[4] - com[1], Pc[5] - com[2]};
            P1[0] = cy * d1[0] - sy * d1[1]; P1[1] = sy * d1[0] + cy * d1[1]; P1[2] = d1[2];
            P2[0] = cy * d2[0] - sy * d2[1]; P2[1] = sy * d2[0] + cy * d2[1]; P2[2] = d2[2];
        }
        // ContactRedistributetwomod(0.99, 0.26, 0.1, 1.0, 0.9, 0.9, ...) (dwbc.cpp:1600)
        const real_t eta_cust = real_t(0.99), footlength = real_t(0.26), footwidth = real_t(0.1), mu_s = real_t(1.0), ratio = real_t(0.9);
        real_t Rf[6];
        for (int a = 0; a < 3; a++) Rf[a] = F12[a] + F12[6 + a];
        Rf[3] = F12[3] + F12[9] + (P1[1] * F12[2] - P1[2] * F12[1]) + (P2[1] * F12[8] - P2[2] * F12[7]);
        Rf[4] = F12[4] + F12[10] + (P1[2] * F12[0] - P1[0] * F12[2]) + (P2[2] * F12[6] - P2[0] * F12[8]);
        Rf[5] = F12[5] + F12[11] + (P1[0] * F12[1] - P1[1] * F12[0]) + (P2[0] * F12[7] - P2[1] * F12[6]);
        const real_t d0 = P1[0] - P2[0], d1 = P1[1] - P2[1], d2 = P1[2] - P2[2];
        const real_t A3 = d2 * Rf[1] - d1 * Rf[2], B3 = Rf[3] + P2[2] * Rf[1] - P2[1] * Rf[2];
        const real_t A4 = -d2 * Rf[0] + d0 * Rf[2], B4 = Rf[4] - P2[2] * Rf[0] + P2[0] * Rf[2];
        const real_t A5 = -d0 * Rf[1] + d1 * Rf[0], B5 = Rf[5] + P2[1] * Rf[0] - P2[0] * Rf[1];
        real_t lb = real_t(1.0) - eta_cust, ub = eta_cust;
        eta_bound(A3, B3, ratio * footwidth / real_t(2.0) * fabs(Rf[2]), lb, ub);
        eta_bound(A4, B4, ratio * footlength / real_t(2.0) * fabs(Rf[2]), lb, ub);
        eta_bound(A5, B5, mu_s * fabs(Rf[2]), lb, ub);
        const real_t eta_s = -B3 / A3;
        real_t eta = eta_s;
        if (eta_s > ub) eta = ub;
        else if (eta_s < lb) eta = lb;
        if (!((eta <= eta_cust) && (eta >= real_t(1.0) - eta_cust))) eta = real_t(0.5);  // also catches NaN
        const real_t om = real_t(1.0) - eta;
        const real_t red[6] = {om * Rf[0], om * Rf[1], om * Rf[2], om * (A3 * eta + B3), om * (A4 * eta + B4), om * (A5 * eta + B5)};  // contact 2
        // fc_redist_ = force_rot_yaw^T ResultRedistribution_ ; desired_force[6:12] = -ContactForce_[6:12] + fc_redist_[6:12]
        for (int b = 0; b < 2; b++) {
            des[3 * b] = -cf[6 + 3 * b] + (cy * red[3 * b] + sy * red[3 * b + 1]);
            des[3 * b + 1] = -cf[6 + 3 * b + 1] + (-sy * red[3 * b] + cy * red[3 * b + 1]);
            des[3 * b + 2] = -cf[6 + 3 * b + 2] + red[3 * b + 2];
        }
    }
    DWBC_SYNC();
    gj_inverse<NT>(th, X, 6, 6, X + 36, 6, X + 72);
    for (int i = th.tid; i < M; i += NT) {
        real_t acc = real_t(0.0);
        for (int j = 0; j < 6; j++) {
            real_t z = real_t(0.0);
            for (int c = 0; c < 6; c++) z += X[36 + j * 6 + c] * des[c];
            acc += L[S::NwJw + i * k + j] * z;
        }
        L[S::tc + i] = acc;  // torque_contact_ = V2^T (J̄_2 V2^T)^-1 desired_force[6:12]  (dwbc.cpp:1608)
    }
    DWBC_SYNC();
    return 1;
}

}  // namespace dwbc

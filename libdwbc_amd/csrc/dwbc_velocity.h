// dwbc_velocity.h -- velocity-dependent outputs of RobotData::UpdateKinematics (reference src/dwbc.cpp:279-371):
//   link_[i].v / .w  (Link::UpdatePos, src/link.cpp:85-88: RBDL CalcPointVelocity6D at the link origin)
//   B_ = C(q, qdot) qdot + g(q)  (src/dwbc.cpp:343-344: RBDL NonlinearEffects = recursive Newton-Euler, zero qddot)
// None of them feeds the OSF torque path; they are produced into the dump record when the caller handed a
// qdot to dwbc_batch_set_state and enabled the dump, and they feed the on-device task reference (dwbc_fstar.h).
//
// World-frame spatial algebra about the pelvis origin O (the same frame as the motion subspaces S_j of stage 0,
// [angular; linear]): V_b = sum_{j on the path} S_j qd_j, bias acceleration a_b = a_g + sum_{path} (V_body(j) x S_j) qd_j,
// f_b = I_b a_b + V_b x* (I_b V_b), tau_j = S_j . sum_{subtree(j)} f.  Gravity enters as the fictitious base
// acceleration a_g = (0; 0, 0, +g).  Runs right after stage 0 (needs k_S, Rw, pw; overwrites k_Rl, k_Iw, k_F).
#pragma once
#include "dwbc_cycle.h"

namespace dwbc {

// h = I V for the rigid body (mass m, COM c relative to O, inertia Ic about the COM, world axes): [moment about O; force]
DWBC_DEV void spatial_inertia_apply(real_t m, const real_t *c, const real_t *Ic, const real_t *V, real_t *h) {
    const real_t *w = V, *v = V + 3;
    const real_t u0 = v[0] + (w[1] * c[2] - w[2] * c[1]), u1 = v[1] + (w[2] * c[0] - w[0] * c[2]), u2 = v[2] + (w[0] * c[1] - w[1] * c[0]);
    const real_t p0 = m * u0, p1 = m * u1, p2 = m * u2;  // linear momentum
    h[3] = p0; h[4] = p1; h[5] = p2;
    h[0] = Ic[0] * w[0] + Ic[1] * w[1] + Ic[2] * w[2] + (c[1] * p2 - c[2] * p1);
    h[1] = Ic[3] * w[0] + Ic[4] * w[1] + Ic[5] * w[2] + (c[2] * p0 - c[0] * p2);
    h[2] = Ic[6] * w[0] + Ic[7] * w[1] + Ic[8] * w[2] + (c[0] * p1 - c[1] * p0);
}

// Vout (nb x 6, [w; v_O]) stays in L + S::k_Rl for the caller (task-reference front end).
template <class S, int N, int NB, int NT>
DWBC_DEV void velocity_rnea(Thr th, const Setup &su, const io_t *qd, const real_t *body, const int *topo, real_t *L, real_t *Bout,
                            real_t *link_v, real_t *link_w) {
    const int nb = su.nb;
    const real_t *Sm = L + S::k_S, *Rw = L + S::Rw, *pw = L + S::pw;
    real_t *V = L + S::k_Rl, *Ab = L + S::k_Iw, *Cj = L + S::k_F;
    DWBC_SYNC();
    for (int b = th.tid; b < nb; b += NT) {
        real_t v[6] = {0, 0, 0, 0, 0, 0};
        for (int j = 0; j < 6; j++)
            for (int a = 0; a < 6; a++) v[a] += Sm[j * 6 + a] * (real_t)qd[j];
        for (int c = b; c > 0; c = su.parent[c])
            for (int a = 0; a < 6; a++) v[a] += Sm[(c + 5) * 6 + a] * (real_t)qd[c + 5];
        for (int a = 0; a < 6; a++) V[b * 6 + a] = v[a];
        if (link_w) {
            const real_t d0 = pw[b * 3] - pw[0], d1 = pw[b * 3 + 1] - pw[1], d2 = pw[b * 3 + 2] - pw[2];
            link_w[b * 3] = v[0]; link_w[b * 3 + 1] = v[1]; link_w[b * 3 + 2] = v[2];
            link_v[b * 3] = v[3] + (v[1] * d2 - v[2] * d1);      // v_O + w x (p_b - O)
            link_v[b * 3 + 1] = v[4] + (v[2] * d0 - v[0] * d2);
            link_v[b * 3 + 2] = v[5] + (v[0] * d1 - v[1] * d0);
        }
    }
    DWBC_SYNC();
    if (!Bout) return;
    // c_j = (V_body(j) x_m S_j) qd_j ; zero for the world-axis translation dofs (constant subspace)
    for (int j = th.tid; j < N; j += NT) {
        real_t c[6] = {0, 0, 0, 0, 0, 0};
        if (j >= 3) {
            const real_t *Vb = V + (j < 6 ? 0 : j - 5) * 6, *s = Sm + j * 6;
            const real_t qj = (real_t)qd[j];
            const real_t *w = Vb, *v = Vb + 3, *a = s, *bb = s + 3;
            c[0] = (w[1] * a[2] - w[2] * a[1]) * qj;
            c[1] = (w[2] * a[0] - w[0] * a[2]) * qj;
            c[2] = (w[0] * a[1] - w[1] * a[0]) * qj;
            c[3] = ((w[1] * bb[2] - w[2] * bb[1]) + (v[1] * a[2] - v[2] * a[1])) * qj;
            c[4] = ((w[2] * bb[0] - w[0] * bb[2]) + (v[2] * a[0] - v[0] * a[2])) * qj;
            c[5] = ((w[0] * bb[1] - w[1] * bb[0]) + (v[0] * a[1] - v[1] * a[0])) * qj;
        }
        for (int a = 0; a < 6; a++) Cj[j * 6 + a] = c[a];
    }
    DWBC_SYNC();
    for (int b = th.tid; b < nb; b += NT) {
        real_t a[6] = {0, 0, 0, 0, 0, kGrav};
        for (int j = 3; j < 6; j++)
            for (int x = 0; x < 6; x++) a[x] += Cj[j * 6 + x];
        for (int c = b; c > 0; c = su.parent[c])
            for (int x = 0; x < 6; x++) a[x] += Cj[(c + 5) * 6 + x];
        for (int x = 0; x < 6; x++) Ab[b * 6 + x] = a[x];
    }
    DWBC_SYNC();
    real_t *F = L + S::k_F;  // the c_j are dead
    for (int b = th.tid; b < nb; b += NT) {
        const real_t *bd = body + b * kBodyStride;
        const real_t *R = Rw + b * 9;
        const real_t m = bd[BF_MASS];
        real_t c[3], Ic[9], Tm[9];
        for (int a = 0; a < 3; a++)
            c[a] = pw[b * 3 + a] + R[a * 3] * bd[BF_COM] + R[a * 3 + 1] * bd[BF_COM + 1] + R[a * 3 + 2] * bd[BF_COM + 2] - pw[a];
        const real_t I6[9] = {bd[BF_ICOM], bd[BF_ICOM + 1], bd[BF_ICOM + 2], bd[BF_ICOM + 1], bd[BF_ICOM + 3],
                              bd[BF_ICOM + 4], bd[BF_ICOM + 2], bd[BF_ICOM + 4], bd[BF_ICOM + 5]};
        for (int a = 0; a < 3; a++)
            for (int e = 0; e < 3; e++) Tm[a * 3 + e] = R[a * 3] * I6[e] + R[a * 3 + 1] * I6[3 + e] + R[a * 3 + 2] * I6[6 + e];
        for (int a = 0; a < 3; a++)
            for (int e = 0; e < 3; e++) Ic[a * 3 + e] = Tm[a * 3] * R[e * 3] + Tm[a * 3 + 1] * R[e * 3 + 1] + Tm[a * 3 + 2] * R[e * 3 + 2];
        real_t hv[6], ha[6];
        spatial_inertia_apply(m, c, Ic, V + b * 6, hv);
        spatial_inertia_apply(m, c, Ic, Ab + b * 6, ha);
        const real_t *w = V + b * 6, *v = V + b * 6 + 3, *n = hv, *f = hv + 3;
        // V x* h = [w x n + v x f ; w x f]
        F[b * 6 + 0] = ha[0] + (w[1] * n[2] - w[2] * n[1]) + (v[1] * f[2] - v[2] * f[1]);
        F[b * 6 + 1] = ha[1] + (w[2] * n[0] - w[0] * n[2]) + (v[2] * f[0] - v[0] * f[2]);
        F[b * 6 + 2] = ha[2] + (w[0] * n[1] - w[1] * n[0]) + (v[0] * f[1] - v[1] * f[0]);
        F[b * 6 + 3] = ha[3] + (w[1] * f[2] - w[2] * f[1]);
        F[b * 6 + 4] = ha[4] + (w[2] * f[0] - w[0] * f[2]);
        F[b * 6 + 5] = ha[5] + (w[0] * f[1] - w[1] * f[0]);
    }
    DWBC_SYNC();
    for (int j = th.tid; j < N; j += NT) {
        const int b = j < 6 ? 0 : j - 5;
        const int len = topo[2 * nb + b];  // bodies are numbered depth first: subtree(b) = [b, b + len)
        real_t fs[6] = {0, 0, 0, 0, 0, 0};
        for (int d = b; d < b + len; d++)
            for (int a = 0; a < 6; a++) fs[a] += F[d * 6 + a];
        const real_t *s = Sm + j * 6;
        Bout[j] = s[0] * fs[0] + s[1] * fs[1] + s[2] * fs[2] + s[3] * fs[3] + s[4] * fs[4] + s[5] * fs[5];
    }
    DWBC_SYNC();
}

}  // namespace dwbc

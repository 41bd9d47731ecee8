// dwbc_fstar.h -- on-device task reference: TaskLink trajectory (quintic / slerp) + PD -> f*  (SURVEY.md §8 row f2).
//
// Reference functions restated (file:line in the reference tree):
//   QuinticSpline                 src/math.cpp:127-182
//   GetPhi                        src/math.cpp:275-291
//   TaskLink::GetFstarPosPD       src/task.cpp:268-293
//   TaskLink::GetFstarRotPD       src/task.cpp:295-339   (Eigen Quaternion(Matrix3), slerp, AngleAxis(Quaternion) [ext])
//   RobotData::UpdateTaskSpace    src/dwbc.cpp:708-780   (which point / velocity of the link each task mode feeds in)
// A link whose trajectory was registered with dwbc_batch_set_trajectory gets its f* segment from here, every other
// segment keeps the value of dwbc_batch_set_fstar (SetTaskSpace), exactly like traj_pos_set / traj_rot_set in the
// reference.  The per-instance record is TRAJ_STRIDE doubles:
//   t0 t1 | pos_init vel_init pos_desired vel_desired | rot_init (row-major 3x3) rot_desired | has_pos has_rot
// One lane per task link does the scalar work (a handful of transcendental calls); it runs only when a trajectory is set.
#pragma once
#include "dwbc_cycle.h"

namespace dwbc {

constexpr int kTrajStride = 34;

DWBC_DEV void quintic_spline(real_t t, real_t t0, real_t tf, real_t x0, real_t xd0, real_t xf, real_t xdf, real_t *out) {
    // zero start / end acceleration (the only way the reference calls it: task.cpp:280,314)
    if (t < t0) { out[0] = x0; out[1] = xd0; out[2] = real_t(0.0); return; }
    if (t > tf) { out[0] = xf; out[1] = xdf; out[2] = real_t(0.0); return; }
    const real_t ts = tf - t0, ts2 = ts * ts, ts3 = ts2 * ts;
    // Temp^-1 R_temp in closed form: the 3x3 system of src/math.cpp:157-170 solved symbolically
    const real_t r0 = xf - x0 - xd0 * ts, r1 = xdf - xd0, r2 = real_t(0.0);
    const real_t a4 = (real_t(10.0) * r0 - real_t(4.0) * r1 * ts + real_t(0.5) * r2 * ts2) / ts3;
    const real_t a5 = (-real_t(15.0) * r0 + real_t(7.0) * r1 * ts - r2 * ts2) / (ts3 * ts);
    const real_t a6 = (real_t(6.0) * r0 - real_t(3.0) * r1 * ts + real_t(0.5) * r2 * ts2) / (ts3 * ts2);
    const real_t s = t - t0, s2 = s * s, s3 = s2 * s, s4 = s3 * s;
    out[0] = x0 + xd0 * s + a4 * s3 + a5 * s4 + a6 * s4 * s;
    out[1] = xd0 + real_t(3.0) * a4 * s2 + real_t(4.0) * a5 * s3 + real_t(5.0) * a6 * s4;
    out[2] = real_t(6.0) * a4 * s + real_t(12.0) * a5 * s2 + real_t(20.0) * a6 * s3;
}

DWBC_DEV void quat_from_rot(const real_t *m, real_t *q) {  // Eigen Quaternion(Matrix3) [ext]; m row-major, q = (x, y, z, w)
    real_t t = m[0] + m[4] + m[8];
    if (t > real_t(0.0)) {
        t = sqrt(t + real_t(1.0));
        q[3] = real_t(0.5) * t;
        t = real_t(0.5) / t;
        q[0] = (m[7] - m[5]) * t;
        q[1] = (m[2] - m[6]) * t;
        q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 4]) i = 2;
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        t = sqrt(m[i * 4] - m[j * 4] - m[k * 4] + real_t(1.0));
        real_t qq[4];
        qq[i] = real_t(0.5) * t;
        t = real_t(0.5) / t;
        qq[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
        qq[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        qq[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
        for (int a = 0; a < 4; a++) q[a] = qq[a];
    }
}

// f* segment of one task link (src/dwbc.cpp:708-780).  R, p: link rotation / origin; w, v: its angular / origin velocity
DWBC_DEV void link_fstar(int mode, real_t t, const real_t *tr, const real_t *g, const real_t *R, const real_t *p, const real_t *w,
                         const real_t *v, const real_t *com_l, const real_t *tpoint, real_t *out) {
    const bool six = mode <= TASK_LINK_6D_CUSTOM_FRAME;
    const bool posm = mode >= TASK_LINK_POSITION && mode <= TASK_LINK_POSITION_CUSTOM_FRAME;
    const bool has_pos = tr[32] != real_t(0.0), has_rot = tr[33] != real_t(0.0);
    if ((six || posm) && has_pos) {
        real_t cp[3], cv[3];
        if (mode == TASK_LINK_6D_COM_FRAME || mode == TASK_LINK_POSITION_COM_FRAME) {
            real_t rc[3];
            for (int a = 0; a < 3; a++) rc[a] = R[a * 3] * com_l[0] + R[a * 3 + 1] * com_l[1] + R[a * 3 + 2] * com_l[2];
            for (int a = 0; a < 3; a++) cp[a] = p[a] + rc[a];  // xipos, vi (link.cpp:83,94)
            cv[0] = v[0] + (w[1] * rc[2] - w[2] * rc[1]);
            cv[1] = v[1] + (w[2] * rc[0] - w[0] * rc[2]);
            cv[2] = v[2] + (w[0] * rc[1] - w[1] * rc[0]);
        } else if (mode == TASK_LINK_6D_CUSTOM_FRAME || mode == TASK_LINK_POSITION_CUSTOM_FRAME) {
            for (int a = 0; a < 3; a++) cp[a] = p[a] + R[a * 3] * tpoint[0] + R[a * 3 + 1] * tpoint[1] + R[a * 3 + 2] * tpoint[2];
            cv[0] = v[0] + (w[1] * tpoint[2] - w[2] * tpoint[1]);  // sic: w x task_point_ with the LOCAL point (dwbc.cpp:742)
            cv[1] = v[1] + (w[2] * tpoint[0] - w[0] * tpoint[2]);
            cv[2] = v[2] + (w[0] * tpoint[1] - w[1] * tpoint[0]);
        } else {
            for (int a = 0; a < 3; a++) { cp[a] = p[a]; cv[a] = v[a]; }
        }
        for (int j = 0; j < 3; j++) {  // GetFstarPosPD (task.cpp:268-293); gains: pos_p pos_d pos_a rot_p rot_d
            real_t qn[3];
            quintic_spline(t, tr[0], tr[1], tr[2 + j], tr[5 + j], tr[8 + j], tr[11 + j], qn);
            out[j] = g[6 + j] * qn[2] + g[j] * (qn[0] - cp[j]) + g[3 + j] * (qn[1] - cv[j]);
        }
    }
    if ((six || !posm) && has_rot) {  // GetFstarRotPD (task.cpp:295-339)
        real_t qs[3], qi[4], qd[4], qt[4];
        quintic_spline(t, tr[0], tr[1], real_t(0.0), real_t(0.0), real_t(1.0), real_t(0.0), qs);
        quat_from_rot(tr + 14, qi);
        quat_from_rot(tr + 23, qd);
        {  // Eigen slerp [ext]
            const real_t d = qi[0] * qd[0] + qi[1] * qd[1] + qi[2] * qd[2] + qi[3] * qd[3], ad = fabs(d);
            real_t s0, s1;
            if (ad >= real_t(1.0) - real_t(2.220446049250313e-16)) { s0 = real_t(1.0) - qs[0]; s1 = qs[0]; }
            else {
                const real_t th = acos(ad), st = sin(th);
                s0 = sin((real_t(1.0) - qs[0]) * th) / st;
                s1 = sin(qs[0] * th) / st;
            }
            if (d < real_t(0.0)) s1 = -s1;
            for (int a = 0; a < 4; a++) qt[a] = s0 * qi[a] + s1 * qd[a];
        }
        real_t Rt[9];
        {
            const real_t x = qt[0], y = qt[1], z = qt[2], ww = qt[3];
            Rt[0] = 1 - 2 * (y * y + z * z); Rt[1] = 2 * (x * y - ww * z); Rt[2] = 2 * (x * z + ww * y);
            Rt[3] = 2 * (x * y + ww * z); Rt[4] = 1 - 2 * (x * x + z * z); Rt[5] = 2 * (y * z - ww * x);
            Rt[6] = 2 * (x * z - ww * y); Rt[7] = 2 * (y * z + ww * x); Rt[8] = 1 - 2 * (x * x + y * y);
        }
        // AngleAxis(rq_desired * rq_init.inverse()) [ext]
        real_t ang, ax[3];
        {
            const real_t n2 = qi[0] * qi[0] + qi[1] * qi[1] + qi[2] * qi[2] + qi[3] * qi[3];
            const real_t bx = -qi[0] / n2, by = -qi[1] / n2, bz = -qi[2] / n2, bw = qi[3] / n2;
            const real_t ax_ = qd[0], ay = qd[1], az = qd[2], aw = qd[3];
            const real_t e[4] = {aw * bx + ax_ * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax_ * bz,
                                 aw * bz + az * bw + ax_ * by - ay * bx, aw * bw - ax_ * bx - ay * by - az * bz};
            real_t n = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
            if (n != real_t(0.0)) {
                ang = real_t(2.0) * atan2(n, fabs(e[3]));
                if (e[3] < real_t(0.0)) n = -n;
                ax[0] = e[0] / n; ax[1] = e[1] / n; ax[2] = e[2] / n;
            } else { ang = real_t(0.0); ax[0] = real_t(1.0); ax[1] = real_t(0.0); ax[2] = real_t(0.0); }
        }
        // GetPhi(current_rot, rot_traj) = 1/2 sum_i col_i(R) x col_i(Rt)
        real_t phi[3] = {0, 0, 0};
        for (int i = 0; i < 3; i++) {
            const real_t a0 = R[i], a1 = R[3 + i], a2 = R[6 + i], b0 = Rt[i], b1 = Rt[3 + i], b2 = Rt[6 + i];
            phi[0] += a1 * b2 - a2 * b1;
            phi[1] += a2 * b0 - a0 * b2;
            phi[2] += a0 * b1 - a1 * b0;
        }
        const int o = six ? 3 : 0;
        for (int j = 0; j < 3; j++) out[o + j] = g[9 + j] * (real_t(0.5) * phi[j]) + g[12 + j] * (ang * qs[1] * ax[j] - w[j]);
    }
}

// fills L[S::fs ..) with this instance's f* (SetTaskSpace values, overridden per link by the trajectories).  Vb: the
// (nb x 6) body velocities [w; v_O] of velocity_rnea(), or nullptr for zero velocity.
template <class S, int N, int NB, int NT, bool kExtras = true>
DWBC_DEV void task_reference(Thr th, const Setup &su, const BatchIO &io, int inst, const real_t *body, real_t *L, const real_t *Vb) {
    real_t *fs = L + S::fs;
    const io_t *qd = io.qdot ? io.qdot + (size_t)inst * N : nullptr;
    const io_t *fin = io.fstar + (size_t)inst * su.fstar_total;
    for (int i = th.tid; i < su.fstar_total; i += NT) fs[i] = (real_t)fin[i];
    DWBC_SYNC();
    if (!kExtras || !io.traj || su.n_traj == 0) return;
    const real_t tnow = io.ctime ? (real_t)io.ctime[inst] : real_t(0.0);
    for (int idx = th.tid; idx < kMaxLevels * kMaxTaskLinks; idx += NT) {
        const int lv = idx / kMaxTaskLinks, li = idx - lv * kMaxTaskLinks;
        if (lv >= su.n_levels || li >= su.t_nlinks[lv]) continue;
        const int slot = su.t_traj_slot[lv][li];
        if (slot < 0) continue;
        int off = su.fstar_off[lv];
        for (int a = 0; a < li; a++) off += su.t_mode[lv][a] <= TASK_LINK_6D_CUSTOM_FRAME ? 6 : 3;
        const int link = su.t_link[lv][li];
        const bool is_com = link == su.nb;  // COM link: xpos = com_pos, rotm = pelvis rotm, (v, w) = jac_ qdot (dwbc.cpp:326,350,360-367)
        const real_t *R = L + S::Rw + (is_com ? 0 : link) * 9, *p = is_com ? L + S::Jcm + 6 * N : L + S::pw + link * 3, *O = L + S::pw;
        real_t w[3] = {0, 0, 0}, v[3] = {0, 0, 0};
        if (is_com) {
            if (qd)
                for (int j = 0; j < N; j++)
                    for (int a = 0; a < 3; a++) { v[a] += L[S::Jcm + a * N + j] * (real_t)qd[j]; w[a] += L[S::Jcm + (3 + a) * N + j] * (real_t)qd[j]; }
        } else if (Vb) {
            const real_t *V = Vb + link * 6;
            const real_t d0 = p[0] - O[0], d1 = p[1] - O[1], d2 = p[2] - O[2];
            w[0] = V[0]; w[1] = V[1]; w[2] = V[2];
            v[0] = V[3] + (w[1] * d2 - w[2] * d1);
            v[1] = V[4] + (w[2] * d0 - w[0] * d2);
            v[2] = V[5] + (w[0] * d1 - w[1] * d0);
        }
        real_t gl[15], tp[3];  // Setup keeps host doubles
        for (int a = 0; a < 15; a++) gl[a] = (real_t)su.t_gain[lv][li][a];
        for (int a = 0; a < 3; a++) tp[a] = (real_t)su.t_point[lv][li][a];
        real_t trl[kTrajStride];
        for (int a = 0; a < kTrajStride; a++) trl[a] = (real_t)io.traj[((size_t)inst * su.n_traj + slot) * kTrajStride + a];
        link_fstar(su.t_mode[lv][li], tnow, trl, gl, R, p, w, v,
                   body + (is_com ? 0 : link) * kBodyStride + BF_COM, tp, fs + off);
    }
    DWBC_SYNC();
}

}  // namespace dwbc

// emu_cycle.cpp -- TEST HARNESS ONLY.  Compiles the kernel source libdwbc_amd/csrc/dwbc_cycle.h for the host with
// one "thread" per workgroup (NT = 1, barriers are no-ops) so that the arithmetic and indexing of the fused
// cycle can be checked against the oracle on a machine without a GPU (pytest -m "not gpu").  It shares the
// product's URDF reader and Setup builder, which this also covers.  It is never linked into libdwbc_hip.so and
// is not a fallback: the product has no CPU path.
#define DWBC_HOST_EMU 1
#include <cstdlib>
#include <cstring>
#include <limits>
#include <algorithm>
#include <string>
#include <vector>

#include "../../libdwbc_amd/csrc/dwbc_reduced.h"
#include "../../libdwbc_amd/csrc/dwbc_cycle2p.h"
#include "../../libdwbc_amd/csrc/dwbc_cycle_gc.h"
#include "../../libdwbc_amd/csrc/dwbc_hqp.h"
#include "../../libdwbc_amd/csrc/dwbc_model.h"
#include "../../libdwbc_amd/csrc/dwbc_setup.h"

using namespace dwbc;

// the boundary buffers are double (io_t) in both builds; the model table and the dump record are real_t (float in libdwbc_emu_f32.so)
template <class T>
static std::vector<real_t> to_real(const T *p, size_t n) {
    std::vector<real_t> v(p ? n : 0);
    for (size_t i = 0; i < v.size(); i++) v[i] = (real_t)p[i];
    return v;
}
static void from_real(const std::vector<real_t> &v, double *out) {
    if (!out) return;
    for (size_t i = 0; i < v.size(); i++) out[i] = (double)v[i];
}

struct EmuCtx {
    Model model;
    Setup su;
    std::vector<double> body;
    std::vector<int> topo;
    std::string err;
};

extern "C" {

EmuCtx *emu_create(const char *urdf) {
    auto *c = new EmuCtx();
    if (!load_urdf(urdf, true, c->model, c->err)) return c;
    setup_init(c->su, c->model.nb, c->model.ndof, c->model.maxdepth);
    c->model.body_table(c->body);
    c->model.topo_table(c->topo);
    setup_set_parents(c->su, c->topo.data());
    return c;
}
// a model given as arrays (e.g. the result of model surgery): same tables as emu_create builds from a URDF
EmuCtx *emu_create_from_arrays(int nb, const int *parent, const double *R_T, const double *p_T, const double *axis, const double *mass, const double *com, const double *inertia) {
    auto *c = new EmuCtx();
    Model &m = c->model;
    m.parent.assign(parent, parent + nb);
    m.R_T.assign(R_T, R_T + 9 * nb);
    m.p_T.assign(p_T, p_T + 3 * nb);
    m.axis.assign(axis, axis + 3 * nb);
    m.mass.assign(mass, mass + nb);
    m.com.assign(com, com + 3 * nb);
    m.inertia.assign(inertia, inertia + 9 * nb);
    for (int i = 0; i < nb; i++) m.names.push_back("link" + std::to_string(i));
    m.finalize();
    setup_init(c->su, m.nb, m.ndof, m.maxdepth);
    m.body_table(c->body);
    m.topo_table(c->topo);
    setup_set_parents(c->su, c->topo.data());
    return c;
}
const char *emu_error(EmuCtx *c) { return c->err.c_str(); }
void emu_destroy(EmuCtx *c) { delete c; }
int emu_nb(EmuCtx *c) { return c->model.nb; }
int emu_ndof(EmuCtx *c) { return c->model.ndof; }
int emu_link_id(EmuCtx *c, const char *name) { return c->model.link_id(name); }
void emu_get_model(EmuCtx *c, int *parent, double *R_T, double *p_T, double *axis, double *mass, double *com, double *inertia) {
    const Model &m = c->model;
    for (int i = 0; i < m.nb; i++) parent[i] = m.parent[i];
    memcpy(R_T, m.R_T.data(), 8 * 9 * m.nb);
    memcpy(p_T, m.p_T.data(), 8 * 3 * m.nb);
    memcpy(axis, m.axis.data(), 8 * 3 * m.nb);
    memcpy(mass, m.mass.data(), 8 * m.nb);
    memcpy(com, m.com.data(), 8 * 3 * m.nb);
    memcpy(inertia, m.inertia.data(), 8 * 9 * m.nb);
}
int emu_add_contact(EmuCtx *c, int link, const double *pt, double lx, double ly, double mu, double muz) {
    return setup_add_contact(c->su, link, 0, pt, lx, ly, mu, muz, c->err);
}
int emu_add_task(EmuCtx *c, int level, int mode, int link, const double *pt) { return setup_add_task(c->su, level, mode, link, pt, c->err) ? 1 : 0; }
void emu_set_tau_lim(EmuCtx *c, const double *lim) {
    c->su.has_tau_lim = lim != nullptr;
    if (lim) for (int i = 0; i < c->model.ndof - 6; i++) c->su.tau_lim[i] = lim[i];
}
int emu_fstar_total(EmuCtx *c) { return c->su.fstar_total; }
int emu_dump_total(EmuCtx *c) { return DumpLayout::make(c->model.ndof).total; }
int emu_dump_offset(EmuCtx *c, const char *name) {
    DumpLayout d = DumpLayout::make(c->model.ndof);
    std::string n(name);
#define F(x) if (n == #x) return d.x;
    F(A) F(A_inv) F(J_C) F(Lambda_c) F(J_C_INV_T) F(A_inv_N_C) F(W_inv) F(NwJw) F(Vb) F(G) F(P_C) F(link_R) F(link_p)
    F(J_task) F(Lambda_task) F(J_kt) F(X) F(Y) F(fstar_qp) F(contact_qp) F(cf_redis) F(qp_viol) F(CMM) F(com) F(com_inertia) F(J_com) F(B) F(link_v) F(link_w) F(contact_pos) F(contact_rot) F(zmp) F(A_R_inv) F(A_R) F(G_R) F(J_I_nc) F(J_I_nc_inv_T)
#undef F
    return -1;
}
int emu_diag_count() { return DG_COUNT; }
int emu_lds_bytes() { return Lds2<39, 34, 2>::total_bytes; }

int emu_lds_bytes_reduced(int nlv) { return nlv == 1 ? LdsR<39, 34, 1>::total_bytes : nlv == 2 ? LdsR<39, 34, 2>::total_bytes : nlv == 3 ? LdsR<39, 34, 3>::total_bytes : LdsR<39, 34, 4>::total_bytes; }
int emu_lds_bytes_v2(int nlv) { return nlv == 1 ? Lds2<39, 34, 1>::total_bytes : nlv == 2 ? Lds2<39, 34, 2>::total_bytes : nlv == 3 ? Lds2<39, 34, 3>::total_bytes : Lds2<39, 34, 4>::total_bytes; }

int emu_run_reduced(EmuCtx *c, int B, const double *q, const unsigned char *flags, const double *fstar, double *tau,
                    double *wrench, int *status, int *diag, double *dump) {
    if (c->model.ndof != 39 || c->model.nb != 34) { c->err = "emu is instantiated for TOCABI (39 dof) only"; return 0; }
    const int n = 39, m = 33;
    const size_t D = DumpLayout::make(n).total;
    (void)m;
    auto rb = to_real(c->body.data(), c->body.size());
    std::vector<real_t> rdump(dump ? (size_t)B * D : 0);
    BatchIO io{};
    io.B = B; io.q = q; io.flags = flags; io.fstar = fstar; io.tau = tau; io.wrench = wrench; io.status = status;
    io.diag = diag; io.dump = dump ? rdump.data() : nullptr; io.body = rb.data(); io.topo = c->topo.data(); io.hqp = 1;
    std::vector<real_t> lds(LdsR<39, 34, 4>::rtotal + 64);
    std::vector<int> ilds(64);
    for (int b = 0; b < B; b++) {
        Thr th{0};
        if (c->su.n_levels == 1) cycle_instance_reduced<39, 34, 1, 1, TopoTocabi>(th, c->su, io, b, lds.data(), ilds.data());
        else if (c->su.n_levels == 2) cycle_instance_reduced<39, 34, 2, 1, TopoTocabi>(th, c->su, io, b, lds.data(), ilds.data());
        else if (c->su.n_levels == 3) cycle_instance_reduced<39, 34, 3, 1, TopoTocabi>(th, c->su, io, b, lds.data(), ilds.data());
        else cycle_instance_reduced<39, 34, 4, 1, TopoTocabi>(th, c->su, io, b, lds.data(), ilds.data());
    }
    from_real(rdump, dump);
    return 1;
}

static int g_emu_hqp = 1;
static int g_emu_dense = 0;
void emu_set_hqp(int hqp) { g_emu_hqp = hqp; }
// 1: two-level full-dynamics runs use the TopoGeneric (dense A^-1 sweep) instantiation instead of TOCABI's constant tree
void emu_set_dense(int d) { g_emu_dense = d; }
static const double *g_emu_custom = nullptr;
int emu_add_custom_task(EmuCtx *c, int level, int dof) { return setup_add_custom_task(c->su, level, dof, c->err) ? 1 : 0; }
void emu_set_custom(const double *J) { g_emu_custom = J; }
static const double *g_emu_traj = nullptr, *g_emu_ctime = nullptr;
void emu_set_traj(EmuCtx *c, int level, int link_index, int slot, const double *gains15) {
    c->su.t_traj_slot[level][link_index] = slot;
    if (slot + 1 > c->su.n_traj) c->su.n_traj = slot + 1;
    for (int a = 0; a < 15; a++) c->su.t_gain[level][link_index][a] = gains15[a];
}
void emu_set_traj_data(const double *traj, const double *ctime) { g_emu_traj = traj; g_emu_ctime = ctime; }
static int g_emu_warm = 0;
void emu_set_warm(int on) { g_emu_warm = on; }
// 1: the lean build on the compact LDS map (Lds3: the throughput kernel of batches beyond four instances per CU); LDS is
// poisoned with NaN before every instance, so a read of a block that nothing has written yet shows up in the result
static int g_emu_compact = 0;
void emu_set_compact(int on) { g_emu_compact = on; }
int emu_lds_bytes_pair(int nlv) { return nlv == 1 ? Lds4<39, 34, 1>::total_bytes : Lds4<39, 34, 2>::total_bytes; }
int emu_lds_bytes_compact(int nlv) { return nlv == 1 ? Lds3<39, 34, 1>::total_bytes : nlv == 2 ? Lds3<39, 34, 2>::total_bytes : nlv == 3 ? Lds3<39, 34, 3>::total_bytes : Lds3<39, 34, 4>::total_bytes; }
static const double *g_emu_qdot = nullptr;
void emu_set_qdot(const double *qd) { g_emu_qdot = qd; }

int emu_run(EmuCtx *c, int B, const double *q, const unsigned char *flags, const double *fstar, double *tau, double *wrench,
            int *status, int *diag, double *dump) {
    if (c->model.ndof != 39 || c->model.nb != 34) { c->err = "emu is instantiated for TOCABI (39 dof) only"; return 0; }
    const int n = 39, m = 33;
    const size_t D = DumpLayout::make(n).total;
    (void)m;
    auto rb = to_real(c->body.data(), c->body.size());
    std::vector<real_t> rdump(dump ? (size_t)B * D : 0);
    BatchIO io{};
    io.B = B;
    io.q = q;
    io.qdot = g_emu_qdot;
    io.traj = c->su.n_traj > 0 ? g_emu_traj : nullptr;
    io.ctime = g_emu_ctime;
    io.custom_J = c->su.n_custom > 0 ? g_emu_custom : nullptr;
    io.flags = flags;
    io.fstar = fstar;
    io.tau = tau;
    io.wrench = wrench;
    io.status = status;
    io.diag = diag;
    io.dump = dump ? rdump.data() : nullptr;
    io.body = rb.data();
    io.topo = c->topo.data();
    io.hqp = g_emu_hqp;
    io.warm = g_emu_warm;
    std::vector<real_t> lds(Lds2<39, 34, 4>::total + 64);
    std::vector<int> ilds(64);
    if (g_emu_compact == 2) {
        // the paired kernel of dwbc_cycle2p.h with both roles run one after the other in every phase (wave = -1), LDS poisoned
        if (dump) { c->err = "the paired (lean) build has no dump record"; return 0; }
        if (c->su.n_levels > 2) { c->err = "the paired kernel is built for one and two task levels"; return 0; }
        std::vector<real_t> l4(Lds4<39, 34, 2>::total + 64);
        for (int b = 0; b < B; b++) {
            Thr th{0};
            std::fill(l4.begin(), l4.end(), std::numeric_limits<real_t>::quiet_NaN());
            if (c->su.n_levels == 1) cycle_instance_v2p<39, 34, 1, 1, TopoTocabi>(-1, th, c->su, io, b, l4.data());
            else cycle_instance_v2p<39, 34, 2, 1, TopoTocabi>(-1, th, c->su, io, b, l4.data());
        }
        return 1;
    }
    if (g_emu_compact) {
        if (dump) { c->err = "the compact (lean) build has no dump record"; return 0; }
        io.diag = diag;
        for (int b = 0; b < B; b++) {
            Thr th{0};
            const real_t nan_ = std::numeric_limits<real_t>::quiet_NaN();
            auto run = [&](auto nlv) {
                constexpr int NLV = decltype(nlv)::value;
                std::fill(lds.begin(), lds.end(), nan_);
                cycle_instance_v2<39, 34, NLV, 1, false, TopoTocabi, true>(th, c->su, io, b, lds.data(), ilds.data());
            };
            if (c->su.n_levels == 1) run(std::integral_constant<int, 1>{});
            else if (c->su.n_levels == 2) run(std::integral_constant<int, 2>{});
            else if (c->su.n_levels == 3) run(std::integral_constant<int, 3>{});
            else run(std::integral_constant<int, 4>{});
        }
        return 1;
    }
    for (int b = 0; b < B; b++) {
        Thr th{0};
        if (c->su.n_levels == 1) cycle_instance_v2<39, 34, 1, 1, true, TopoTocabi>(th, c->su, io, b, lds.data(), ilds.data());
        else if (c->su.n_levels == 2 && g_emu_dense) cycle_instance_v2<39, 34, 2, 1, true, TopoGeneric>(th, c->su, io, b, lds.data(), ilds.data());
        else if (c->su.n_levels == 2) cycle_instance_v2<39, 34, 2, 1, true, TopoTocabi>(th, c->su, io, b, lds.data(), ilds.data());
        else if (c->su.n_levels == 3) cycle_instance_v2<39, 34, 3, 1, true, TopoTocabi>(th, c->su, io, b, lds.data(), ilds.data());
        else cycle_instance_v2<39, 34, 4, 1, true, TopoTocabi>(th, c->su, io, b, lds.data(), ilds.data());
    }
    from_real(rdump, dump);
    return 1;
}

// the general-contact kernel of dwbc_cycle_gc.h (up to three active contacts; wrench: B x 18), LDS poisoned per instance
int emu_run_gc(EmuCtx *c, int B, const double *q, const unsigned char *flags, const double *fstar, double *tau, double *wrench,
               int *status, int *diag) {
    const int n_ = c->model.ndof, nb_ = c->model.nb;
    if (!((n_ == 39 && nb_ == 34) || (n_ == 37 && nb_ == 32) || (n_ == 23 && nb_ == 18))) { c->err = "emu_run_gc: instantiated for (39, 34), (37, 32) and (23, 18)"; return 0; }
    auto rb = to_real(c->body.data(), c->body.size());
    BatchIO io{};
    io.B = B;
    io.q = q;
    io.flags = flags;
    io.fstar = fstar;
    io.tau = tau;
    io.wrench = wrench;
    io.wrench_ld = 18;
    io.status = status;
    io.diag = diag;
    io.body = rb.data();
    io.topo = c->topo.data();
    io.hqp = 1;
    const bool wide_tasks = setup_wide_tasks(c->su);  // a level of more than 6 dof: the TG = 12 instantiation (TOCABI's size)
    if (wide_tasks && n_ != 39) { c->err = "emu_run_gc: task levels of more than 6 dof are instantiated for (39, 34)"; return 0; }
    std::vector<real_t> lds(LdsG<39, 34, 3, kMaxTaskDofWide>::total + 64);
    static_assert(LdsG<37, 32, 3>::total <= LdsG<39, 34, 3>::total && LdsG<23, 18, 3>::total <= LdsG<39, 34, 3>::total &&
                  LdsG<39, 34, 3>::total <= LdsG<39, 34, 3, kMaxTaskDofWide>::total, "one buffer");
    for (int b = 0; b < B; b++) {
        std::fill(lds.begin(), lds.end(), std::numeric_limits<real_t>::quiet_NaN());
        if (wide_tasks) cycle_instance_gc<39, 34, 3, 1, kMaxTaskDofWide>(Thr{0}, c->su, io, b, lds.data());
        else if (n_ == 39) cycle_instance_gc<39, 34, 3, 1>(Thr{0}, c->su, io, b, lds.data());
        else if (n_ == 37) cycle_instance_gc<37, 32, 3, 1>(Thr{0}, c->su, io, b, lds.data());
        else cycle_instance_gc<23, 18, 3, 1>(Thr{0}, c->su, io, b, lds.data());
    }
    return 1;
}

// ---- other model sizes (the kernel packs of dwbc_pack.hip): the same source instantiated for (37, 32) and (23, 18), two task
//      levels, TopoGeneric -- the sizes tests/test_model_packs.py builds from the TOCABI fixture (43 / 38: four links added to a hand)
}  // extern "C"
template <int N, int NB>
static void emu_run_size(EmuCtx *c, BatchIO &io, int B) {
    std::vector<real_t> lds(Lds2<N, NB, 2>::total + 64);
    std::vector<int> ilds(64);
    for (int b = 0; b < B; b++) cycle_instance_v2<N, NB, 2, 1, true, TopoGeneric>(Thr{0}, c->su, io, b, lds.data(), ilds.data());
}
extern "C" {
int emu_run_other(EmuCtx *c, int B, const double *q, const unsigned char *flags, const double *fstar, double *tau, double *wrench,
                  int *status, int *diag, double *dump) {
    const int n = c->model.ndof, nb = c->model.nb;
    if (c->su.n_levels != 2) { c->err = "emu_run_other: two task levels"; return 0; }
    const size_t D = DumpLayout::make(n).total;
    auto rb = to_real(c->body.data(), c->body.size());
    std::vector<real_t> rdump(dump ? (size_t)B * D : 0);
    BatchIO io{};
    io.B = B; io.q = q; io.flags = flags; io.fstar = fstar; io.tau = tau; io.wrench = wrench; io.status = status;
    io.diag = diag; io.dump = dump ? rdump.data() : nullptr; io.body = rb.data(); io.topo = c->topo.data(); io.hqp = 1;
    if (n == 37 && nb == 32) emu_run_size<37, 32>(c, io, B);
    else if (n == 23 && nb == 18) emu_run_size<23, 18>(c, io, B);
    else if (n == 43 && nb == 38) emu_run_size<43, 38>(c, io, B);
    else { c->err = "emu_run_other: instantiated for (37, 32), (23, 18) and (43, 38)"; return 0; }
    from_real(rdump, dump);
    return 1;
}

// ---- generic HQP class (dwbc_hqp.h): levels described by (m, e, has_cost); per-instance records laid out by hqp_layout()
struct EmuHqp {
    HqpDesc d;
    std::vector<double> rec, scratch;
    std::vector<int> stat;
    int B;
};
void emu_hqp_set_exact(EmuHqp *h, int level, int on);
EmuHqp *emu_hqp_create(int B, int nv, int n_levels, const int *m, const int *e, const int *has_cost, int share_cost, int solve_first) {
    auto *h = new EmuHqp();
    h->d = HqpDesc{};
    h->d.nv = nv; h->d.n_levels = n_levels; h->d.solve_first = solve_first; h->d.max_iter = 400; h->d.eps = 1.0e-6; h->d.tol = 1.0e-6;
    for (int i = 0; i < n_levels; i++) { h->d.m[i] = m[i]; h->d.e[i] = e[i]; h->d.has_cost[i] = has_cost[i]; }
    hqp_layout(h->d, share_cost != 0);
    h->B = B;
    h->rec.assign((size_t)B * h->d.rec, 0.0);
    h->scratch.assign((size_t)B * h->d.scratch, 0.0);
    h->stat.assign((size_t)B * HQS_COUNT, 0);
    return h;
}
void emu_hqp_destroy(EmuHqp *h) { delete h; }
int emu_hqp_rec(EmuHqp *h) { return h->d.rec; }
int emu_hqp_offset(EmuHqp *h, int level, int what) {  // 0 A 1 a 2 B 3 b 4 H 5 y 6 v 7 w
    const HqpDesc &d = h->d;
    const int o[8] = {d.oA[level], d.oa[level], d.oB[level], d.ob[level], d.oH[level], d.oy[level], d.ov[level], d.ow[level]};
    return o[what];
}
double *emu_hqp_data(EmuHqp *h) { return h->rec.data(); }
int *emu_hqp_stat(EmuHqp *h) { return h->stat.data(); }
int emu_hqp_lds_bytes(EmuHqp *h) { return h->d.lds * 8; }
void emu_hqp_solve(EmuHqp *h) {
    HqpIO io{h->B, h->rec.data(), h->scratch.data(), h->stat.data()};
    std::vector<double> lds(h->d.lds + 16);
    for (int b = 0; b < h->B; b++) hqp_instance<1>(Thr{0}, h->d, io, b, lds.data());
}
void emu_hqp_set_exact(EmuHqp *h, int level, int on) { h->d.exact[level] = on; }
static LqpCfg emu_cfg(EmuCtx *c, int nc, const int *act) {
    LqpCfg cfg{};
    cfg.n = c->model.ndof; cfg.nc = nc; cfg.cd = 6 * nc; cfg.n_tasks = c->su.n_levels;
    const DumpLayout dl = DumpLayout::make(cfg.n);
    for (int i = 0; i < c->su.n_levels; i++) { cfg.t_dof[i] = c->su.t_dof[i]; cfg.fstar_off[i] = c->su.fstar_off[i]; }
    cfg.fstar_total = c->su.fstar_total;
    for (int a = 0; a < nc; a++) { cfg.act[a] = act[a]; cfg.lx[a] = c->su.c_lx[act[a]]; cfg.ly[a] = c->su.c_ly[act[a]]; cfg.mu[a] = c->su.c_mu[act[a]]; cfg.muz[a] = c->su.c_muz[act[a]]; }
    cfg.oBn = dl.G;
    cfg.tlim = 200.0; cfg.alim = 5.0;
    cfg.oNorm = -1; cfg.tlim_idx = -1; cfg.jacc_mt = cfg.n - 6;
    return cfg;
}
// the reduced system (record written by reduced_record_instance): contact-chain levels src[0..n_src) packed in order
static LqpCfg emu_cfg_r(EmuCtx *c, int nc, const int *act, int RS, int n_src, const int *src) {
    LqpCfg cfg = emu_cfg(c, nc, act);
    const DumpLayout dr = DumpLayout::make(RS);
    cfg.n = RS; cfg.n_tasks = n_src;
    for (int i = 0; i < n_src; i++) { cfg.t_dof[i] = c->su.t_dof[src[i]]; cfg.fstar_off[i] = c->su.fstar_off[src[i]]; }
    cfg.oBn = dr.G; cfg.oNorm = dr.com; cfg.tlim_idx = RS - 6 - 4; cfg.tlim_special = 600.0; cfg.jacc_mt = RS - 12;
    return cfg;
}
int emu_rrec_total(int RS) { return DumpLayout::make(RS).total; }
int emu_rrec_offset(int RS, const char *name) {
    const DumpLayout d = DumpLayout::make(RS);
    const std::string s(name);
    if (s == "A") return d.A; if (s == "A_inv") return d.A_inv; if (s == "J_C") return d.J_C; if (s == "G") return d.G;
    if (s == "J_task") return d.J_task; if (s == "com") return d.com;
    return -1;
}
void emu_reduced_record(EmuCtx *c, int B, int vcd, int cd, int n_src, const int *src, const double *dump, double *rrec) {
    ReducedRecCfg rc{};
    rc.n = c->model.ndof; rc.vcd = vcd; rc.cd = cd; rc.n_src = n_src;
    for (int i = 0; i < n_src; i++) { rc.src[i] = src[i]; rc.t_dof[i] = c->su.t_dof[src[i]]; }
    for (int b = 0; b < B; b++) reduced_record_instance<1>(Thr{0}, rc, dump, rrec, b);
}
int emu_jacc_rec(EmuCtx *c) { return jacc_rec_size(c->model.ndof); }
// CalcSingleTaskTorqueWithJACC_QP for one level: h must have been created with the two levels (152 | e0, 0 | t) and exact level 0;
// prev: level x (B x jacc_rec) results of the earlier levels; out: B x jacc_rec
void emu_jacc_solve(EmuHqp *h, EmuCtx *c, int nc, const int *act, int level, const double *dump, const double *fstar, const double *const *prev, double *out, int *status) {
    const LqpCfg cfg = emu_cfg(c, nc, act);
    JaccPrev pv{};
    for (int i = 0; i < level; i++) pv.rec[i] = prev[i];
    HqpIO io{h->B, h->rec.data(), h->scratch.data(), h->stat.data()};
    std::vector<double> lds(h->d.lds + 16);
    for (int b = 0; b < h->B; b++) {
        jacc_configure_instance<1>(Thr{0}, cfg, level, pv, h->d, io, dump, fstar, b);
        hqp_instance<1>(Thr{0}, h->d, io, b, lds.data());
        jacc_extract_instance<1>(Thr{0}, cfg, level, h->d, io, dump, fstar, out, status, b);
    }
}
// RobotData::ConfigureLQP from a dump record (B x DumpLayout::total doubles) + f*, then the cascade and the LQP torque
void emu_lqp_configure(EmuHqp *h, EmuCtx *c, int nc, const int *act, int use_B, const double *dump, const double *fstar) {
    LqpCfg cfg = emu_cfg(c, nc, act);
    const DumpLayout dl = DumpLayout::make(cfg.n);
    cfg.oBn = use_B ? dl.B : dl.G;
    HqpIO io{h->B, h->rec.data(), h->scratch.data(), h->stat.data()};
    for (int b = 0; b < h->B; b++) lqp_configure_instance<1>(Thr{0}, cfg, h->d, io, dump, fstar, b);
}
void emu_lqp_torque(EmuHqp *h, EmuCtx *c, int nc, int use_B, const double *dump, double *tau) {
    LqpCfg cfg{};
    cfg.n = c->model.ndof; cfg.nc = nc; cfg.cd = 6 * nc;
    const DumpLayout dl = DumpLayout::make(cfg.n);
    cfg.oBn = use_B ? dl.B : dl.G;
    HqpIO io{h->B, h->rec.data(), h->scratch.data(), h->stat.data()};
    for (int b = 0; b < h->B; b++) lqp_torque_instance<1>(Thr{0}, cfg, h->d, io, dump, tau, b);
}
// ---- the reduced variants: the same device functions on the record of emu_reduced_record
void emu_lqp_configure_r(EmuHqp *h, EmuCtx *c, int nc, const int *act, int RS, int n_src, const int *src, const double *rrec, const double *fstar) {
    const LqpCfg cfg = emu_cfg_r(c, nc, act, RS, n_src, src);
    HqpIO io{h->B, h->rec.data(), h->scratch.data(), h->stat.data()};
    for (int b = 0; b < h->B; b++) lqp_configure_instance<1>(Thr{0}, cfg, h->d, io, rrec, fstar, b);
}
void emu_lqp_torque_r(EmuHqp *h, EmuCtx *c, int nc, const int *act, int RS, const double *rrec, double *tau) {
    const LqpCfg cfg = emu_cfg_r(c, nc, act, RS, 0, nullptr);
    HqpIO io{h->B, h->rec.data(), h->scratch.data(), h->stat.data()};
    for (int b = 0; b < h->B; b++) lqp_torque_instance<1>(Thr{0}, cfg, h->d, io, rrec, tau, b);
}
int emu_jacc_rec_r(int RS) { return jacc_rec_size(RS); }
void emu_jacc_solve_r(EmuHqp *h, EmuCtx *c, int nc, const int *act, int RS, int n_src, const int *src, int level, const double *rrec, const double *fstar,
                      const double *const *prev, double *out, int *status) {
    const LqpCfg cfg = emu_cfg_r(c, nc, act, RS, n_src, src);
    JaccPrev pv{};
    for (int i = 0; i < level; i++) pv.rec[i] = prev[i];
    HqpIO io{h->B, h->rec.data(), h->scratch.data(), h->stat.data()};
    std::vector<double> lds(h->d.lds + 16);
    for (int b = 0; b < h->B; b++) {
        jacc_configure_instance<1>(Thr{0}, cfg, level, pv, h->d, io, rrec, fstar, b);
        hqp_instance<1>(Thr{0}, h->d, io, b, lds.data());
        jacc_extract_instance<1>(Thr{0}, cfg, level, h->d, io, rrec, fstar, out, status, b);
    }
}
static NcCfg emu_nc(EmuCtx *c, int vcd, int level, int prev_stride, int prev_off) {
    NcCfg n{};
    n.n = c->model.ndof; n.vcd = vcd; n.level = level; n.link = c->su.t_link[level][0]; n.t = c->su.t_dof[level];
    n.fstar_off = c->su.fstar_off[level]; n.fstar_total = c->su.fstar_total; n.prev_stride = prev_stride; n.prev_off = prev_off;
    return n;
}
void emu_lqp_nc_configure(EmuHqp *h, EmuCtx *c, int vcd, int level, const double *dump, const double *fstar, const double *prev, int prev_stride, int prev_off) {
    const NcCfg n = emu_nc(c, vcd, level, prev_stride, prev_off);
    HqpIO io{h->B, h->rec.data(), h->scratch.data(), h->stat.data()};
    for (int b = 0; b < h->B; b++) lqp_nc_configure_instance<1>(Thr{0}, n, h->d, io, dump, fstar, prev, b);
}
void emu_hqp_solve_levels(EmuHqp *h, int n_levels, int solve_first) {  // HQP::solvefirst = (1, 1); solveSequential = (all, 0)
    HqpDesc d = h->d;
    d.n_levels = n_levels; d.solve_first = solve_first;
    HqpIO io{h->B, h->rec.data(), h->scratch.data(), h->stat.data()};
    std::vector<double> lds(h->d.lds + 16);
    for (int b = 0; b < h->B; b++) hqp_instance<1>(Thr{0}, d, io, b, lds.data());
}
int emu_jacc_nc_rec(int ncd) { return jacc_nc_rec_size(ncd); }
void emu_jacc_nc_solve(EmuHqp *h, EmuCtx *c, int vcd, int level, const double *dump, const double *fstar, const double *prev, int prev_stride, double *out, int *status) {
    const NcCfg n = emu_nc(c, vcd, level, prev_stride, 0);
    HqpDesc d = h->d;
    d.solve_first = 1;
    HqpIO io{h->B, h->rec.data(), h->scratch.data(), h->stat.data()};
    std::vector<double> lds(h->d.lds + 16);
    for (int b = 0; b < h->B; b++) {
        jacc_nc_configure_instance<1>(Thr{0}, n, d, io, dump, fstar, prev, b);
        hqp_instance<1>(Thr{0}, d, io, b, lds.data());
        jacc_nc_extract_instance<1>(Thr{0}, n, d, io, dump, fstar, prev, out, status, b);
    }
}
}

// dwbc_capi.hip -- C-ABI (include/dwbc_batch.h) + kernel launch for the MI355X-native batched libdwbc hot path.
// gfx950 only.  No CPU fallback: every compute entry point needs a HIP device and fails loudly without one.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <mutex>
#include <vector>

#include "dwbc_kernels.h"
#include "dwbc_capi_internal.h"
#include "dwbc_setup.h"

using namespace dwbc;

extern "C" int dwbc_f32_lookup(int n, int nb, int nlv, int which, int lean, int topo, const void **fn, const void **fn_wide, int *lds_bytes, int *lds_bytes_wide, int *topo_out);

// DWBC_F32 batches: the fp32 kernels (dwbc_kernels_f32.hip) work on the double buffers of the boundary; the model table is
// converted to float once
__global__ void dwbc_cvt_d2f(const double *__restrict__ in, float *__restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

// part of [torque_grav_ | torque_task_ | torque_contact_] (sel 0..2) or their sum (sel 3), B x m
__global__ void dwbc_tau_select(const double *__restrict__ tau, double *__restrict__ out, int m, size_t cnt, int sel) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    const size_t inst = i / m, j = i - inst * m;
    const double *s = tau + inst * 3 * m;
    out[i] = sel == 3 ? s[j] + s[m + j] + s[2 * m + j] : s[sel * m + j];
}

namespace dwbc {
std::string &capi_err() {
    static thread_local std::string e;
    return e;
}
}  // namespace dwbc
namespace {
int fail(const std::string &s) { return dwbc::capi_fail(s); }
}  // namespace
#define g_err (dwbc::capi_err())

extern "C" {

const char *dwbc_last_error(void) { return g_err.c_str(); }

int dwbc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

dwbc_model *dwbc_model_create_from_urdf(const char *path, int floating_base) {
    auto *mm = new dwbc_model();
    std::string err;
    if (!load_urdf(path, floating_base != 0, mm->m, err)) {
        g_err = err;
        delete mm;
        return nullptr;
    }
    return mm;
}

dwbc_model *dwbc_model_create_from_arrays(int nb, const int32_t *parent, const double *R_T, const double *p_T, const double *axis,
                                          const double *mass, const double *com, const double *inertia) {
    if (nb < 1 || nb > kMaxBodies) { g_err = "bad body count"; return nullptr; }
    auto *mm = new dwbc_model();
    Model &m = mm->m;
    m.parent.assign(parent, parent + nb);
    m.R_T.assign(R_T, R_T + 9 * nb);
    m.p_T.assign(p_T, p_T + 3 * nb);
    m.axis.assign(axis, axis + 3 * nb);
    m.mass.assign(mass, mass + nb);
    m.com.assign(com, com + 3 * nb);
    m.inertia.assign(inertia, inertia + 9 * nb);
    for (int i = 0; i < nb; i++) {
        m.names.push_back("link" + std::to_string(i));
        if (i > 0 && (parent[i] < 0 || parent[i] >= i)) { g_err = "parent[] must be a DFS pre-order"; delete mm; return nullptr; }
    }
    m.finalize();
    return mm;
}

void dwbc_model_destroy(dwbc_model *m) { delete m; }
int dwbc_model_num_links(const dwbc_model *m) { return m->m.nb; }
int dwbc_model_system_dof(const dwbc_model *m) { return m->m.ndof; }
double dwbc_model_total_mass(const dwbc_model *m) { return m->m.total_mass; }
int dwbc_model_link_id(const dwbc_model *m, const char *name) { return m->m.link_id(name); }
const char *dwbc_model_link_name(const dwbc_model *m, int link) {
    if (link == m->m.nb) return "COM";
    return (link >= 0 && link < m->m.nb) ? m->m.names[link].c_str() : "";
}
int dwbc_model_get_arrays(const dwbc_model *mm, int32_t *parent, double *R_T, double *p_T, double *axis, double *mass, double *com,
                          double *inertia) {
    const Model &m = mm->m;
    for (int i = 0; i < m.nb; i++) parent[i] = m.parent[i];
    memcpy(R_T, m.R_T.data(), sizeof(double) * 9 * m.nb);
    memcpy(p_T, m.p_T.data(), sizeof(double) * 3 * m.nb);
    memcpy(axis, m.axis.data(), sizeof(double) * 3 * m.nb);
    memcpy(mass, m.mass.data(), sizeof(double) * m.nb);
    memcpy(com, m.com.data(), sizeof(double) * 3 * m.nb);
    memcpy(inertia, m.inertia.data(), sizeof(double) * 9 * m.nb);
    return 1;
}

// ---- kernel packs: the cycle kernels of model sizes other than TOCABI's (dwbc_pack.hip), loaded on demand.  A pack is either
//      generic (any tree of its size) or built for one parent table (TopoPack: the tree-sparse sweep); the latter is preferred
//      when its table equals the model's
namespace {
struct KernelPack { void *dl; const KernelEntry *tab; int count; std::vector<int> parents; GcEntry gc; };  // parents empty: generic; gc.fn: the size's general-contact kernel or nullptr
std::vector<KernelPack> g_packs;
std::mutex g_pack_mutex;
bool builtin_has(int n, int nb) {
    for (const auto &k : kKernels)
        if (k.n == n && k.nb == nb) return true;
    return false;
}
std::vector<int> clean_parents(const Model &m) {
    std::vector<int> p(m.parent.size());
    for (size_t i = 0; i < p.size(); i++) p[i] = m.parent[i] < 0 ? 0 : m.parent[i];
    return p;
}
unsigned tree_tag(const std::vector<int> &parents) {  // FNV-1a over the parents as little-endian 32-bit words (libdwbc_amd.build_pack names the file with it)
    unsigned h = 2166136261u;
    for (int p : parents)
        for (int b = 0; b < 4; b++) { h ^= (unsigned)((p >> (8 * b)) & 0xff); h *= 16777619u; }
    return h;
}
// nlv < 0: any level count.  tree: only a pack built for exactly these parents (or, generic = true, only a generic one)
const KernelEntry *pack_lookup(int n, int nb, int nlv, const std::vector<int> &parents, bool generic) {
    std::lock_guard<std::mutex> lk(g_pack_mutex);
    for (const auto &p : g_packs) {
        if (generic ? !p.parents.empty() : p.parents != parents) continue;
        for (int i = 0; i < p.count; i++)
            if (p.tab[i].n == n && p.tab[i].nb == nb && (nlv < 0 || p.tab[i].nlv == nlv)) return &p.tab[i];
    }
    return nullptr;
}
// the general-contact kernel of a model size: built in (TOCABI) or from a loaded pack of that size
GcEntry find_gc(int n, int nb) {  // fn == nullptr: none
    if (const GcEntry *g = lookup_gc(n, nb)) return *g;
    std::lock_guard<std::mutex> lk(g_pack_mutex);
    for (const auto &p : g_packs)
        if (p.gc.fn && p.gc.n == n && p.gc.nb == nb) return p.gc;
    return GcEntry{0, 0, nullptr, 0, nullptr, 0};
}
const KernelEntry *pack_pick(int n, int nb, int nlv, const std::vector<int> &parents) {
    if (const KernelEntry *ke = pack_lookup(n, nb, nlv, parents, false)) return ke;
    return pack_lookup(n, nb, nlv, parents, true);
}
std::string lib_dir_impl() {
    Dl_info di;
    if (dladdr((const void *)&builtin_has, &di) && di.dli_fname) {
        std::string f(di.dli_fname);
        const size_t s = f.rfind('/');
        return s == std::string::npos ? std::string(".") : f.substr(0, s);
    }
    return ".";
}
// 1 loaded, 0 no such file, -1 unusable (err set)
int try_load_pack(const std::string &path, const std::vector<int> &parents, bool want_tree, std::string &err) {
    void *dl = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!dl) return 0;
    typedef const KernelEntry *(*table_fn)(int *, unsigned *);
    typedef const int *(*parents_fn)(int *);
    table_fn tf = (table_fn)dlsym(dl, "dwbc_pack_table");
    parents_fn pf = (parents_fn)dlsym(dl, "dwbc_pack_parents");
    int count = 0;
    unsigned tag = 0;
    const KernelEntry *tab = tf ? tf(&count, &tag) : nullptr;
    if (!tab || tag != kernel_abi_tag()) {
        dlclose(dl);
        err = path + " was built from another version of the kernels: rebuild it (make -C libdwbc_amd/csrc pack ...)";
        return -1;
    }
    KernelPack kp{dl, tab, count, {}, GcEntry{0, 0, nullptr, 0}};
    typedef const void *(*gc_fn)(int *);
    if (gc_fn gf = (gc_fn)dlsym(dl, "dwbc_pack_gc")) {
        int lds = 0;
        const void *fn = gf(&lds);
        if (fn && count > 0) kp.gc = GcEntry{tab[0].n, tab[0].nb, reinterpret_cast<void (*)(const Setup, const BatchIO)>(const_cast<void *>(fn)), lds, nullptr, 0};
    }
    if (pf) {
        int pnb = 0;
        const int *pp = pf(&pnb);
        kp.parents.assign(pp, pp + pnb);
    }
    if (want_tree && kp.parents != parents) {  // a file with this tag but another tree (hash collision or a stale file): not ours
        dlclose(dl);
        return 0;
    }
    std::lock_guard<std::mutex> lk(g_pack_mutex);
    g_packs.push_back(kp);
    return 1;
}
// true when kernels for the model exist: built in, already loaded, or in a pack next to this library / under $DWBC_PACK_DIR
bool ensure_kernels(const Model &m, std::string &err) {
    const int n = m.ndof, nb = m.nb;
    if (builtin_has(n, nb)) return true;
    const std::vector<int> parents = clean_parents(m);
    if (pack_lookup(n, nb, -1, parents, false)) return true;
    char tagbuf[16];
    snprintf(tagbuf, sizeof tagbuf, "%08x", tree_tag(parents));
    const std::string base = "libdwbc_pack_" + std::to_string(n) + "_" + std::to_string(nb);
    std::vector<std::string> dirs;
    if (const char *e = getenv("DWBC_PACK_DIR")) dirs.push_back(e);
    dirs.push_back(lib_dir_impl());
    std::string tried;
    for (const auto &d : dirs) {  // the pack of this very tree first
        const int r = try_load_pack(d + "/" + base + "_t" + tagbuf + ".so", parents, true, err);
        if (r < 0) return false;
        if (r > 0) return true;
    }
    if (pack_lookup(n, nb, -1, parents, true)) return true;
    for (const auto &d : dirs) {
        const std::string path = d + "/" + base + ".so";
        const int r = try_load_pack(path, parents, false, err);
        if (r < 0) return false;
        if (r > 0) return true;
        tried += " " + path;
    }
    err = "no kernel for a model with " + std::to_string(n) + " dof / " + std::to_string(nb) + " bodies: the cycle kernels are compiled per model size; build the pack once with"
          " `make -C libdwbc_amd/csrc pack N=" + std::to_string(n) + " NB=" + std::to_string(nb) + "` (Python: libdwbc_amd.build_pack(model)); looked for" + tried;
    return false;
}
}  // namespace

dwbc_batch *dwbc_batch_create(const dwbc_model *m, int B, int device, int dtype) {
    if (!m || B < 1) { g_err = "bad arguments"; return nullptr; }
    if (dtype != DWBC_F64 && dtype != DWBC_F32) { g_err = "dtype must be DWBC_F64 or DWBC_F32"; return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_err = "no HIP device: libdwbc_hip has no CPU path"; return nullptr; }
    if (device < 0 || device >= ndev) { g_err = "bad device index"; return nullptr; }
    if (m->m.ndof != m->m.nb + 5 || m->m.ndof > 50 || m->m.nb > kMaxBodies) {
        g_err = "model outside the kernels' range (floating base + one revolute joint per body, at most 50 dof)";
        return nullptr;
    }
    if (!ensure_kernels(m->m, g_err)) return nullptr;
    if (dtype == DWBC_F32 && !builtin_has(m->m.ndof, m->m.nb)) { g_err = "kernel packs are fp64 only"; return nullptr; }
    auto *b = new dwbc_batch();
    b->model = m;
    b->B = B;
    b->device = device;
    b->dtype = dtype;
    b->n = m->m.ndof;
    b->m = b->n - 6;
    b->kern = nullptr;  // picked at the first solve (pick_kernel: the level count is not known yet)
    b->dl = DumpLayout::make(b->n);
    setup_init(b->su, m->m.nb, b->n, m->m.maxdepth);
    auto bad = [&](const char *what, hipError_t e) {
        g_err = std::string(what) + ": " + hipGetErrorString(e);
        dwbc_batch_destroy(b);
        return (dwbc_batch *)nullptr;
    };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return bad("hipSetDevice", e);
    std::vector<double> body;
    std::vector<int> topo;
    m->m.body_table(body);
    m->m.topo_table(topo);
    setup_set_parents(b->su, topo.data());
    if (getenv("DWBC_DENSE_SWEEP")) b->su.topo_kind = 0;  // test hook: the generic (dense) A^-1 sweep on a model that has a constant tree
    if ((e = hipMalloc(&b->d_body, body.size() * sizeof(double))) != hipSuccess) return bad("hipMalloc", e);
    if ((e = hipMalloc(&b->d_topo, topo.size() * sizeof(int))) != hipSuccess) return bad("hipMalloc", e);
    if ((e = hipMemcpy(b->d_body, body.data(), body.size() * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) return bad("hipMemcpy", e);
    if ((e = hipMemcpy(b->d_topo, topo.data(), topo.size() * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) return bad("hipMemcpy", e);
    if ((e = hipMalloc(&b->d_q, (size_t)B * (b->n + 1) * sizeof(double))) != hipSuccess) return bad("hipMalloc", e);
    b->own_q = true;
    if ((e = hipMalloc(&b->d_tau, (size_t)B * 3 * b->m * sizeof(double))) != hipSuccess) return bad("hipMalloc", e);
    b->own_tau = true;
    if ((e = hipMalloc(&b->d_wrench, (size_t)B * 12 * sizeof(double))) != hipSuccess) return bad("hipMalloc", e);
    b->own_wrench = true;
    if ((e = hipMalloc(&b->d_status, (size_t)B * sizeof(int))) != hipSuccess) return bad("hipMalloc", e);
    b->own_status = true;
    if ((e = hipMalloc(&b->d_diag, (size_t)B * DG_COUNT * sizeof(int))) != hipSuccess) return bad("hipMalloc", e);
    hipMemset(b->d_diag, 0, (size_t)B * DG_COUNT * sizeof(int));
    hipMemset(b->d_status, 0, (size_t)B * sizeof(int));
    b->h_q.assign((size_t)B * (b->n + 1), 0.0);
    return b;
}

void dwbc_batch_destroy(dwbc_batch *b) {
    if (!b) return;
    hipSetDevice(b->device);
    if (b->own_q) hipFree(b->d_q);
    if (b->d_qdot) hipFree(b->d_qdot);
    if (b->f_body) hipFree(b->f_body);
    if (b->d_traj) hipFree(b->d_traj);
    if (b->d_ctime) hipFree(b->d_ctime);
    if (b->d_custom) hipFree(b->d_custom);
    if (b->own_fstar) hipFree(b->d_fstar);
    if (b->own_flags) hipFree(b->d_flags);
    if (b->own_tau) hipFree(b->d_tau);
    if (b->own_wrench) hipFree(b->d_wrench);
    if (b->own_status) hipFree(b->d_status);
    hipFree(b->d_diag);
    hipFree(b->d_total);
    for (int l = 0; l < kMaxLevels; l++) hipFree(b->d_jacc[l]);
    hipFree(b->d_jacc_status);
    hipFree(b->d_rrec);
    hipFree(b->d_jacc_nc);
    hipFree(b->d_jacc_nc_status);
    hipFree(b->d_dump);
    hipFree(b->d_body);
    hipFree(b->d_topo);
    if (b->ev_upload) (void)hipEventDestroy(b->ev_upload);
    delete b;
}

int dwbc_batch_size(const dwbc_batch *b) { return b->B; }

int dwbc_batch_add_contact(dwbc_batch *b, int link, int contact_type, const double point[3], double lx, double ly, double mu,
                           double mu_z) {
    std::string err;
    const int i = setup_add_contact(b->su, link, contact_type, point, lx, ly, mu, mu_z, err);
    if (i < 0) { fail(err); return -1; }
    b->h_flags.assign((size_t)b->B * b->su.n_contacts, 0);
    b->dirty_flags = true;
    return i;
}

int dwbc_batch_clear_contacts(dwbc_batch *b) {
    b->su.n_contacts = 0;
    b->h_flags.clear();
    return 1;
}

int dwbc_batch_add_task(dwbc_batch *b, int level, int mode, int link, const double point[3]) {
    std::string err;
    if (!setup_add_task(b->su, level, mode, link, point, err)) return fail(err);
    b->h_fstar.assign((size_t)b->B * b->su.fstar_total, 0.0);
    b->dirty_fstar = true;
    return 1;
}

int dwbc_batch_add_custom_task(dwbc_batch *b, int level, int task_dof) {
    std::string err;
    if (!setup_add_custom_task(b->su, level, task_dof, err)) return fail(err);
    b->h_fstar.assign((size_t)b->B * b->su.fstar_total, 0.0);
    b->dirty_fstar = true;
    b->h_custom.assign((size_t)b->B * b->su.n_custom * kMaxTaskDof * b->n, 0.0);
    if (b->d_custom) { hipFree(b->d_custom); b->d_custom = nullptr; }
    b->dirty_custom = true;
    return 1;
}

int dwbc_batch_set_custom_task(dwbc_batch *b, int level, const double *fstar, const double *J) {
    if (level < 0 || level >= b->su.n_levels) return fail("ERROR : task space size overflow");
    const int slot = b->su.t_custom_slot[level];
    if (slot < 0) return fail("not a TASK_CUSTOM level");
    if (!J) return fail("J_task is NULL");
    if (fstar && !dwbc_batch_set_fstar(b, level, fstar)) return 0;
    const int t = b->su.t_dof[level], n = b->n, ns = b->su.n_custom;
    const size_t stride = (size_t)kMaxTaskDof * n;
    for (int i = 0; i < b->B; i++) memcpy(&b->h_custom[((size_t)i * ns + slot) * stride], J + (size_t)i * t * n, sizeof(double) * t * n);
    b->dirty_custom = true;
    return 1;
}

int dwbc_batch_clear_tasks(dwbc_batch *b) {
    b->su.n_levels = 0;
    b->su.has_com_task = 0;
    b->su.n_custom = 0;
    for (int l = 0; l < kMaxLevels; l++) b->su.t_custom_slot[l] = -1;
    b->h_custom.clear();
    setup_fstar_layout(b->su);
    b->h_fstar.clear();
    b->su.n_traj = 0;
    for (int l = 0; l < kMaxLevels; l++)
        for (int j = 0; j < kMaxTaskLinks; j++) b->su.t_traj_slot[l][j] = -1;
    b->h_traj.clear();
    return 1;
}

int dwbc_batch_set_task_gain(dwbc_batch *b, int level, int link_index, const double *pos_p, const double *pos_d, const double *pos_a,
                             const double *rot_p, const double *rot_d, const double *rot_a) {
    if (level < 0 || level >= b->su.n_levels || link_index < 0 || link_index >= b->su.t_nlinks[level]) return fail("bad task level / link index");
    (void)rot_a;  // stored by the reference, never read by GetFstarRotPD (src/task.cpp:338)
    double *g = b->su.t_gain[level][link_index];
    for (int a = 0; a < 3; a++) { g[a] = pos_p[a]; g[3 + a] = pos_d[a]; g[6 + a] = pos_a[a]; g[9 + a] = rot_p[a]; g[12 + a] = rot_d[a]; }
    return 1;
}

int dwbc_batch_set_trajectory(dwbc_batch *b, int level, int link_index, const double *traj) {
    if (level < 0 || level >= b->su.n_levels || link_index < 0 || link_index >= b->su.t_nlinks[level]) return fail("bad task level / link index");
    int &slot = b->su.t_traj_slot[level][link_index];
    if (!traj) {  // back to SetTaskSpace values for this link (slots of other links keep their place)
        slot = -1;
        return 1;
    }
    if (slot < 0) {
        if (b->su.n_traj >= kMaxLevels * kMaxTaskLinks) return fail("too many trajectories");
        // records are instance-major: re-stride the host copy for the new slot count
        const int old = b->su.n_traj, now = old + 1;
        std::vector<double> h((size_t)b->B * now * kTrajStride, 0.0);
        for (int i = 0; i < b->B; i++)
            for (int sidx = 0; sidx < old; sidx++)
                memcpy(&h[((size_t)i * now + sidx) * kTrajStride], &b->h_traj[((size_t)i * old + sidx) * kTrajStride], kTrajStride * sizeof(double));
        b->h_traj.swap(h);
        slot = old;
        b->su.n_traj = now;
        if (b->d_traj) { hipFree(b->d_traj); b->d_traj = nullptr; }
    }
    const int ns = b->su.n_traj;
    for (int i = 0; i < b->B; i++)
        memcpy(&b->h_traj[((size_t)i * ns + slot) * kTrajStride], traj + (size_t)i * kTrajStride, kTrajStride * sizeof(double));
    b->dirty_traj = true;
    return 1;
}

int dwbc_batch_set_control_time(dwbc_batch *b, const double *t) {
    if (!t) return fail("control time is NULL");
    b->h_ctime.assign(t, t + b->B);
    b->dirty_ctime = true;
    return 1;
}

int dwbc_batch_set_torque_limit(dwbc_batch *b, const double *tau_lim) {
    if (!tau_lim) { b->su.has_tau_lim = 0; return 1; }
    b->su.has_tau_lim = 1;
    for (int i = 0; i < b->m; i++) b->su.tau_lim[i] = tau_lim[i];
    return 1;
}

int dwbc_batch_fstar_size(const dwbc_batch *b) { return b->su.fstar_total; }
int dwbc_batch_task_dof(const dwbc_batch *b, int level) { return (level >= 0 && level < b->su.n_levels) ? b->su.t_dof[level] : 0; }

// An upload of the page-locked mirrors may still be in flight (hipMemcpyAsync returns at once): wait for it before a mirror is
// rewritten, so that a pipelined `solve(); set_state(next); solve();` never tears the inputs of the first solve.
static void wait_uploads(dwbc_batch *b) {
    if (!b->upload_pending) return;
    if (b->ev_upload) (void)hipEventSynchronize(b->ev_upload);
    b->upload_pending = false;
}

int dwbc_batch_set_state(dwbc_batch *b, const double *q, const double *qdot, const double *qddot) {
    (void)qddot;  // the reference hands it to RBDL's UpdateKinematicsCustom only; nothing on this path reads accelerations
    if (!q) return fail("q is NULL");
    if (!b->own_q) return fail("q is bound to a device buffer");
    wait_uploads(b);
    if (q != b->h_q.data()) memcpy(b->h_q.data(), q, b->h_q.size() * sizeof(double));  // (dwbc_batch_host_ptr: already in place)
    b->dirty_q = true;
    if (qdot) {  // B_, link velocities (dump record) and the on-device task reference need it; the torque path does not
        b->h_qdot.assign(qdot, qdot + (size_t)b->B * b->n);
        b->dirty_qdot = true;
    }
    return 1;
}

int dwbc_batch_set_contact(dwbc_batch *b, const uint8_t *flags) {
    if (b->su.n_contacts == 0) return fail("Contact Constraint size mismatch");  // include/dwbc.h:438-441
    if (b->d_flags && !b->own_flags) return fail("contact flags are bound to a device buffer");
    // the product kernels stack two simultaneous 6D contacts, the general-contact kernel three (the reference: any number,
    // src/dwbc.cpp:445-453); an instance with more than the batch is set up for is refused here instead of failing on the device
    const int ncn = b->su.n_contacts;
    for (int i = 0; i < b->B; i++) {
        int on = 0;
        for (int c = 0; c < ncn; c++) on += flags[(size_t)i * ncn + c] ? 1 : 0;
        if (on > b->max_active)
            return fail(b->max_active > 2 ? "more than 3 simultaneously active contacts in one instance: not supported by the device path"
                                          : "more than 2 simultaneously active contacts in one instance: call dwbc_batch_set_max_active_contacts(b, 3) first (3 is the most the device path stacks)");
    }
    wait_uploads(b);
    if (flags != b->h_flags.data()) memcpy(b->h_flags.data(), flags, b->h_flags.size());
    b->dirty_flags = true;
    return 1;
}

// SetContact with more than two flags raised (reference src/dwbc.cpp:445-453 stacks every flagged contact): n = 3 routes the
// batch's solves through the general-contact kernel (dwbc_cycle_gc.h) and widens the wrench output to 6 n doubles per instance
int dwbc_batch_set_max_active_contacts(dwbc_batch *b, int n) {
    if (n != 2 && n != kGcContacts) return fail("max active contacts: 2 (default) or 3");
    if (n == b->max_active) return 1;
    if (n > 2) {
        if (b->dtype == DWBC_F32) return fail("three active contacts: fp64 batches only");
        if (!find_gc(b->n, b->su.nb).fn) return fail("no general-contact kernel for this model size (built in for TOCABI; kernel packs carry one for models of at most 40 dof)");
    }
    if (!b->own_wrench) return fail("wrench is bound to a device buffer: set the contact capacity before binding");
    if (n < b->max_active) {  // lowering the capacity: the flags already set must fit it (they were validated against the old one)
        const int ncn = b->su.n_contacts;
        for (int i = 0; i < b->B && ncn > 0 && !b->h_flags.empty(); i++) {
            int on = 0;
            for (int c = 0; c < ncn; c++) on += b->h_flags[(size_t)i * ncn + c] ? 1 : 0;
            if (on > n) return fail("max active contacts: the contact flags of this batch hold an instance with more active contacts than the new capacity");
        }
    }
    HIP_OK(hipSetDevice(b->device));  // (the new buffer must live on the batch's device whatever device is current in the caller's thread)
    HIP_OK(hipStreamSynchronize(b->stream));
    // the new buffer first, the swap only on success: a failed allocation leaves the batch as it was
    double *nw = nullptr;
    HIP_OK(hipMalloc(&nw, (size_t)b->B * 6 * n * sizeof(double)));
    if (hipMemset(nw, 0, (size_t)b->B * 6 * n * sizeof(double)) != hipSuccess) {
        (void)hipFree(nw);
        return fail("max active contacts: hipMemset of the new wrench buffer failed");
    }
    (void)hipFree(b->d_wrench);
    b->d_wrench = nw;
    b->max_active = n;
    return 1;
}
int dwbc_batch_max_active_contacts(const dwbc_batch *b) { return b->max_active; }

int dwbc_batch_set_fstar(dwbc_batch *b, int level, const double *fstar) {
    if (level < 0 || level >= b->su.n_levels) return fail("ERROR : task space size overflow");  // src/dwbc.cpp:668-671
    if (b->d_fstar && !b->own_fstar) return fail("f* is bound to a device buffer");
    const int t = b->su.t_dof[level], off = b->su.fstar_off[level], F = b->su.fstar_total;
    wait_uploads(b);
    if (fstar != b->h_fstar.data() + off)  // (a caller that filled the mirror in place passes host_ptr + off)
        for (int i = 0; i < b->B; i++) memcpy(b->h_fstar.data() + (size_t)i * F + off, fstar + (size_t)i * t, sizeof(double) * t);
    b->dirty_fstar = true;
    return 1;
}

void *dwbc_batch_host_ptr(dwbc_batch *b, int field) {
    wait_uploads(b);  // the caller is about to write into the mirror
    switch (field) {
        case DWBC_IN_Q: return b->h_q.empty() ? nullptr : b->h_q.data();
        case DWBC_IN_CONTACT: return b->h_flags.empty() ? nullptr : b->h_flags.data();
        case DWBC_IN_FSTAR: return b->h_fstar.empty() ? nullptr : b->h_fstar.data();
        default: return nullptr;
    }
}

int dwbc_batch_bind_device(dwbc_batch *b, int field, void *p) {
    if (!p) return fail("NULL device pointer");
    hipSetDevice(b->device);
    switch (field) {
        case DWBC_IN_Q: if (b->own_q) hipFree(b->d_q); b->d_q = (double *)p; b->own_q = false; b->dirty_q = false; return 1;
        case DWBC_IN_CONTACT: if (b->own_flags) hipFree(b->d_flags); b->d_flags = (unsigned char *)p; b->own_flags = false; b->dirty_flags = false; return 1;
        case DWBC_IN_FSTAR: if (b->own_fstar) hipFree(b->d_fstar); b->d_fstar = (double *)p; b->own_fstar = false; b->dirty_fstar = false; return 1;
        case DWBC_TAU: if (b->own_tau) hipFree(b->d_tau); b->d_tau = (double *)p; b->own_tau = false; return 1;
        case DWBC_WRENCH: if (b->own_wrench) hipFree(b->d_wrench); b->d_wrench = (double *)p; b->own_wrench = false; return 1;
        case DWBC_STATUS: if (b->own_status) hipFree(b->d_status); b->d_status = (int *)p; b->own_status = false; return 1;
        default: return fail("field cannot be bound");
    }
}

int dwbc_batch_set_stream(dwbc_batch *b, void *s) { b->stream = (hipStream_t)s; return 1; }

int dwbc_batch_enable_dump(dwbc_batch *b, int on) {
    hipSetDevice(b->device);
    if (on && !b->d_dump) HIP_OK(hipMalloc(&b->d_dump, (size_t)b->B * b->dl.total * sizeof(double)));
    b->dump_on = on != 0;
    return 1;
}

static int upload_inputs(dwbc_batch *b) {
    if (b->su.n_contacts > 0 && (!b->d_flags || (b->own_flags && b->flags_alloc != b->su.n_contacts))) {
        if (b->own_flags && b->d_flags) hipFree(b->d_flags);
        HIP_OK(hipMalloc(&b->d_flags, (size_t)b->B * b->su.n_contacts));
        b->own_flags = true;
        b->flags_alloc = b->su.n_contacts;
        b->dirty_flags = true;
    }
    if (b->su.fstar_total > 0 && (!b->d_fstar || (b->own_fstar && b->fstar_alloc != b->su.fstar_total))) {
        if (b->own_fstar && b->d_fstar) hipFree(b->d_fstar);
        HIP_OK(hipMalloc(&b->d_fstar, (size_t)b->B * b->su.fstar_total * sizeof(double)));
        b->own_fstar = true;
        b->fstar_alloc = b->su.fstar_total;
        b->dirty_fstar = true;
    }
    if (b->dirty_traj && b->su.n_traj > 0) {
        if (!b->d_traj) HIP_OK(hipMalloc(&b->d_traj, b->h_traj.size() * sizeof(double)));
        HIP_OK(hipMemcpyAsync(b->d_traj, b->h_traj.data(), b->h_traj.size() * sizeof(double), hipMemcpyHostToDevice, b->stream));
        b->dirty_traj = false;
    }
    if (b->dirty_custom && b->su.n_custom > 0) {
        if (!b->d_custom) HIP_OK(hipMalloc(&b->d_custom, b->h_custom.size() * sizeof(double)));
        HIP_OK(hipMemcpyAsync(b->d_custom, b->h_custom.data(), b->h_custom.size() * sizeof(double), hipMemcpyHostToDevice, b->stream));
        b->dirty_custom = false;
    }
    if (b->dirty_ctime) {
        if (!b->d_ctime) HIP_OK(hipMalloc(&b->d_ctime, (size_t)b->B * sizeof(double)));
        HIP_OK(hipMemcpyAsync(b->d_ctime, b->h_ctime.data(), b->h_ctime.size() * sizeof(double), hipMemcpyHostToDevice, b->stream));
        b->dirty_ctime = false;
    }
    const bool qdot_copied = b->dirty_qdot;
    if (b->dirty_qdot) {
        if (!b->d_qdot) HIP_OK(hipMalloc(&b->d_qdot, (size_t)b->B * b->n * sizeof(double)));
        HIP_OK(hipMemcpyAsync(b->d_qdot, b->h_qdot.data(), b->h_qdot.size() * sizeof(double), hipMemcpyHostToDevice, b->stream));
        b->dirty_qdot = false;
    }
    if (b->dirty_q && b->own_q) HIP_OK(hipMemcpyAsync(b->d_q, b->h_q.data(), b->h_q.size() * sizeof(double), hipMemcpyHostToDevice, b->stream));
    if (b->dirty_flags && b->own_flags) HIP_OK(hipMemcpyAsync(b->d_flags, b->h_flags.data(), b->h_flags.size(), hipMemcpyHostToDevice, b->stream));
    if (b->dirty_fstar && b->own_fstar) HIP_OK(hipMemcpyAsync(b->d_fstar, b->h_fstar.data(), b->h_fstar.size() * sizeof(double), hipMemcpyHostToDevice, b->stream));
    const bool copied = (b->dirty_q && b->own_q) || (b->dirty_flags && b->own_flags) || (b->dirty_fstar && b->own_fstar) || qdot_copied;
    b->dirty_q = b->dirty_flags = b->dirty_fstar = false;
    if (copied) {
        if (!b->ev_upload) HIP_OK(hipEventCreateWithFlags(&b->ev_upload, hipEventDisableTiming));
        HIP_OK(hipEventRecord(b->ev_upload, b->stream));
        b->upload_pending = true;
    }
    return 1;
}

// the lean instantiation (EXTRAS = false) serves every launch that uses none of the optional paths
static bool lean_ok(const dwbc_batch *b) {
    return b->hqp && !b->warm && b->su.n_traj == 0 && !b->su.has_com_task && b->su.n_custom == 0 && !b->dump_on && !getenv("DWBC_NO_LEAN");
}

// dynamic LDS of one flavour of an entry: the lean capped build may be laid out on the compact map (Lds3)
static int entry_lds(const KernelEntry *ke, void (*fn)(const Setup, const BatchIO)) {
    return (fn == ke->fn_lean && ke->lds_bytes_lean) ? ke->lds_bytes_lean : ke->lds_bytes;
}

// the paired kernel serves the lean full-model cycle of small batches (one instance per SIMD); DWBC_NO_PAIR switches it off
static bool pair_ok(const dwbc_batch *b, const KernelEntry *ke, bool wide, bool lean, bool reduced) {
    return ke && ke->fn_pair && wide && lean && !reduced && b->dtype != DWBC_F32 && !getenv("DWBC_NO_PAIR");
}

static const KernelEntry *pick_kernel(const dwbc_batch *b, bool reduced) {
    const int which = reduced ? 2 : 0;
    if (const KernelEntry *ke = lookup_kernel(b->n, b->su.nb, b->su.n_levels, which, b->su.topo_kind)) return ke;
    return reduced ? nullptr : pack_pick(b->n, b->su.nb, b->su.n_levels, clean_parents(b->model->m));  // packs hold the full-model cycle only
}

// fp32 launch: the fp32 kernels read and write the double buffers of the boundary themselves (io_t); only the model table is
// kept in float
static int launch_f32(dwbc_batch *b, bool reduced) {
    const int which = reduced ? 2 : 0;
    const int lean = lean_ok(b) ? 1 : 0;
    const int key = ((which * 16 + b->su.n_levels) * 2 + lean) * 2 + (b->su.topo_kind ? 1 : 0);
    if (key != b->f32_key) {
        if (!dwbc_f32_lookup(b->n, b->su.nb, b->su.n_levels, which, lean, b->su.topo_kind, &b->f32_fn, &b->f32_fn_wide, &b->f32_lds, &b->f32_lds_wide, &b->f32_topo))
            return fail("no fp32 kernel for this model / number of task levels");
        HIP_OK(hipFuncSetAttribute(b->f32_fn, hipFuncAttributeMaxDynamicSharedMemorySize, b->f32_lds));
        if (b->f32_fn_wide) HIP_OK(hipFuncSetAttribute(b->f32_fn_wide, hipFuncAttributeMaxDynamicSharedMemorySize, b->f32_lds_wide));
        hipDeviceProp_t prop;
        HIP_OK(hipGetDeviceProperties(&prop, b->device));
        b->n_cu = prop.multiProcessorCount;
        b->f32_key = key;
    }
    if (b->dump_on) return fail("the dump record is not available on DWBC_F32 batches");
    if (!b->f_body) {
        std::vector<double> body;
        b->model->m.body_table(body);
        HIP_OK(hipMalloc(&b->f_body, body.size() * sizeof(float)));
        hipLaunchKernelGGL(dwbc_cvt_d2f, dim3((unsigned)((body.size() + 255) / 256)), dim3(256), 0, b->stream, b->d_body, b->f_body, body.size());
    }
    struct IoF32 {  // BatchIO of the fp32 namespace: same members, real_t = float
        int B;
        const double *q, *qdot;
        const unsigned char *flags;
        const double *fstar, *traj, *ctime, *custom_J;
        double *tau, *wrench;
        int *status, *diag;
        float *dump;
        const float *body;
        const int *topo;
        int hqp, pair_swap_bit, warm, wrench_ld;
    } io{};
    static_assert(sizeof(IoF32) == sizeof(BatchIO), "BatchIO layouts of the two builds must match");
    io.B = b->B;
    io.q = b->d_q;
    io.qdot = b->d_qdot;
    io.traj = b->su.n_traj > 0 ? b->d_traj : nullptr;
    io.ctime = b->d_ctime;
    io.custom_J = b->su.n_custom > 0 ? b->d_custom : nullptr;
    io.flags = b->d_flags;
    io.fstar = b->d_fstar;
    io.tau = b->d_tau;
    io.wrench = b->d_wrench;
    io.status = b->d_status;
    io.diag = b->d_diag;
    io.dump = nullptr;
    io.body = b->f_body;
    io.topo = b->d_topo;
    io.hqp = b->hqp;
    io.pair_swap_bit = -1;
    io.warm = (b->warm && b->ws_valid) ? 1 : 0;
    b->ws_valid = !lean;  // the full build leaves every QP's working set in the diagnostics record
    const bool wide = b->f32_fn_wide && b->B <= 4 * b->n_cu && !getenv("DWBC_NO_WIDE");
    void *args[] = {(void *)&b->su, (void *)&io};
    HIP_OK(hipLaunchKernel(wide ? b->f32_fn_wide : b->f32_fn, dim3(b->B), dim3(kNT), args, wide ? b->f32_lds_wide : b->f32_lds, b->stream));
    return 1;
}

// three active contacts, or a task level of more than six dof: every instance of the batch goes through the general-contact kernel
// (lean scope); its TG = 12 instantiation when a level is wider than six
static int launch_gc(dwbc_batch *b) {
    const GcEntry gc_ = find_gc(b->n, b->su.nb), *g = &gc_;
    const bool wide_tasks = setup_wide_tasks(b->su);
    if (!g->fn) return fail("no general-contact kernel for this model size (built in for TOCABI; kernel packs carry one for models of at most 40 dof)");
    if (wide_tasks && !g->fn_wide_tasks) return fail("task levels of more than 6 dof: built in for TOCABI's size only");
    if (!b->hqp) return fail("three active contacts / task levels of more than 6 dof: hqp = true only (the reference's closed-form redistribution is written for two contacts, src/dwbc.cpp:1570-1619)");
    if (b->su.n_traj > 0 || b->su.has_com_task || b->su.n_custom > 0 || b->dump_on)
        return fail("three active contacts / task levels of more than 6 dof: link tasks with f* from SetTaskSpace only (no trajectories, COM or custom levels, no dump record)");
    auto fn = wide_tasks ? g->fn_wide_tasks : g->fn;
    const int lds_bytes = wide_tasks ? g->lds_bytes_wide_tasks : g->lds_bytes;
    if (b->gc_attr_set != (wide_tasks ? 2 : 1)) {
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        b->gc_attr_set = wide_tasks ? 2 : 1;
    }
    BatchIO io{};
    io.B = b->B;
    io.q = b->d_q;
    io.flags = b->d_flags;
    io.fstar = b->d_fstar;
    io.tau = b->d_tau;
    io.wrench = b->d_wrench;
    io.wrench_ld = 6 * b->max_active;
    io.status = b->d_status;
    io.diag = b->d_diag;
    io.body = b->d_body;
    io.topo = b->d_topo;
    io.hqp = 1;
    b->ws_valid = false;  // cold-started QPs, no working sets kept
    hipLaunchKernelGGL(fn, dim3(b->B), dim3(kNT), lds_bytes, b->stream, b->su, io);
    return hipGetLastError() == hipSuccess ? 1 : fail("general-contact kernel launch failed");
}

static int launch(dwbc_batch *b, bool reduced = false) {
    if (b->max_active > 2 || setup_wide_tasks(b->su)) {
        if (reduced) return fail("three active contacts / task levels of more than 6 dof: not built on the reduced dynamics path");
        if (b->dtype == DWBC_F32) return fail("three active contacts / task levels of more than 6 dof: fp64 batches only");
        return launch_gc(b);
    }
    if (b->dtype == DWBC_F32) return launch_f32(b, reduced);
    const KernelEntry *ke = pick_kernel(b, reduced);
    if (!ke) return fail("no kernel for this model / number of task levels");
    if (ke != b->kern) {
        b->kern = ke;
        b->attr_set = false;
    }
    BatchIO io{};
    io.B = b->B;
    io.q = b->d_q;
    io.qdot = b->d_qdot;
    io.traj = b->su.n_traj > 0 ? b->d_traj : nullptr;
    io.ctime = b->d_ctime;
    io.custom_J = b->su.n_custom > 0 ? b->d_custom : nullptr;
    io.flags = b->d_flags;
    io.fstar = b->d_fstar;
    io.tau = b->d_tau;
    io.wrench = b->d_wrench;
    io.status = b->d_status;
    io.diag = b->d_diag;
    io.dump = b->dump_on ? b->d_dump : nullptr;
    io.body = b->d_body;
    io.topo = b->d_topo;
    io.hqp = b->hqp;
    io.warm = (b->warm && b->ws_valid) ? 1 : 0;
    if (!b->attr_set) {
        for (auto fn : {b->kern->fn, b->kern->fn_wide, b->kern->fn_lean, b->kern->fn_wide_lean})
            if (fn) HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, entry_lds(b->kern, fn)));
        if (b->kern->fn_pair)
            HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(b->kern->fn_pair), hipFuncAttributeMaxDynamicSharedMemorySize, b->kern->lds_bytes_pair));
        hipDeviceProp_t prop;
        HIP_OK(hipGetDeviceProperties(&prop, b->device));
        b->n_cu = prop.multiProcessorCount;
        b->attr_set = true;
    }
    const bool wide = b->kern->fn_wide && b->B <= 4 * b->n_cu && !getenv("DWBC_NO_WIDE");
    const bool lean = b->kern->fn_lean && lean_ok(b);
    b->ws_valid = !lean && !reduced;  // the full build leaves every QP's working set in the diagnostics record (DG_QP_ACT)
    // (DWBC_PAIR_ALWAYS=1: development switch, the two-wave kernel at any batch size)
    if (pair_ok(b, b->kern, wide || getenv("DWBC_PAIR_ALWAYS"), lean, reduced)) {
        // two waves per instance, side chains on the helper wave (dwbc_cycle2p.h).  Which wave of a workgroup is the main one can be
        // swapped per workgroup (DWBC_PAIR_SWAP_BIT = bit of the workgroup index) to steer the main waves of a CU's four workgroups
        // onto different SIMDs; measured at B = 1024 (profiles/r03e_pair_roles.txt): no swap 82.3 us per launch, bit 8 / bit 9 99 - 102 us
        // (the dispatcher already spreads wave 0 of consecutive workgroups), so the default is no swap.
        const char *sb = getenv("DWBC_PAIR_SWAP_BIT");
        io.pair_swap_bit = sb ? atoi(sb) : -1;
        hipLaunchKernelGGL(b->kern->fn_pair, dim3(b->B), dim3(2 * kNT), b->kern->lds_bytes_pair, b->stream, b->su, io);
        return hipGetLastError() == hipSuccess ? 1 : fail("paired kernel launch failed");
    }
    auto fn = wide ? (lean ? b->kern->fn_wide_lean : b->kern->fn_wide) : (lean ? b->kern->fn_lean : b->kern->fn);
    hipLaunchKernelGGL(fn, dim3(b->B), dim3(kNT), entry_lds(b->kern, fn), b->stream, b->su, io);
    HIP_OK(hipGetLastError());
    return 1;
}

int dwbc_batch_solve(dwbc_batch *b, unsigned flags) {
    b->hqp = (flags & DWBC_SOLVE_HQP) ? 1 : 0;
    // init = false (DWBC_SOLVE_INIT clear): hot start from the working sets of the previous solve (src/dwbc.cpp:1064-1074).  The
    // first solve of a batch, and the first after the lean kernel ran, has no working set to start from and runs cold.
    b->warm = (flags & DWBC_SOLVE_INIT) ? 0 : 1;
    if (!b->hqp && (flags & DWBC_SOLVE_REDUCED)) return fail("hqp=false is not built on the reduced dynamics path");
    if (b->su.n_levels < 1) return fail("no task space");
    if (b->su.n_contacts < 1) return fail("no contact constraint");
    const bool reduced = flags & DWBC_SOLVE_REDUCED;
    b->last_reduced = reduced;
    if (reduced && b->su.n_custom > 0) return fail("TASK_CUSTOM levels are not built on the reduced dynamics path");
    if (reduced && b->su.has_tau_lim)
        return fail("reduced dynamics path with a torque limit is inconsistent in the reference (src/dwbc.cpp:3462-3467,3513; "
                    "its harness disables the limit, tests/sp_test/redu_dyn_test.cpp:63): call dwbc_batch_set_torque_limit(b, NULL)");
    HIP_OK(hipSetDevice(b->device));
    if (!upload_inputs(b)) return 0;
    return launch(b, reduced);
}

int dwbc_batch_copy_kinematics(dwbc_batch *dst, const dwbc_batch *src) {
    // RobotData::CopyKinematicsData (reference src/dwbc.cpp:1711-1762): state, contacts (with their flags), task spaces (with
    // their f*), torque limit and control time go to the target object, which then runs its own Calc* sequence.  Everything
    // derived (A_, A_inv_, link_, G_, CMM_, B_) is recomputed by the fused kernel from the copied state.
    if (!dst || !src) return fail("NULL batch");
    if (dst == src) return 1;
    if (dst->B != src->B || dst->n != src->n || dst->su.nb != src->su.nb) return fail("CopyKinematicsData: batch size / model mismatch");
    HIP_OK(hipSetDevice(dst->device));
    wait_uploads(dst);  // the target's mirrors are rewritten below
    dst->su = src->su;
    auto clone = [&](double *&dd, bool &own, const double *sd, size_t count, PinnedVec<double> &hd, const PinnedVec<double> &hs, bool &dirty,
                     bool src_own) -> int {
        hd = hs;
        if (count == 0) return 1;
        if (!src_own || hs.size() != count) {  // the source reads a caller-owned device buffer: device-to-device copy
            if (!own || !dd) { HIP_OK(hipMalloc(&dd, count * sizeof(double))); own = true; }
            HIP_OK(hipMemcpy(dd, sd, count * sizeof(double), hipMemcpyDeviceToDevice));
            hd.assign(count, 0.0);
            HIP_OK(hipMemcpy(hd.data(), sd, count * sizeof(double), hipMemcpyDeviceToHost));
            dirty = false;
        } else {
            dirty = true;
        }
        return 1;
    };
    if (!clone(dst->d_q, dst->own_q, src->d_q, (size_t)src->B * (src->n + 1), dst->h_q, src->h_q, dst->dirty_q, src->own_q)) return 0;
    // f* and flags: (re)allocated by upload_inputs when the layout changed
    dst->h_fstar = src->h_fstar;
    dst->h_flags = src->h_flags;
    if (dst->own_fstar && dst->d_fstar) { hipFree(dst->d_fstar); dst->d_fstar = nullptr; }
    if (dst->own_flags && dst->d_flags) { hipFree(dst->d_flags); dst->d_flags = nullptr; }
    dst->own_fstar = dst->own_flags = true;
    dst->fstar_alloc = dst->flags_alloc = 0;
    dst->dirty_fstar = dst->dirty_flags = true;
    if (!src->own_fstar && src->d_fstar && src->su.fstar_total > 0) {
        dst->h_fstar.assign((size_t)src->B * src->su.fstar_total, 0.0);
        HIP_OK(hipMemcpy(dst->h_fstar.data(), src->d_fstar, dst->h_fstar.size() * sizeof(double), hipMemcpyDeviceToHost));
    }
    if (!src->own_flags && src->d_flags && src->su.n_contacts > 0) {
        dst->h_flags.assign((size_t)src->B * src->su.n_contacts, 0);
        HIP_OK(hipMemcpy(dst->h_flags.data(), src->d_flags, dst->h_flags.size(), hipMemcpyDeviceToHost));
    }
    // the contact capacity travels with the flags (a three-contact source would otherwise hand the two-contact product kernels rows
    // with three flags raised: status 0 on every such instance); after the flags, so that lowering is checked against the copied ones
    if (dst->max_active != src->max_active && !dwbc_batch_set_max_active_contacts(dst, src->max_active)) return 0;
    dst->h_qdot = src->h_qdot; dst->dirty_qdot = !src->h_qdot.empty();
    dst->h_ctime = src->h_ctime; dst->dirty_ctime = !src->h_ctime.empty();
    dst->h_traj = src->h_traj; dst->dirty_traj = !src->h_traj.empty();
    if (dst->d_traj) { hipFree(dst->d_traj); dst->d_traj = nullptr; }
    dst->h_custom = src->h_custom; dst->dirty_custom = !src->h_custom.empty();
    if (dst->d_custom) { hipFree(dst->d_custom); dst->d_custom = nullptr; }
    return 1;
}

int dwbc_batch_sync(dwbc_batch *b) {
    HIP_OK(hipSetDevice(b->device));
    HIP_OK(hipStreamSynchronize(b->stream));
    return 1;
}

int dwbc_batch_time_solves(dwbc_batch *b, unsigned flags, int steps, float *ms) {
    HIP_OK(hipSetDevice(b->device));
    if (!dwbc_batch_solve(b, flags)) return 0;  // uploads + warm launch
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, b->stream));
    for (int i = 0; i < steps; i++)
        if (!launch(b, flags & DWBC_SOLVE_REDUCED)) return 0;
    HIP_OK(hipEventRecord(e1, b->stream));
    HIP_OK(hipEventSynchronize(e1));
    HIP_OK(hipEventElapsedTime(ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return 1;
}

size_t dwbc_batch_field_bytes(const dwbc_batch *b, int field) {
    const size_t B = b->B, n = b->n, m = b->m;
    switch (field) {
        case DWBC_IN_Q: return B * (n + 1) * 8;
        case DWBC_IN_CONTACT: return B * b->su.n_contacts;
        case DWBC_IN_FSTAR: return B * b->su.fstar_total * 8;
        case DWBC_TAU: return B * 3 * m * 8;
        case DWBC_WRENCH: return B * 6 * b->max_active * 8;
        case DWBC_STATUS: return B * 4;
        case DWBC_DIAG: return B * DG_COUNT * 4;
        case DWBC_TAU_GRAV: case DWBC_TAU_TASK: case DWBC_TAU_CONTACT: case DWBC_TAU_TOTAL: return B * m * 8;
        case DWBC_A: case DWBC_A_INV: case DWBC_A_INV_N_C: return B * n * n * 8;
        case DWBC_J_C: case DWBC_J_C_INV_T: return B * 12 * n * 8;
        case DWBC_LAMBDA_C: return B * 144 * 8;
        case DWBC_W_INV: return B * m * m * 8;
        case DWBC_NWJW: return B * m * 6 * 8;
        case DWBC_G: return B * n * 8;
        case DWBC_P_C: return B * 12 * 8;
        case DWBC_LINK_R: return B * kMaxBodies * 9 * 8;
        case DWBC_LINK_P: return B * kMaxBodies * 3 * 8;
        case DWBC_FSTAR_QP: case DWBC_CONTACT_QP: return B * kMaxLevels * 6 * 8;
        case DWBC_CF_REDIS: return B * 6 * 8;
        case DWBC_J_TASK: return B * kMaxLevels * 6 * n * 8;
        case DWBC_LAMBDA_TASK: return B * kMaxLevels * 36 * 8;
        case DWBC_J_KT: return B * kMaxLevels * m * 6 * 8;
        case DWBC_QP_VIOL: return B * (kMaxLevels + 1) * 8;
        case DWBC_DUMP_RAW: return B * (size_t)b->dl.total * 8;
        case DWBC_CMM: case DWBC_J_COM: return B * 6 * n * 8;
        case DWBC_COM: return B * 3 * 8;
        case DWBC_COM_INERTIA: return B * 9 * 8;
        case DWBC_B: return B * n * 8;
        case DWBC_CONTACT_POS: return B * kMaxActiveContacts * 3 * 8;
        case DWBC_CONTACT_ROT: return B * kMaxActiveContacts * 9 * 8;
        case DWBC_ZMP: return B * (3 + kMaxActiveContacts * 3) * 8;
        case DWBC_LINK_V: case DWBC_LINK_W: return B * kMaxBodies * 3 * 8;
        case DWBC_A_R: case DWBC_A_R_INV: return B * kMaxReducedDof * kMaxReducedDof * 8;
        case DWBC_G_R: return B * kMaxReducedDof * 8;
        case DWBC_J_I_NC: case DWBC_J_I_NC_INV_T: return B * 6 * (n - 12) * 8;
        default: return 0;
    }
}

int dwbc_batch_get(dwbc_batch *b, int field, void *out, size_t bytes) {
    const size_t need = dwbc_batch_field_bytes(b, field);
    if (need == 0) return fail("unknown field");
    if (bytes < need) return fail("output buffer too small");
    HIP_OK(hipSetDevice(b->device));
    HIP_OK(hipStreamSynchronize(b->stream));
    const size_t B = b->B, m = b->m;
    // device -> page-locked staging (asynchronous on the batch's stream, PCIe rate) -> the caller's memory
    auto d2h = [&](const void *src, size_t nbytes) -> int {
        if (b->h_stage.size() < nbytes) b->h_stage.resize(nbytes);
        HIP_OK(hipMemcpyAsync(b->h_stage.data(), src, nbytes, hipMemcpyDeviceToHost, b->stream));
        HIP_OK(hipStreamSynchronize(b->stream));
        memcpy(out, b->h_stage.data(), nbytes);
        return 1;
    };
    switch (field) {
        case DWBC_IN_Q: return d2h(b->d_q, need);
        case DWBC_IN_CONTACT: return d2h(b->d_flags, need);
        case DWBC_IN_FSTAR: return d2h(b->d_fstar, need);
        case DWBC_TAU: return d2h(b->d_tau, need);
        case DWBC_WRENCH: return d2h(b->d_wrench, need);
        case DWBC_STATUS: return d2h(b->d_status, need);
        case DWBC_DIAG: return d2h(b->d_diag, need);
        case DWBC_TAU_GRAV: case DWBC_TAU_TASK: case DWBC_TAU_CONTACT: case DWBC_TAU_TOTAL: {
            // one third of the bytes over PCIe: the part (or the sum, getTorqueCommand-style) is formed on the device
            if (!b->d_total) HIP_OK(hipMalloc(&b->d_total, B * m * sizeof(double)));
            const int sel = field == DWBC_TAU_GRAV ? 0 : field == DWBC_TAU_TASK ? 1 : field == DWBC_TAU_CONTACT ? 2 : 3;
            const size_t cnt = B * m;
            hipLaunchKernelGGL(dwbc_tau_select, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, b->stream, (const double *)b->d_tau, b->d_total, (int)m, cnt, sel);
            HIP_OK(hipGetLastError());
            return d2h(b->d_total, need);
        }
        default: break;
    }
    if (!b->d_dump || !b->dump_on) return fail("intermediates need dwbc_batch_enable_dump(b, 1) before the solve");
    const DumpLayout &dl = b->dl;
    int off = 0, len = 0;
    const int n = b->n, K = 6, T = kMaxTaskDof, L = kMaxLevels;
    switch (field) {
        case DWBC_A: off = dl.A; len = n * n; break;
        case DWBC_A_INV: off = dl.A_inv; len = n * n; break;
        case DWBC_J_C: off = dl.J_C; len = 12 * n; break;
        case DWBC_LAMBDA_C: off = dl.Lambda_c; len = 144; break;
        case DWBC_J_C_INV_T: off = dl.J_C_INV_T; len = 12 * n; break;
        case DWBC_A_INV_N_C: off = dl.A_inv_N_C; len = n * n; break;
        case DWBC_W_INV: off = dl.W_inv; len = (int)(m * m); break;
        case DWBC_NWJW: off = dl.NwJw; len = (int)m * K; break;
        case DWBC_G: off = dl.G; len = n; break;
        case DWBC_CMM: off = dl.CMM; len = 6 * n; break;
        case DWBC_J_COM: off = dl.J_com; len = 6 * n; break;
        case DWBC_COM: off = dl.com; len = 3; break;
        case DWBC_COM_INERTIA: off = dl.com_inertia; len = 9; break;
        case DWBC_B: off = dl.B; len = n; break;
        case DWBC_CONTACT_POS: off = dl.contact_pos; len = kMaxActiveContacts * 3; break;
        case DWBC_CONTACT_ROT: off = dl.contact_rot; len = kMaxActiveContacts * 9; break;
        case DWBC_ZMP: off = dl.zmp; len = 3 + kMaxActiveContacts * 3; break;
        case DWBC_LINK_V: off = dl.link_v; len = kMaxBodies * 3; break;
        case DWBC_LINK_W: off = dl.link_w; len = kMaxBodies * 3; break;
        case DWBC_P_C: off = dl.P_C; len = 12; break;
        case DWBC_LINK_R: off = dl.link_R; len = kMaxBodies * 9; break;
        case DWBC_LINK_P: off = dl.link_p; len = kMaxBodies * 3; break;
        case DWBC_FSTAR_QP: off = dl.fstar_qp; len = L * T; break;
        case DWBC_CONTACT_QP: off = dl.contact_qp; len = L * K; break;
        case DWBC_CF_REDIS: off = dl.cf_redis; len = K; break;
        case DWBC_J_TASK: off = dl.J_task; len = L * T * n; break;
        case DWBC_LAMBDA_TASK: off = dl.Lambda_task; len = L * T * T; break;
        case DWBC_J_KT: off = dl.J_kt; len = L * (int)m * T; break;
        case DWBC_QP_VIOL: off = dl.qp_viol; len = L + 1; break;
        case DWBC_DUMP_RAW: off = 0; len = dl.total; break;
        case DWBC_A_R: off = dl.A_R; len = kMaxReducedDof * kMaxReducedDof; break;
        case DWBC_A_R_INV: off = dl.A_R_inv; len = kMaxReducedDof * kMaxReducedDof; break;
        case DWBC_G_R: off = dl.G_R; len = kMaxReducedDof; break;
        case DWBC_J_I_NC: off = dl.J_I_nc; len = 6 * (n - 12); break;
        case DWBC_J_I_NC_INV_T: off = dl.J_I_nc_inv_T; len = 6 * (n - 12); break;
        default: return fail("unknown field");
    }
    HIP_OK(hipMemcpy2D(out, (size_t)len * 8, b->d_dump + off, (size_t)dl.total * 8, (size_t)len * 8, B, hipMemcpyDeviceToHost));
    return 1;
}

const char *dwbc_batch_kernel_name(const dwbc_batch *b) {
    static thread_local std::string name;
    if (b->max_active > 2 || setup_wide_tasks(b->su)) {
        name = "dwbc::dwbc_cycle_kernel_gc<" + std::to_string(b->n) + ", " + std::to_string(b->su.nb) + ", 64, " + (setup_wide_tasks(b->su) ? "12" : "6") + ">";
        return name.c_str();
    }
    const KernelEntry *ke = pick_kernel(b, b->last_reduced);
    const std::string pre = b->dtype == DWBC_F32 ? "dwbc_f32::" : "dwbc::";  // as rocprofv3 prints the instantiations
    if (!ke) return "";
    const std::string topo = ", " + pre + (ke->topo == 1 ? "TopoTocabi" : (ke->topo == 2 ? "TopoPack" : "TopoGeneric"));
    const std::string sz = std::to_string(ke->n) + ", " + std::to_string(ke->nb);
    if (b->last_reduced) {
        name = pre + "dwbc_cycle_kernel_reduced<" + sz + ", " + std::to_string(ke->nlv) + ", 64" + topo + ">";
        return name.c_str();
    }
    int n_cu = b->n_cu;
    if (!n_cu) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, b->device) == hipSuccess) n_cu = prop.multiProcessorCount;
    }
    const bool wide = ke->fn_wide && b->B <= 4 * n_cu && !getenv("DWBC_NO_WIDE");
    if (pair_ok(b, ke, wide, ke->fn_lean && lean_ok(b), false)) {
        name = pre + "dwbc_cycle_kernel_v2p<" + sz + ", " + std::to_string(ke->nlv) + topo + ">";
        return name.c_str();
    }
    name = pre + (wide ? "dwbc_cycle_kernel_v2w<" : "dwbc_cycle_kernel_v2<") + sz + ", " + std::to_string(ke->nlv) + ", 64" +
           (ke->fn_lean && lean_ok(b) ? ", false" : ", true") + topo + ((!wide && ke->fn_lean && lean_ok(b) && ke->lds_bytes_lean) ? ", true" : "") + ">";
    return name.c_str();
}

int dwbc_batch_launch_info(const dwbc_batch *b, int *threads, int *lds) {
    if (b->max_active > 2 || setup_wide_tasks(b->su)) {
        const GcEntry g = find_gc(b->n, b->su.nb);
        if (threads) *threads = kNT;
        if (lds) *lds = setup_wide_tasks(b->su) ? g.lds_bytes_wide_tasks : (g.fn ? g.lds_bytes : 0);
        return 1;
    }
    const KernelEntry *ke = pick_kernel(b, b->last_reduced);
    if (threads) *threads = kNT;
    int n_cu = b->n_cu;
    if (!n_cu) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, b->device) == hipSuccess) n_cu = prop.multiProcessorCount;
    }
    const bool wide = ke && ke->fn_wide && b->B <= 4 * n_cu && !getenv("DWBC_NO_WIDE");
    const bool compact = ke && !wide && !b->last_reduced && ke->fn_lean && lean_ok(b) && ke->lds_bytes_lean;
    if (pair_ok(b, ke, wide, ke && ke->fn_lean && lean_ok(b), b->last_reduced)) {
        if (threads) *threads = 2 * kNT;
        if (lds) *lds = ke->lds_bytes_pair;
        return 1;
    }
    if (lds) *lds = b->dtype == DWBC_F32 && b->f32_lds ? (wide ? b->f32_lds_wide : b->f32_lds) : (ke ? (compact ? ke->lds_bytes_lean : ke->lds_bytes) : 0);
    return 1;
}

}  // extern "C"

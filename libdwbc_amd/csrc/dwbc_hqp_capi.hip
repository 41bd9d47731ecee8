// dwbc_hqp_capi.hip -- C-ABI of the batched hierarchical-QP class and of the LQP configurator (include/dwbc_batch.h, section
// "generic hierarchical-QP class"): kernels of dwbc_hqp.h + host bookkeeping.  gfx950 only, no CPU path.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "dwbc_capi_internal.h"
#include "dwbc_hqp.h"

using namespace dwbc;

namespace {
int fail(const std::string &s) { return capi_fail(s); }
constexpr int kNT = 256;  // four wavefronts per instance: the matrices of this path live in HBM, more lanes keep more loads in flight
}  // namespace

__global__ __launch_bounds__(kNT) void dwbc_hqp_kernel(const HqpDesc d, const HqpIO io) {
    extern __shared__ __attribute__((aligned(16))) double hqp_lds[];
    const int inst = blockIdx.x;
    if (inst >= io.B) return;
    hqp_instance<kNT>(Thr{(int)threadIdx.x}, d, io, inst, hqp_lds);
}
__global__ __launch_bounds__(kNT) void dwbc_hqp_normalize_kernel(const HqpDesc d, const HqpIO io, int level) {
    const int inst = blockIdx.x;
    if (inst >= io.B) return;
    double *rec = io.rec + (size_t)inst * d.rec;
    const Thr th{(int)threadIdx.x};
    if (d.m[level] > 0) hqp_normalize_rows<kNT>(th, rec + d.oA[level], rec + d.oa[level], d.m[level], d.nv);
    hqp_normalize_rows<kNT>(th, rec + d.oB[level], rec + d.ob[level], d.e[level], d.nv);
}
__global__ __launch_bounds__(kNT) void dwbc_lqp_configure_kernel(const LqpCfg cfg, const HqpDesc d, const HqpIO io, const double *dump, const double *fstar) {
    const int inst = blockIdx.x;
    if (inst >= io.B) return;
    lqp_configure_instance<kNT>(Thr{(int)threadIdx.x}, cfg, d, io, dump, fstar, inst);
}
__global__ __launch_bounds__(kNT) void dwbc_lqp_torque_kernel(const LqpCfg cfg, const HqpDesc d, const HqpIO io, const double *dump, double *tau) {
    const int inst = blockIdx.x;
    if (inst >= io.B) return;
    lqp_torque_instance<kNT>(Thr{(int)threadIdx.x}, cfg, d, io, dump, tau, inst);
}

__global__ __launch_bounds__(kNT) void dwbc_jacc_configure_kernel(const LqpCfg cfg, int level, const JaccPrev prev, const HqpDesc d, const HqpIO io, const double *dump,
                                                                  const double *fstar) {
    const int inst = blockIdx.x;
    if (inst >= io.B) return;
    jacc_configure_instance<kNT>(Thr{(int)threadIdx.x}, cfg, level, prev, d, io, dump, fstar, inst);
}
__global__ __launch_bounds__(kNT) void dwbc_jacc_extract_kernel(const LqpCfg cfg, int level, const HqpDesc d, const HqpIO io, const double *dump, const double *fstar,
                                                                double *out, int *status) {
    const int inst = blockIdx.x;
    if (inst >= io.B) return;
    jacc_extract_instance<kNT>(Thr{(int)threadIdx.x}, cfg, level, d, io, dump, fstar, out, status, inst);
}

__global__ __launch_bounds__(kNT) void dwbc_reduced_record_kernel(const ReducedRecCfg rc, int B, const double *dump, double *rrec) {
    const int inst = blockIdx.x;
    if (inst >= B) return;
    reduced_record_instance<kNT>(Thr{(int)threadIdx.x}, rc, dump, rrec, inst);
}
__global__ __launch_bounds__(kNT) void dwbc_lqp_nc_configure_kernel(const NcCfg c, const HqpDesc d, const HqpIO io, const double *dump, const double *fstar, const double *prev) {
    const int inst = blockIdx.x;
    if (inst >= io.B) return;
    lqp_nc_configure_instance<kNT>(Thr{(int)threadIdx.x}, c, d, io, dump, fstar, prev, inst);
}
__global__ __launch_bounds__(kNT) void dwbc_jacc_nc_configure_kernel(const NcCfg c, const HqpDesc d, const HqpIO io, const double *dump, const double *fstar, const double *prev) {
    const int inst = blockIdx.x;
    if (inst >= io.B) return;
    jacc_nc_configure_instance<kNT>(Thr{(int)threadIdx.x}, c, d, io, dump, fstar, prev, inst);
}
__global__ __launch_bounds__(kNT) void dwbc_jacc_nc_extract_kernel(const NcCfg c, const HqpDesc d, const HqpIO io, const double *dump, const double *fstar, const double *prev,
                                                                   double *out, int *status) {
    const int inst = blockIdx.x;
    if (inst >= io.B) return;
    jacc_nc_extract_instance<kNT>(Thr{(int)threadIdx.x}, c, d, io, dump, fstar, prev, out, status, inst);
}

struct dwbc_hqp {
    int B = 0, device = 0;
    int acc = 0, torque = 0, contact = 0;
    HqpDesc d{};
    bool laid_out = false, share_cost = false;
    // host staging of what was handed over before the layout is fixed (dwbc_hqp_prepare), per level
    struct Stage { std::vector<double> A, a, Bm, b, H, y, v; bool normalize = false; };
    std::vector<Stage> stage;
    double *d_rec = nullptr, *d_scratch = nullptr;
    int *d_stat = nullptr;
    hipStream_t stream = nullptr;
    bool attr_set = false;
    LqpCfg lqp{};  // set by dwbc_batch_configure_lqp / _lqp_r
    bool is_lqp = false, lqp_reduced = false;
    // row weights V_ / W_ of HQP_Hierarch (src/dwbc_hqp.cpp:503-553), per level, B x m x m and B x e x e (empty: identity)
    std::vector<std::vector<double>> wV, wW;
};

namespace {
int hqp_free(dwbc_hqp *h) {
    if (h->d_rec) hipFree(h->d_rec);
    if (h->d_scratch) hipFree(h->d_scratch);
    if (h->d_stat) hipFree(h->d_stat);
    h->d_rec = h->d_scratch = nullptr;
    h->d_stat = nullptr;
    h->laid_out = false;
    return 1;
}
HqpIO hqp_io(const dwbc_hqp *h) { return HqpIO{h->B, h->d_rec, h->d_scratch, h->d_stat}; }
// copy a per-instance block (len doubles at offset off of every record) host -> device
int put_block(dwbc_hqp *h, int off, int len, const double *src) {
    if (len == 0 || !src) return 1;
    HIP_OK(hipMemcpy2D(h->d_rec + off, (size_t)h->d.rec * 8, src, (size_t)len * 8, (size_t)len * 8, h->B, hipMemcpyHostToDevice));
    return 1;
}
int layout_and_alloc(dwbc_hqp *h) {
    HIP_OK(hipSetDevice(h->device));
    hqp_free(h);
    if (h->d.n_levels < 1) return fail("HQP: no hierarchy");
    hqp_layout(h->d, h->share_cost);
    if (h->d.lds * 8 > 160 * 1024) return fail("HQP: the solver state of these sizes does not fit the 160 KB of LDS");
    HIP_OK(hipMalloc(&h->d_rec, (size_t)h->B * h->d.rec * 8));
    HIP_OK(hipMalloc(&h->d_scratch, (size_t)h->B * h->d.scratch * 8));
    HIP_OK(hipMalloc(&h->d_stat, (size_t)h->B * HQS_COUNT * sizeof(int)));
    HIP_OK(hipMemset(h->d_rec, 0, (size_t)h->B * h->d.rec * 8));
    HIP_OK(hipMemset(h->d_stat, 0, (size_t)h->B * HQS_COUNT * sizeof(int)));
    h->laid_out = true;
    h->attr_set = false;
    return 1;
}
int launch_solve(dwbc_hqp *h, int solve_first) {
    if (!h->laid_out) return fail("HQP: call dwbc_hqp_prepare first");
    HIP_OK(hipSetDevice(h->device));
    if (!h->attr_set) {
        HIP_OK(hipFuncSetAttribute((const void *)dwbc_hqp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, h->d.lds * 8));
        h->attr_set = true;
    }
    HqpDesc d = h->d;
    d.solve_first = solve_first;
    hipLaunchKernelGGL(dwbc_hqp_kernel, dim3(h->B), dim3(kNT), (size_t)h->d.lds * 8, h->stream, d, hqp_io(h));
    HIP_OK(hipGetLastError());
    return 1;
}
}  // namespace

extern "C" {

dwbc_hqp *dwbc_hqp_create(int B, int device, int acceleration_size, int torque_size, int contact_size) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { capi_err() = "no HIP device: libdwbc_hip has no CPU path"; return nullptr; }
    if (B < 1 || device < 0 || device >= ndev) { capi_err() = "bad arguments"; return nullptr; }
    const int nv = acceleration_size + torque_size + contact_size;
    if (nv < 1 || nv > kHqpMaxVar) { capi_err() = "HQP: variable size must be 1.." + std::to_string(kHqpMaxVar); return nullptr; }
    auto *h = new dwbc_hqp();
    h->B = B; h->device = device;
    h->acc = acceleration_size; h->torque = torque_size; h->contact = contact_size;
    h->d = HqpDesc{};
    h->d.nv = nv;
    h->d.max_iter = 400;
    h->d.eps = 1.0e-6;   // Tikhonov weight of the canon (oracle/hqp_np.py HQP_EPS)
    h->d.tol = 1.0e-6;   // HQP_TOL
    return h;
}
void dwbc_hqp_destroy(dwbc_hqp *h) {
    if (!h) return;
    hipSetDevice(h->device);
    hqp_free(h);
    delete h;
}
int dwbc_hqp_clear(dwbc_hqp *h) {
    hipSetDevice(h->device);
    hqp_free(h);
    h->d.n_levels = 0;
    h->stage.clear();
    h->wV.clear();
    h->wW.clear();
    h->is_lqp = false;
    h->share_cost = false;
    return 1;
}
int dwbc_hqp_num_levels(const dwbc_hqp *h) { return h->d.n_levels; }

int dwbc_hqp_add_hierarchy(dwbc_hqp *h, int ineq, int eq) {
    if (h->laid_out) { fail("HQP: hierarchy is fixed after dwbc_hqp_prepare (dwbc_hqp_clear starts over)"); return -1; }
    if (h->d.n_levels >= kHqpMaxLevels) { fail("HQP: too many levels"); return -1; }
    if (ineq < 0 || eq < 0 || eq > kHqpMaxEq) { fail("HQP: bad constraint sizes (equalities per level <= " + std::to_string(kHqpMaxEq) + ")"); return -1; }
    const int lv = h->d.n_levels++;
    h->d.m[lv] = ineq; h->d.e[lv] = eq; h->d.has_cost[lv] = 0; h->d.exact[lv] = 0;
    h->stage.emplace_back();
    return lv;
}

int dwbc_hqp_update_constraint_matrix(dwbc_hqp *h, int level, const double *A, const double *a, const double *Bm, const double *b) {
    if (level < 0 || level >= h->d.n_levels) return fail("HQP: bad level");
    const int nv = h->d.nv, m = h->d.m[level], e = h->d.e[level];
    if (m > 0 && (!A || !a)) return fail("HQP: inequality matrices missing");
    if (e > 0 && (!Bm || !b)) return fail("HQP: equality matrices missing");
    if (!h->laid_out) {
        auto &st = h->stage[level];
        if (m > 0) { st.A.assign(A, A + (size_t)h->B * m * nv); st.a.assign(a, a + (size_t)h->B * m); }
        if (e > 0) { st.Bm.assign(Bm, Bm + (size_t)h->B * e * nv); st.b.assign(b, b + (size_t)h->B * e); }
        return 1;
    }
    HIP_OK(hipSetDevice(h->device));
    return put_block(h, h->d.oA[level], m * nv, A) && put_block(h, h->d.oa[level], m, a) && put_block(h, h->d.oB[level], e * nv, Bm) &&
           put_block(h, h->d.ob[level], e, b);
}

int dwbc_hqp_update_cost_matrix(dwbc_hqp *h, int level, const double *H, const double *g) {
    (void)g;  // stored by the reference (updateCostMatrix) and never read by solveSequentialSingle
    if (level < 0 || level >= h->d.n_levels) return fail("HQP: bad level");
    if (!H) return fail("HQP: cost matrix missing");
    const int nv = h->d.nv;
    if (!h->laid_out) {
        h->d.has_cost[level] = 1;
        h->stage[level].H.assign(H, H + (size_t)h->B * nv * nv);
        return 1;
    }
    if (!h->d.has_cost[level]) return fail("HQP: this level was prepared without a cost (set it before dwbc_hqp_prepare)");
    HIP_OK(hipSetDevice(h->device));
    return put_block(h, h->d.oH[level], nv * nv, H);
}

int dwbc_hqp_normalize_constraint_matrix(dwbc_hqp *h, int level) {
    if (level < 0 || level >= h->d.n_levels) return fail("HQP: bad level");
    if (!h->laid_out) { h->stage[level].normalize = true; return 1; }
    HIP_OK(hipSetDevice(h->device));
    hipLaunchKernelGGL(dwbc_hqp_normalize_kernel, dim3(h->B), dim3(kNT), 0, h->stream, h->d, hqp_io(h), level);
    HIP_OK(hipGetLastError());
    return 1;
}

int dwbc_hqp_set_answer(dwbc_hqp *h, int level, const double *y_ans, const double *v_ans) {
    if (level < 0 || level >= h->d.n_levels) return fail("HQP: bad level");
    if (!h->laid_out) {
        auto &st = h->stage[level];
        if (y_ans) st.y.assign(y_ans, y_ans + (size_t)h->B * h->d.nv);
        if (v_ans) st.v.assign(v_ans, v_ans + (size_t)h->B * h->d.m[level]);
        return 1;
    }
    HIP_OK(hipSetDevice(h->device));
    return put_block(h, h->d.oy[level], h->d.nv, y_ans) && put_block(h, h->d.ov[level], h->d.m[level], v_ans);
}

int dwbc_hqp_prepare(dwbc_hqp *h) {
    if (h->laid_out) return 1;
    if (!layout_and_alloc(h)) return 0;
    const int nv = h->d.nv;
    for (int lv = 0; lv < h->d.n_levels; lv++) {
        auto &st = h->stage[lv];
        const int m = h->d.m[lv], e = h->d.e[lv];
        if (!st.A.empty() && !(put_block(h, h->d.oA[lv], m * nv, st.A.data()) && put_block(h, h->d.oa[lv], m, st.a.data()))) return 0;
        if (!st.Bm.empty() && !(put_block(h, h->d.oB[lv], e * nv, st.Bm.data()) && put_block(h, h->d.ob[lv], e, st.b.data()))) return 0;
        if (!st.H.empty() && !put_block(h, h->d.oH[lv], nv * nv, st.H.data())) return 0;
        if (!st.y.empty() && !put_block(h, h->d.oy[lv], nv, st.y.data())) return 0;
        if (!st.v.empty() && !put_block(h, h->d.ov[lv], m, st.v.data())) return 0;
        if (st.normalize && !dwbc_hqp_normalize_constraint_matrix(h, lv)) return 0;
        st = dwbc_hqp::Stage{};
    }
    return 1;
}

// HQP_Hierarch::updateInequalityCostWeight / updateEqualityCostWeight / updateConstraintWeight (src/dwbc_hqp.cpp:503-553).  The
// reference reads V_ and W_ in solvefirst only (:245-254: qp_A = V A, ubA = -V a, cost |W B y + W b|^2) and nowhere in
// solveSequentialSingle, so weights of levels beyond 0 are stored and never used, exactly as there.  V: B x m x m, W: B x e x e, row major,
// per instance; NULL = identity.
int dwbc_hqp_update_constraint_weight(dwbc_hqp *h, int level, const double *V, const double *W) {
    if (level < 0 || level >= h->d.n_levels) return fail("HQP: bad level");
    const size_t m = h->d.m[level], e = h->d.e[level];
    if ((int)h->wV.size() < h->d.n_levels) { h->wV.resize(h->d.n_levels); h->wW.resize(h->d.n_levels); }
    if (V) h->wV[level].assign(V, V + (size_t)h->B * m * m); else h->wV[level].clear();
    if (W) h->wW[level].assign(W, W + (size_t)h->B * e * e); else h->wW[level].clear();
    return 1;
}

namespace {
int get_block(dwbc_hqp *h, int off, int len, double *dst) {
    if (len == 0) return 1;
    HIP_OK(hipMemcpy2D(dst, (size_t)len * 8, h->d_rec + off, (size_t)h->d.rec * 8, (size_t)len * 8, h->B, hipMemcpyDeviceToHost));
    return 1;
}
// out (r x c per instance) = Wt (r x r per instance) * in (r x c per instance)
void weigh(const std::vector<double> &Wt, const std::vector<double> &in, std::vector<double> &out, int B, int r, int c) {
    out.assign(in.size(), 0.0);
    for (int i = 0; i < B; i++)
        for (int a = 0; a < r; a++)
            for (int k = 0; k < r; k++) {
                const double w = Wt[((size_t)i * r + a) * r + k];
                if (w == 0.0) continue;
                for (int j = 0; j < c; j++) out[((size_t)i * r + a) * c + j] += w * in[((size_t)i * r + k) * c + j];
            }
}
}  // namespace

int dwbc_hqp_solve_first(dwbc_hqp *h, int init) {
    (void)init;
    // level 0 alone over the full variable: run the cascade restricted to one level
    if (!h->laid_out && !dwbc_hqp_prepare(h)) return 0;
    const int nv = h->d.nv, m = h->d.m[0], e = h->d.e[0];
    const bool wv = !h->wV.empty() && !h->wV[0].empty() && m > 0, ww = !h->wW.empty() && !h->wW[0].empty() && e > 0;
    std::vector<double> A, a, Bm, b, t;
    if (wv || ww) {
        // solvefirst poses level 0 on (V A, V a, W B, W b); every later solve reads the unweighted matrices again (as the reference
        // does), so they are weighed for this launch only and put back: a host round trip, on a path the reference takes once per set-up
        HIP_OK(hipSetDevice(h->device));
        HIP_OK(hipStreamSynchronize(h->stream));
        if (wv) {
            A.resize((size_t)h->B * m * nv); a.resize((size_t)h->B * m);
            if (!get_block(h, h->d.oA[0], m * nv, A.data()) || !get_block(h, h->d.oa[0], m, a.data())) return 0;
            weigh(h->wV[0], A, t, h->B, m, nv);
            if (!put_block(h, h->d.oA[0], m * nv, t.data())) return 0;
            weigh(h->wV[0], a, t, h->B, m, 1);
            if (!put_block(h, h->d.oa[0], m, t.data())) return 0;
        }
        if (ww) {
            Bm.resize((size_t)h->B * e * nv); b.resize((size_t)h->B * e);
            if (!get_block(h, h->d.oB[0], e * nv, Bm.data()) || !get_block(h, h->d.ob[0], e, b.data())) return 0;
            weigh(h->wW[0], Bm, t, h->B, e, nv);
            if (!put_block(h, h->d.oB[0], e * nv, t.data())) return 0;
            weigh(h->wW[0], b, t, h->B, e, 1);
            if (!put_block(h, h->d.ob[0], e, t.data())) return 0;
        }
    }
    const int keep = h->d.n_levels;
    h->d.n_levels = 1;
    int ok = launch_solve(h, 1);
    h->d.n_levels = keep;
    if (wv || ww) {
        HIP_OK(hipStreamSynchronize(h->stream));
        if (wv) ok = ok && put_block(h, h->d.oA[0], m * nv, A.data()) && put_block(h, h->d.oa[0], m, a.data());
        if (ww) ok = ok && put_block(h, h->d.oB[0], e * nv, Bm.data()) && put_block(h, h->d.ob[0], e, b.data());
    }
    return ok;
}
int dwbc_hqp_solve_sequential(dwbc_hqp *h, int init) {
    (void)init;
    if (!h->laid_out && !dwbc_hqp_prepare(h)) return 0;
    return launch_solve(h, 0);
}

size_t dwbc_hqp_field_bytes(const dwbc_hqp *h, int level, int field) {
    if (level < 0 || level >= h->d.n_levels) return 0;
    const size_t nv = h->d.nv, m = h->d.m[level], e = h->d.e[level], B = h->B;
    switch (field) {
        case DWBC_HQP_Y_ANS: return B * nv * 8;
        case DWBC_HQP_V_ANS: case DWBC_HQP_a: return B * m * 8;
        case DWBC_HQP_W_ANS: case DWBC_HQP_b: return B * e * 8;
        case DWBC_HQP_STATUS: case DWBC_HQP_ITER: case DWBC_HQP_NULL_SIZE: return B * sizeof(int);
        case DWBC_HQP_A: return B * m * nv * 8;
        case DWBC_HQP_B: return B * e * nv * 8;
        default: return 0;
    }
}

int dwbc_hqp_get(dwbc_hqp *h, int level, int field, void *out, size_t bytes) {
    if (!h->laid_out) return fail("HQP: nothing prepared");
    if (level < 0 || level >= h->d.n_levels) return fail("HQP: bad level");
    if (bytes != dwbc_hqp_field_bytes(h, level, field)) return fail("HQP: size mismatch");
    HIP_OK(hipSetDevice(h->device));
    HIP_OK(hipStreamSynchronize(h->stream));
    if (bytes == 0) return 1;
    int off = -1, len = 0;
    const int nv = h->d.nv, m = h->d.m[level], e = h->d.e[level];
    switch (field) {
        case DWBC_HQP_Y_ANS: off = h->d.oy[level]; len = nv; break;
        case DWBC_HQP_V_ANS: off = h->d.ov[level]; len = m; break;
        case DWBC_HQP_W_ANS: off = h->d.ow[level]; len = e; break;
        case DWBC_HQP_A: off = h->d.oA[level]; len = m * nv; break;
        case DWBC_HQP_a: off = h->d.oa[level]; len = m; break;
        case DWBC_HQP_B: off = h->d.oB[level]; len = e * nv; break;
        case DWBC_HQP_b: off = h->d.ob[level]; len = e; break;
        case DWBC_HQP_STATUS: case DWBC_HQP_ITER: case DWBC_HQP_NULL_SIZE: {
            const int so = (field == DWBC_HQP_STATUS ? HQS_STATUS : field == DWBC_HQP_ITER ? HQS_ITER : HQS_NULL) + level;
            HIP_OK(hipMemcpy2D(out, sizeof(int), h->d_stat + so, HQS_COUNT * sizeof(int), sizeof(int), h->B, hipMemcpyDeviceToHost));
            return 1;
        }
        default: return fail("HQP: unknown field");
    }
    HIP_OK(hipMemcpy2D(out, (size_t)len * 8, h->d_rec + off, (size_t)h->d.rec * 8, (size_t)len * 8, h->B, hipMemcpyDeviceToHost));
    return 1;
}

// what both formulations read from a solved cycle: checks + the contact / task description of the (uniform) batch
// what both formulations read from the batch.  reduced: the cycle that ran was the reduced one; the system handed to the
// configurators is then the record dwbc_reduced_record_kernel writes (see dwbc_hqp.h) and cfg describes THAT system.
struct ReducedInfo { int vcd = 0, RS = 0; unsigned long long comask = 0x3full; int kind[kMaxLevels] = {0, 0, 0, 0}; };

static int formulation_cfg(dwbc_batch *b, dwbc_hqp *h, LqpCfg &cfg, bool reduced, ReducedInfo *ri = nullptr) {
    if (!b || !h) return fail("NULL handle");
    if (b->dtype != DWBC_F64) return fail("LQP / JACC: fp64 batches only");
    if (h->B != b->B || h->device != b->device) return fail("LQP / JACC: the HQP object must have the batch's size and device");
    if (!b->dump_on || !b->d_dump) return fail("LQP / JACC: needs dwbc_batch_enable_dump(b, 1) and a solved cycle (A_, A_inv_, J_C, B_, J_task come from it)");
    if (reduced && !b->last_reduced) return fail("LQP_R / JACC_R: run the reduced cycle first (DWBC_SOLVE_REDUCED; A_R, G_R, J_I_nc_inv_T come from it)");
    if (!reduced && b->last_reduced) return fail("LQP / JACC: run the full-model cycle first (or use the _r entry points after a reduced one)");
    if (b->su.n_custom > 0 || b->su.has_com_task) return fail("LQP / JACC: link task levels only");
    if (b->h_flags.empty() || (b->d_flags && !b->own_flags)) return fail("LQP / JACC: contact flags must be set through dwbc_batch_set_contact (the host checks that they are uniform)");
    const int ncn = b->su.n_contacts;
    cfg = LqpCfg{};
    cfg.n = b->n;
    cfg.nc = 0;
    unsigned long long comask = 0x3full;
    for (int c = 0; c < ncn; c++)
        if (b->h_flags[c]) {
            if (cfg.nc >= kMaxActiveContacts) return fail("LQP / JACC: more than 2 active contacts");
            cfg.act[cfg.nc] = c;
            cfg.lx[cfg.nc] = b->su.c_lx[c]; cfg.ly[cfg.nc] = b->su.c_ly[c]; cfg.mu[cfg.nc] = b->su.c_mu[c]; cfg.muz[cfg.nc] = b->su.c_muz[c];
            comask |= b->su.c_dofmask[c];
            cfg.nc++;
        }
    for (int i = 1; i < b->B; i++)
        if (memcmp(&b->h_flags[(size_t)i * ncn], &b->h_flags[0], ncn) != 0)
            return fail("LQP / JACC: every instance of the batch must be in the same contact state (the level sizes depend on it)");
    if (cfg.nc < 1) return fail("LQP / JACC: no active contact");
    cfg.cd = 6 * cfg.nc;
    cfg.fstar_total = b->su.fstar_total;
    cfg.tlim = 200.0;  // `tlim`, src/dwbc.cpp:4360
    cfg.alim = 5.0;    // `alim`, src/dwbc.cpp:4398
    cfg.oNorm = -1;
    cfg.tlim_idx = -1;
    cfg.tlim_special = 0.0;
    if (!reduced) {
        cfg.n_tasks = b->su.n_levels;
        for (int i = 0; i < b->su.n_levels; i++) { cfg.t_dof[i] = b->su.t_dof[i]; cfg.fstar_off[i] = b->su.fstar_off[i]; }
        cfg.oBn = b->d_qdot ? b->dl.B : b->dl.G;  // B_(q, qdot = 0) = G_
        cfg.jacc_mt = b->n - 6;
        return 1;
    }
    // ---- the reduced system: contact chains (leading dofs) + 6 centroidal coordinates (dwbc.cpp:2818-2823)
    int vcd = 0;
    for (int j = 0; j < b->n; j++) vcd += (int)((comask >> j) & 1ull);
    if (comask != ((1ull << vcd) - 1ull) || (vcd != 12 && vcd != 18)) return fail("LQP_R / JACC_R: the contact chains must occupy the leading joint dofs (TOCABI: L+R or L)");
    const int RS = vcd + 6;
    ReducedInfo info;
    info.vcd = vcd; info.RS = RS; info.comask = comask;
    ReducedRecCfg rc{};
    rc.n = b->n; rc.vcd = vcd; rc.cd = cfg.cd; rc.n_src = 0;
    for (int lv = 0; lv < b->su.n_levels; lv++) {
        int nco = 0, nnc = 0;
        for (int li = 0; li < b->su.t_nlinks[lv]; li++) {
            const int link = b->su.t_link[lv][li];
            if (link == 0 || ((comask >> (link + 5)) & 1ull)) nco++; else nnc++;
        }
        if (nco && nnc) return fail("LQP_R / JACC_R: a task level mixes contact-chain and other links (undefined in the reference, src/task.cpp:134-141)");
        info.kind[lv] = nco ? 1 : 2;
        if (nco) {  // `!ts_[i].noncont_task` (dwbc.cpp:4610): packed in level order
            rc.src[rc.n_src] = lv;
            rc.t_dof[rc.n_src] = b->su.t_dof[lv];
            cfg.t_dof[rc.n_src] = b->su.t_dof[lv];
            cfg.fstar_off[rc.n_src] = b->su.fstar_off[lv];
            rc.n_src++;
        }
    }
    cfg.n_tasks = rc.n_src;
    cfg.n = RS;
    const DumpLayout dr = DumpLayout::make(RS);
    cfg.oBn = dr.G;
    cfg.oNorm = dr.com;             // |A_|_F of the full model (dwbc.cpp:4533)
    cfg.tlim_idx = (RS - 6) - 4;    // `tlim(tlim_size - 4) = 600` (dwbc.cpp:4552)
    cfg.tlim_special = 600.0;
    cfg.jacc_mt = RS - 12;          // `_torque_dof - 6` rows carry the +-200 bound (dwbc.cpp:4096-4097)
    HIP_OK(hipSetDevice(b->device));
    if (b->rrec_n != RS || !b->d_rrec) {
        if (b->d_rrec) hipFree(b->d_rrec);
        b->d_rrec = nullptr;
        HIP_OK(hipMalloc(&b->d_rrec, (size_t)b->B * dr.total * 8));
        HIP_OK(hipMemset(b->d_rrec, 0, (size_t)b->B * dr.total * 8));
        b->rrec_n = RS;
    }
    hipLaunchKernelGGL(dwbc_reduced_record_kernel, dim3(b->B), dim3(kNT), 0, b->stream, rc, b->B, (const double *)b->d_dump, b->d_rrec);
    HIP_OK(hipGetLastError());
    if (ri) *ri = info;
    return 1;
}

static int configure_lqp_common(dwbc_batch *b, dwbc_hqp *h, bool reduced) {
    LqpCfg cfg{};
    if (!formulation_cfg(b, h, cfg, reduced)) return 0;
    const int n = cfg.n, m = n - 6;
    if (h->acc != n || h->torque != 0 || h->contact != cfg.cd)
        return fail("LQP: create the HQP object with (acceleration, torque, contact) = (" + std::to_string(n) + ", 0, " + std::to_string(cfg.cd) + ")");
    if (2 + cfg.n_tasks > kHqpMaxLevels) return fail("LQP: too many task levels");
    // (re)build the hierarchy if its shape changed
    bool same = h->is_lqp && h->laid_out && h->d.n_levels == 2 + cfg.n_tasks && h->lqp.cd == cfg.cd && h->lqp.n == n;
    for (int i = 0; same && i < cfg.n_tasks; i++) same = h->d.e[2 + i] == cfg.t_dof[i];
    if (!same) {
        dwbc_hqp_clear(h);
        h->share_cost = true;
        if (dwbc_hqp_add_hierarchy(h, 2 * m, 6) < 0) return 0;
        if (dwbc_hqp_add_hierarchy(h, 10 * cfg.nc + 2 * m, cfg.cd) < 0) return 0;
        h->d.has_cost[1] = 1;
        for (int i = 0; i < cfg.n_tasks; i++) {
            if (dwbc_hqp_add_hierarchy(h, 0, cfg.t_dof[i]) < 0) return 0;
            h->d.has_cost[2 + i] = 1;
        }
        if (!layout_and_alloc(h)) return 0;
        h->stage.assign(h->d.n_levels, dwbc_hqp::Stage{});
        h->is_lqp = true;
    }
    h->lqp = cfg;
    h->lqp_reduced = reduced;
    h->stream = b->stream;
    HIP_OK(hipSetDevice(b->device));
    hipLaunchKernelGGL(dwbc_lqp_configure_kernel, dim3(b->B), dim3(kNT), 0, b->stream, cfg, h->d, hqp_io(h), (const double *)(reduced ? b->d_rrec : b->d_dump),
                       (const double *)b->d_fstar);
    HIP_OK(hipGetLastError());
    return 1;
}

int dwbc_batch_reduced_dims(dwbc_batch *b, int *vc_dof, int *nc_dof) {
    if (!b || !vc_dof || !nc_dof) return fail("NULL argument");
    if (b->h_flags.empty()) return fail("reduced dims: set the contact flags first");
    unsigned long long comask = 0x3full;
    for (int c = 0; c < b->su.n_contacts; c++)
        if (b->h_flags[c]) comask |= b->su.c_dofmask[c];
    int vcd = 0;
    for (int j = 0; j < b->n; j++) vcd += (int)((comask >> j) & 1ull);
    if (comask != ((1ull << vcd) - 1ull) || (vcd != 12 && vcd != 18)) return fail("reduced dims: the contact chains must occupy the leading joint dofs");
    *vc_dof = vcd;
    *nc_dof = b->n - vcd;
    return 1;
}

int dwbc_batch_configure_lqp(dwbc_batch *b, dwbc_hqp *h) { return configure_lqp_common(b, h, false); }
int dwbc_batch_configure_lqp_r(dwbc_batch *b, dwbc_hqp *h) { return configure_lqp_common(b, h, true); }

int dwbc_batch_lqp_torque(dwbc_batch *b, dwbc_hqp *h, double *tau) {
    if (!b || !h || !tau) return fail("NULL argument");
    if (!h->is_lqp || !h->laid_out) return fail("LQP: dwbc_batch_configure_lqp first");
    if (h->lqp_reduced && (!b->d_rrec || b->rrec_n != h->lqp.n)) return fail("LQP_R: the reduced record of this batch is gone");
    HIP_OK(hipSetDevice(b->device));
    const size_t m = (size_t)h->lqp.n - 6;
    double *d_tau = nullptr;
    HIP_OK(hipMalloc(&d_tau, (size_t)b->B * m * 8));
    hipLaunchKernelGGL(dwbc_lqp_torque_kernel, dim3(b->B), dim3(kNT), 0, b->stream, h->lqp, h->d, hqp_io(h), (const double *)(h->lqp_reduced ? b->d_rrec : b->d_dump), d_tau);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    if (e == hipSuccess) e = hipMemcpy(tau, d_tau, (size_t)b->B * m * 8, hipMemcpyDeviceToHost);
    hipFree(d_tau);
    if (e != hipSuccess) return fail(std::string("lqp torque: ") + hipGetErrorString(e));
    return 1;
}

static int solve_jacc_common(dwbc_batch *b, dwbc_hqp *h, int level, bool reduced) {
    LqpCfg cfg{};
    if (!formulation_cfg(b, h, cfg, reduced)) return 0;
    const int n = cfg.n, m = n - 6;
    if (level < 0 || level >= cfg.n_tasks) return fail("JACC: bad task level");
    if (h->acc != n || h->torque != 0 || h->contact != cfg.cd)
        return fail("JACC: create the HQP object with (acceleration, torque, contact) = (" + std::to_string(n) + ", 0, " + std::to_string(cfg.cd) + ")");
    int e0 = 6 + cfg.cd;
    for (int i = 0; i < level; i++) {
        if (!b->d_jacc[i] || b->jacc_n[i] != n) return fail("JACC: solve the levels in order (level " + std::to_string(i) + " has no result yet)");
        e0 += cfg.t_dof[i];
    }
    if (e0 > kHqpMaxEq) return fail("JACC: too many equality rows");
    // two levels: the exact constraint level and the task level (the buffers are kept while the shape stays the same)
    const int m0 = 10 * cfg.nc + 2 * m + 2 * cfg.jacc_mt;
    const bool same = h->laid_out && !h->is_lqp && h->d.n_levels == 2 && h->d.nv == n + cfg.cd && h->d.m[0] == m0 && h->d.e[0] == e0 && h->d.m[1] == 0 &&
                      h->d.e[1] == cfg.t_dof[level] && h->d.exact[0] == 1 && h->d.exact[1] == 0 && h->d.has_cost[0] == 0 && h->d.has_cost[1] == 1;
    if (!same) {
        dwbc_hqp_clear(h);
        if (dwbc_hqp_add_hierarchy(h, m0, e0) < 0) return 0;
        if (dwbc_hqp_add_hierarchy(h, 0, cfg.t_dof[level]) < 0) return 0;
        h->d.exact[0] = 1;
        h->d.exact[1] = 0;
        h->d.has_cost[1] = 1;
        if (!layout_and_alloc(h)) return 0;
        h->stage.assign(h->d.n_levels, dwbc_hqp::Stage{});
    }
    h->stream = b->stream;
    HIP_OK(hipSetDevice(b->device));
    const size_t rs = (size_t)jacc_rec_size(n);
    if (b->d_jacc[level] && b->jacc_n[level] != n) { hipFree(b->d_jacc[level]); b->d_jacc[level] = nullptr; }
    if (!b->d_jacc[level]) HIP_OK(hipMalloc(&b->d_jacc[level], (size_t)b->B * rs * 8));
    b->jacc_n[level] = n;
    if (!b->d_jacc_status) {
        HIP_OK(hipMalloc(&b->d_jacc_status, (size_t)kMaxLevels * b->B * sizeof(int)));
        HIP_OK(hipMemset(b->d_jacc_status, 0, (size_t)kMaxLevels * b->B * sizeof(int)));
    }
    JaccPrev prev{};
    for (int i = 0; i < level; i++) prev.rec[i] = b->d_jacc[i];
    const double *sys = reduced ? b->d_rrec : b->d_dump;
    hipLaunchKernelGGL(dwbc_jacc_configure_kernel, dim3(b->B), dim3(kNT), 0, b->stream, cfg, level, prev, h->d, hqp_io(h), sys, (const double *)b->d_fstar);
    HIP_OK(hipGetLastError());
    if (!launch_solve(h, 0)) return 0;
    hipLaunchKernelGGL(dwbc_jacc_extract_kernel, dim3(b->B), dim3(kNT), 0, b->stream, cfg, level, h->d, hqp_io(h), sys, (const double *)b->d_fstar,
                       b->d_jacc[level], b->d_jacc_status + (size_t)level * b->B);
    HIP_OK(hipGetLastError());
    return 1;
}

int dwbc_batch_solve_jacc(dwbc_batch *b, dwbc_hqp *h, int level) { return solve_jacc_common(b, h, level, false); }
int dwbc_batch_solve_jacc_r(dwbc_batch *b, dwbc_hqp *h, int level) { return solve_jacc_common(b, h, level, true); }

// the non-contact halves: one 6-D level on a non-contact link; `prev` = the reduced answer it continues
static int nc_cfg(dwbc_batch *b, dwbc_hqp *h, int level, NcCfg &c, ReducedInfo &ri) {
    LqpCfg tmp{};
    if (!formulation_cfg(b, h, tmp, true, &ri)) return 0;
    if (level < 0 || level >= b->su.n_levels) return fail("R_NC: bad task level");
    if (ri.kind[level] != 2 || b->su.t_nlinks[level] != 1 || b->su.t_dof[level] != 6)
        return fail("R_NC: the level must be ONE 6-D task on a non-contact link (the reference's ConfigureLQP_R_NC / JACC_QP_R_NC read ts_[1] as such, src/dwbc.cpp:4144,4728)");
    const int ncd = b->n - ri.vcd;
    if (h->acc != ncd || h->torque != 0 || h->contact != 0)
        return fail("R_NC: create the HQP object with (acceleration, torque, contact) = (" + std::to_string(ncd) + ", 0, 0)");
    c = NcCfg{};
    c.n = b->n; c.vcd = ri.vcd; c.level = level; c.link = b->su.t_link[level][0]; c.t = 6;
    c.fstar_off = b->su.fstar_off[level]; c.fstar_total = b->su.fstar_total;
    return 1;
}

int dwbc_batch_configure_lqp_r_nc(dwbc_batch *b, dwbc_hqp *h, const dwbc_hqp *hr, int level) {
    if (!hr || !hr->is_lqp || !hr->lqp_reduced || !hr->laid_out) return fail("LQP_R_NC: needs the solved reduced LQP (dwbc_batch_configure_lqp_r + solve)");
    if (hr == h) return fail("LQP_R_NC: use a second HQP object");
    NcCfg c{};
    ReducedInfo ri;
    if (!nc_cfg(b, h, level, c, ri)) return 0;
    if (hr->lqp.n != ri.RS || hr->B != b->B) return fail("LQP_R_NC: the reduced LQP belongs to another contact state or batch");
    c.prev_stride = hr->d.rec;
    c.prev_off = hr->d.oy[hr->d.n_levels - 1];
    const int ncd = b->n - ri.vcd;
    dwbc_hqp_clear(h);
    h->share_cost = true;
    if (dwbc_hqp_add_hierarchy(h, 2 * ncd, 6) < 0) return 0;
    if (dwbc_hqp_add_hierarchy(h, 2 * ncd, 6) < 0) return 0;
    h->d.has_cost[0] = h->d.has_cost[1] = 1;
    if (!layout_and_alloc(h)) return 0;
    h->stage.assign(h->d.n_levels, dwbc_hqp::Stage{});
    h->stream = b->stream;
    HIP_OK(hipSetDevice(b->device));
    hipLaunchKernelGGL(dwbc_lqp_nc_configure_kernel, dim3(b->B), dim3(kNT), 0, b->stream, c, h->d, hqp_io(h), (const double *)b->d_dump, (const double *)b->d_fstar,
                       (const double *)hr->d_rec);
    HIP_OK(hipGetLastError());
    return 1;
}

int dwbc_batch_solve_jacc_r_nc(dwbc_batch *b, dwbc_hqp *h, int level, int src_level) {
    NcCfg c{};
    ReducedInfo ri;
    if (!nc_cfg(b, h, level, c, ri)) return 0;
    if (src_level < 0 || src_level >= kMaxLevels || !b->d_jacc[src_level] || b->jacc_n[src_level] != ri.RS)
        return fail("JACC_R_NC: no reduced JACC result for the source level (dwbc_batch_solve_jacc_r first)");
    c.prev_stride = jacc_rec_size(ri.RS);
    c.prev_off = 0;
    dwbc_hqp_clear(h);
    if (dwbc_hqp_add_hierarchy(h, 0, 6 + c.t) < 0) return 0;
    if (!layout_and_alloc(h)) return 0;
    h->stage.assign(h->d.n_levels, dwbc_hqp::Stage{});
    h->stream = b->stream;
    HIP_OK(hipSetDevice(b->device));
    if (!b->d_jacc_nc) {
        HIP_OK(hipMalloc(&b->d_jacc_nc, (size_t)b->B * jacc_nc_rec_size(b->n - 12) * 8));
        HIP_OK(hipMalloc(&b->d_jacc_nc_status, (size_t)b->B * sizeof(int)));
        HIP_OK(hipMemset(b->d_jacc_nc_status, 0, (size_t)b->B * sizeof(int)));
    }
    const double *prev = b->d_jacc[src_level];
    hipLaunchKernelGGL(dwbc_jacc_nc_configure_kernel, dim3(b->B), dim3(kNT), 0, b->stream, c, h->d, hqp_io(h), (const double *)b->d_dump, (const double *)b->d_fstar, prev);
    HIP_OK(hipGetLastError());
    if (!launch_solve(h, 1)) return 0;
    hipLaunchKernelGGL(dwbc_jacc_nc_extract_kernel, dim3(b->B), dim3(kNT), 0, b->stream, c, h->d, hqp_io(h), (const double *)b->d_dump, (const double *)b->d_fstar, prev,
                       b->d_jacc_nc, b->d_jacc_nc_status);
    HIP_OK(hipGetLastError());
    return 1;
}

int dwbc_batch_get_jacc_nc(dwbc_batch *b, int field, void *out, size_t bytes) {
    if (!b || !b->d_jacc_nc) return fail("JACC_R_NC: no result");
    // nc_dof of the contact state the result was computed in
    unsigned long long comask = 0x3full;
    for (int c = 0; c < b->su.n_contacts; c++)
        if (b->h_flags[c]) comask |= b->su.c_dofmask[c];
    int vcd = 0;
    for (int j = 0; j < b->n; j++) vcd += (int)((comask >> j) & 1ull);
    const int ncd = b->n - vcd;
    const size_t rs = (size_t)jacc_nc_rec_size(ncd);
    int off = 0, len = 0;
    switch (field) {
        case DWBC_JACC_ACC: off = 0; len = ncd; break;
        case DWBC_JACC_TORQUE: off = ncd; len = ncd; break;
        case DWBC_JACC_CONTACT: off = 2 * ncd; len = 6; break;  // gacc_qp_
        case DWBC_JACC_FSTAR_QP: off = 2 * ncd + 6; len = kMaxTaskDof; break;
        case DWBC_JACC_STATUS:
            if (bytes != (size_t)b->B * sizeof(int)) return fail("JACC_R_NC: size mismatch");
            HIP_OK(hipSetDevice(b->device));
            HIP_OK(hipStreamSynchronize(b->stream));
            HIP_OK(hipMemcpy(out, b->d_jacc_nc_status, bytes, hipMemcpyDeviceToHost));
            return 1;
        default: return fail("JACC_R_NC: unknown field");
    }
    if (bytes != (size_t)b->B * len * 8) return fail("JACC_R_NC: size mismatch");
    HIP_OK(hipSetDevice(b->device));
    HIP_OK(hipStreamSynchronize(b->stream));
    HIP_OK(hipMemcpy2D(out, (size_t)len * 8, b->d_jacc_nc + off, rs * 8, (size_t)len * 8, b->B, hipMemcpyDeviceToHost));
    return 1;
}

int dwbc_batch_get_jacc(dwbc_batch *b, int level, int field, void *out, size_t bytes) {
    if (level < 0 || level >= kMaxLevels || !b->d_jacc[level]) return fail("JACC: no result for this level");
    const int n = b->jacc_n[level], m = n - 6;  // n of the system the level was solved on (RS after dwbc_batch_solve_jacc_r)
    const size_t rs = (size_t)jacc_rec_size(n);
    int off = 0, len = 0;
    switch (field) {
        case DWBC_JACC_ACC: off = 0; len = n; break;
        case DWBC_JACC_TORQUE: off = n; len = m; break;
        case DWBC_JACC_CONTACT: off = n + m; len = 12; break;
        case DWBC_JACC_FSTAR_QP: off = n + m + 12; len = kMaxTaskDof; break;
        case DWBC_JACC_STATUS:
            if (bytes != (size_t)b->B * sizeof(int)) return fail("JACC: size mismatch");
            HIP_OK(hipSetDevice(b->device));
            HIP_OK(hipStreamSynchronize(b->stream));
            HIP_OK(hipMemcpy(out, b->d_jacc_status + (size_t)level * b->B, bytes, hipMemcpyDeviceToHost));
            return 1;
        default: return fail("JACC: unknown field");
    }
    if (bytes != (size_t)b->B * len * 8) return fail("JACC: size mismatch");
    HIP_OK(hipSetDevice(b->device));
    HIP_OK(hipStreamSynchronize(b->stream));
    HIP_OK(hipMemcpy2D(out, (size_t)len * 8, b->d_jacc[level] + off, rs * 8, (size_t)len * 8, b->B, hipMemcpyDeviceToHost));
    return 1;
}

}  // extern "C"

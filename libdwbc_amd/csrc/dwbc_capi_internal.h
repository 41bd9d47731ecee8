// dwbc_capi_internal.h -- the opaque handles of include/dwbc_batch.h as the translation units of libdwbc_hip.so see them
// (dwbc_capi.hip: cycle kernels + batch API; dwbc_hqp_capi.hip: hierarchical-QP class + LQP configurator).
#pragma once
#include <hip/hip_runtime.h>

#include <new>
#include <string>
#include <vector>

#include "../../include/dwbc_batch.h"
#include "dwbc_model.h"
#include "dwbc_types.h"

namespace dwbc {
// host mirrors of the per-cycle inputs and the read-back staging live in page-locked memory: hipMemcpyAsync on the batch's stream
// then really is asynchronous and runs at PCIe rate (pageable memory goes through the runtime's own staging, synchronously)
template <class T>
struct PinnedAlloc {
    using value_type = T;
    PinnedAlloc() = default;
    template <class U>
    PinnedAlloc(const PinnedAlloc<U> &) {}
    T *allocate(size_t n) {
        void *p = nullptr;
        if (hipHostMalloc(&p, n * sizeof(T), hipHostMallocDefault) != hipSuccess) throw std::bad_alloc();
        return static_cast<T *>(p);
    }
    void deallocate(T *p, size_t) { (void)hipHostFree(p); }
    template <class U>
    bool operator==(const PinnedAlloc<U> &) const { return true; }
    template <class U>
    bool operator!=(const PinnedAlloc<U> &) const { return false; }
};
template <class T>
using PinnedVec = std::vector<T, PinnedAlloc<T>>;
struct KernelEntry;
std::string &capi_err();  // thread-local last error (dwbc_last_error)
inline int capi_fail(const std::string &s) {
    capi_err() = s;
    return 0;
}
}  // namespace dwbc
#define HIP_OK(expr)                                                                                           \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return dwbc::capi_fail(std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

struct dwbc_model {
    dwbc::Model m;
};

struct dwbc_batch {
    const dwbc_model *model = nullptr;
    int B = 0, device = 0, n = 0, m = 0;
    dwbc::Setup su{};
    const dwbc::KernelEntry *kern = nullptr;
    hipStream_t stream = nullptr;
    // device buffers (owned unless bound)
    double *d_qdot = nullptr;  // B x n, allocated when the caller passes a qdot
    double *d_traj = nullptr, *d_ctime = nullptr;  // on-device task reference inputs (dwbc_fstar.h)
    double *d_custom = nullptr;  // B x n_custom x 6 x n: J_task of the TASK_CUSTOM levels
    std::vector<double> h_custom;
    bool dirty_custom = false;
    std::vector<double> h_traj, h_ctime;
    bool dirty_traj = false, dirty_ctime = false;
    dwbc::PinnedVec<double> h_qdot;
    bool dirty_qdot = false;
    double *d_q = nullptr, *d_fstar = nullptr, *d_tau = nullptr, *d_wrench = nullptr, *d_dump = nullptr, *d_body = nullptr;
    unsigned char *d_flags = nullptr;
    int *d_status = nullptr, *d_diag = nullptr, *d_topo = nullptr;
    bool own_q = false, own_fstar = false, own_flags = false, own_tau = false, own_wrench = false, own_status = false;
    int fstar_alloc = 0, flags_alloc = 0;
    bool dump_on = false;
    int dtype = 0;  // DWBC_F64 | DWBC_F32 (arithmetic type of the kernels; the boundary buffers are always double)
    float *f_body = nullptr;
    int max_active = 2;           // simultaneously active contacts per instance the batch solves (2: product kernels; 3: dwbc_cycle_gc.h)
    int gc_attr_set = 0;  // 1: the general-contact kernel's LDS attribute is set, 2: its wide-task instantiation's
    const void *f32_fn = nullptr, *f32_fn_wide = nullptr;
    int f32_lds = 0, f32_lds_wide = 0, f32_key = -1, f32_topo = 0;
    int hqp = 1;
    int warm = 0;           // last solve flags had DWBC_SOLVE_INIT clear
    bool ws_valid = false;  // diag holds the working sets of a full-build launch
    bool last_reduced = false;  // mode of the most recent dwbc_batch_solve (kernel_name / launch_info report it)
    // host mirrors of the inputs
    dwbc::PinnedVec<double> h_q, h_fstar;
    dwbc::PinnedVec<unsigned char> h_flags;
    dwbc::PinnedVec<unsigned char> h_stage;  // read-back staging of dwbc_batch_get
    double *d_total = nullptr;                // B x m scratch of the DWBC_TAU_* getters
    double *d_jacc[dwbc::kMaxLevels] = {nullptr, nullptr, nullptr, nullptr};  // per task level: B x jacc_rec_size (dwbc_batch_solve_jacc)
    int *d_jacc_status = nullptr;             // kMaxLevels x B
    int jacc_n[dwbc::kMaxLevels] = {0, 0, 0, 0};  // system size each record was written with (n, or RS for dwbc_batch_solve_jacc_r)
    double *d_rrec = nullptr;                 // B x DumpLayout::make(RS).total: the reduced system in dump-record layout (dwbc_hqp.h)
    int rrec_n = 0;
    double *d_jacc_nc = nullptr;              // B x jacc_nc_rec_size (dwbc_batch_solve_jacc_r_nc)
    int *d_jacc_nc_status = nullptr;
    bool dirty_q = false, dirty_fstar = false, dirty_flags = false;
    // the mirrors are page-locked, so an upload returns before the DMA engine has read them: this event is recorded behind the
    // host-to-device copies of a solve and waited for before anything rewrites a mirror (dwbc_batch_set_*, dwbc_batch_host_ptr)
    hipEvent_t ev_upload = nullptr;
    bool upload_pending = false;
    bool attr_set = false;
    int n_cu = 0;
    dwbc::DumpLayout dl{};
};


// dwbc_kernels_f32.hip -- the fp32 build of the fused cycle kernels (BASELINE config 5 names an fp32 path).  Same source as the
// fp64 product kernels (dwbc_kernels.h) with DWBC_REAL = float; the namespace is renamed so that both builds link into
// libdwbc_hip.so.  dwbc_capi.hip looks the entry points up through dwbc_f32_lookup() and launches them with hipLaunchKernel.
#define DWBC_REAL float
#define DWBC_NO_PAIR_KERNEL
#define dwbc dwbc_f32
#include "dwbc_kernels.h"
#undef dwbc

extern "C" int dwbc_f32_lookup(int n, int nb, int nlv, int which, int lean, int topo, const void **fn, const void **fn_wide, int *lds_bytes, int *lds_bytes_wide, int *topo_out) {
    const dwbc_f32::KernelEntry *k = dwbc_f32::lookup_kernel(n, nb, nlv, which, topo);
    if (!k || !k->fn) return 0;
    const bool ln = lean && k->fn_lean;
    *fn = reinterpret_cast<const void *>(ln ? k->fn_lean : k->fn);
    *fn_wide = reinterpret_cast<const void *>(ln ? k->fn_wide_lean : k->fn_wide);
    *lds_bytes = (ln && k->lds_bytes_lean) ? k->lds_bytes_lean : k->lds_bytes;  // the lean capped build may use the compact LDS map
    *lds_bytes_wide = k->lds_bytes;
    *topo_out = k->topo;
    return 1;
}

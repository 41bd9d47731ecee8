// dwbc_kernels.h -- __global__ entry points of the fused cycle and the table of instantiations.  Included by dwbc_capi.hip
// (DWBC_REAL = double, the product path) and by dwbc_kernels_f32.hip (DWBC_REAL = float with the namespace renamed to
// dwbc_f32): the same source in both arithmetic types, looked up at run time through KernelEntry.
#pragma once
#include <hip/hip_runtime.h>

#include "dwbc_reduced.h"
#include "dwbc_cycle2p.h"
#include "dwbc_cycle_gc.h"

namespace dwbc {

// register-resident kernel (dwbc_cycle2.h): one workgroup (one 64-lane wavefront) per robot instance.  Two builds of the same body:
//   _v2   amdgpu_waves_per_eu(2): VGPR + AGPR <= 256, so a fifth workgroup of a CU (the LDS map allows 5 at <= 31 KB) can
//         share a SIMD -- the throughput build for batches larger than 4 instances per CU
//   _v2w  no register cap (one wave per SIMD): ~7 % shorter single-instance latency -- used while B <= 4 x CUs
#define DWBC_V2_BODY(COMPACT)                                                            \
    static_assert(NT == 64, "one wavefront per instance");                               \
    extern __shared__ __attribute__((aligned(16))) real_t lds[];                         \
    const int inst = blockIdx.x;                                                         \
    if (inst >= io.B) return;                                                            \
    Thr th{(int)threadIdx.x};                                                            \
    int *iL = reinterpret_cast<int *>(lds + V2Lds<N, NB, NLV, COMPACT>::type::total);    \
    cycle_instance_v2<N, NB, NLV, NT, EXTRAS, Topo, COMPACT>(th, su, io, inst, lds, iL);
// EXTRAS: see cycle_instance_v2 -- false = the lean build the launcher uses when no optional path is requested
// Topo: a constant kinematic tree (dwbc_topo.h) whose sparsity the A^-1 sweep uses, or TopoGeneric
// COMPACT: the 20 KB LDS map (Lds3, dwbc_cycle2.h) -- the lean capped build of a constant-tree model takes it: eight workgroups
// share a CU (two waves per SIMD) instead of five
template <int N, int NB, int NLV, bool COMPACT>
struct V2Lds { using type = typename std::conditional<COMPACT, Lds3<N, NB, NLV>, Lds2<N, NB, NLV>>::type; };
template <int N, int NB, int NLV, int NT, bool EXTRAS, class Topo, bool COMPACT = false>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2))) void dwbc_cycle_kernel_v2(const Setup su, const BatchIO io) {
    DWBC_V2_BODY(COMPACT)
}
#ifndef DWBC_WIDE_ATTR
#define DWBC_WIDE_ATTR
#endif
template <int N, int NB, int NLV, int NT, bool EXTRAS, class Topo>
__global__ __launch_bounds__(NT) DWBC_WIDE_ATTR void dwbc_cycle_kernel_v2w(const Setup su, const BatchIO io) {
    DWBC_V2_BODY(false)
}

// two wavefronts per instance (dwbc_cycle2p.h): the lean cycle of batches of at most one instance per SIMD, side chains on a helper wave.
// roles: wave 0 is the main wave, except in the workgroups whose index has bit io.pair_swap_bit set (see dwbc_batch launch: the
// waves of consecutive workgroups of a CU land on SIMDs round-robin, and the main waves should not share one)
template <int N, int NB, int NLV, class Topo>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2))) void dwbc_cycle_kernel_v2p(const Setup su, const BatchIO io) {
    extern __shared__ __attribute__((aligned(16))) real_t lds[];
    const int inst = blockIdx.x;
    if (inst >= io.B) return;
    int wave = (int)(threadIdx.x >> 6);
    if (io.pair_swap_bit >= 0 && ((blockIdx.x >> io.pair_swap_bit) & 1)) wave ^= 1;
    Thr th{(int)(threadIdx.x & 63u)};
    cycle_instance_v2p<N, NB, NLV, 64, Topo>(wave, th, su, io, inst, lds);
}

// up to three simultaneously active contacts (dwbc_cycle_gc.h): the general statement of the cycle, matrices in LDS (80 KB: two
// workgroups per CU), any number of task levels -- for batches that opt in with dwbc_batch_set_max_active_contacts(b, 3)
// TG = 12: the same cycle sized for task levels of up to 12 dof (two 6D links on one level -- both hands, reference
// tests/sp_test/regulation_test.cpp:90-91): QPs of up to 24 variables, 105 KB of LDS, one workgroup per CU
constexpr int kGcContacts = 3;
template <int N, int NB, int NT, int TG = kMaxTaskDof>
__global__ __launch_bounds__(NT) void dwbc_cycle_kernel_gc(const Setup su, const BatchIO io) {
    static_assert(NT == 64, "one wavefront per instance");
    extern __shared__ __attribute__((aligned(16))) real_t lds[];
    const int inst = blockIdx.x;
    if (inst >= io.B) return;
    Thr th{(int)threadIdx.x};
    cycle_instance_gc<N, NB, kGcContacts, NT, TG>(th, su, io, inst, lds);
}

// reduced (centroidal) dynamics model, dwbc_reduced.h: Reduced* call sequence of reference include/dwbc.h:411-416
template <int N, int NB, int NLV, int NT, class Topo>
__global__ __launch_bounds__(NT) void dwbc_cycle_kernel_reduced(const Setup su, const BatchIO io) {
    static_assert(NT == 64, "one wavefront per instance");
    extern __shared__ __attribute__((aligned(16))) real_t lds[];
    const int inst = blockIdx.x;
    if (inst >= io.B) return;
    Thr th{(int)threadIdx.x};
    int *iL = reinterpret_cast<int *>(lds + LdsR<N, NB, NLV>::rtotal);
    cycle_instance_reduced<N, NB, NLV, NT, Topo>(th, su, io, inst, lds, iL);
}

constexpr int kNT = 64;
// the paired kernel exists in the fp64 build, for one and two task levels (three levels no longer fit four workgroups per CU)
#ifdef DWBC_NO_PAIR_KERNEL
#define DWBC_PAIR_ENTRY(NLV) nullptr, 0
#else
#define DWBC_PAIR_ENTRY(NLV) dwbc_cycle_kernel_v2p<39, 34, NLV, TopoTocabi>, Lds4<39, 34, NLV>::total_bytes
#endif

struct KernelEntry {
    int n, nb, nlv;  // nlv = task levels the LDS map is sized for (0: any)
    int topo;        // Setup::topo_kind the instantiation was built for (1: TopoTocabi's constant tree); 0 = any tree
    void (*fn)(const Setup, const BatchIO);
    int lds_bytes;
    void (*fn_wide)(const Setup, const BatchIO);  // uncapped-register build for batches of at most 4 instances per CU
    void (*fn_lean)(const Setup, const BatchIO);       // the same two without the optional paths (EXTRAS = false), or nullptr
    void (*fn_wide_lean)(const Setup, const BatchIO);
    int lds_bytes_lean;  // dynamic LDS of fn_lean when it differs from lds_bytes (the compact map), else 0
    void (*fn_pair)(const Setup, const BatchIO);  // two waves per instance (128 threads), lean, batches of at most 4 instances per CU; or nullptr
    int lds_bytes_pair;
};
// the general-contact kernel of a model size (any tree, any number of levels): one per (n, nb)
struct GcEntry {
    int n, nb;
    void (*fn)(const Setup, const BatchIO);
    int lds_bytes;
    void (*fn_wide_tasks)(const Setup, const BatchIO);  // task levels of up to kMaxTaskDofWide dof, or nullptr
    int lds_bytes_wide_tasks;
};
// instantiated model sizes (system dof, bodies).  TOCABI = (39, 34), the only model in BASELINE.json's configs: its four
// flavours use the constant tree; any other 34-body tree runs the TopoGeneric build (full flavour only).  Other model sizes come
// from kernel packs (dwbc_pack.hip: this header instantiated for one (N, NB), loaded by the C-ABI at model-load time).
#if !defined(DWBC_PACK_N) && !defined(DWBC_NO_PAIR_KERNEL)
const GcEntry kKernelsGc[] = {
    {39, 34, dwbc_cycle_kernel_gc<39, 34, kNT>, LdsG<39, 34, kGcContacts>::total_bytes,
     dwbc_cycle_kernel_gc<39, 34, kNT, kMaxTaskDofWide>, LdsG<39, 34, kGcContacts, kMaxTaskDofWide>::total_bytes},
};
inline const GcEntry *lookup_gc(int n, int nb) {
    for (const GcEntry &e : kKernelsGc)
        if (e.n == n && e.nb == nb) return &e;
    return nullptr;
}
#endif
#ifdef DWBC_PACK_N
#elif defined(DWBC_EXPERIMENT)
// A/B build (make experiment VARIANT=.. XFLAGS=..): only the BASELINE config[1] instantiation, seconds to compile
const KernelEntry kKernels[] = {
    {39, 34, 2, 1, dwbc_cycle_kernel_v2<39, 34, 2, kNT, true, TopoTocabi>, Lds2<39, 34, 2>::total_bytes, dwbc_cycle_kernel_v2w<39, 34, 2, kNT, true, TopoTocabi>,
     dwbc_cycle_kernel_v2<39, 34, 2, kNT, false, TopoTocabi, true>, dwbc_cycle_kernel_v2w<39, 34, 2, kNT, false, TopoTocabi>, Lds3<39, 34, 2>::total_bytes,
     DWBC_PAIR_ENTRY(2)},
};
const KernelEntry kKernelsReduced[] = {
    {39, 34, 2, 1, dwbc_cycle_kernel_reduced<39, 34, 2, kNT, TopoTocabi>, LdsR<39, 34, 2>::total_bytes, nullptr, nullptr, nullptr},
};
#else
const KernelEntry kKernels[] = {
    {39, 34, 1, 1, dwbc_cycle_kernel_v2<39, 34, 1, kNT, true, TopoTocabi>, Lds2<39, 34, 1>::total_bytes, dwbc_cycle_kernel_v2w<39, 34, 1, kNT, true, TopoTocabi>,
     dwbc_cycle_kernel_v2<39, 34, 1, kNT, false, TopoTocabi, true>, dwbc_cycle_kernel_v2w<39, 34, 1, kNT, false, TopoTocabi>, Lds3<39, 34, 1>::total_bytes,
     DWBC_PAIR_ENTRY(1)},
    {39, 34, 2, 1, dwbc_cycle_kernel_v2<39, 34, 2, kNT, true, TopoTocabi>, Lds2<39, 34, 2>::total_bytes, dwbc_cycle_kernel_v2w<39, 34, 2, kNT, true, TopoTocabi>,
     dwbc_cycle_kernel_v2<39, 34, 2, kNT, false, TopoTocabi, true>, dwbc_cycle_kernel_v2w<39, 34, 2, kNT, false, TopoTocabi>, Lds3<39, 34, 2>::total_bytes,
     DWBC_PAIR_ENTRY(2)},
    {39, 34, 3, 1, dwbc_cycle_kernel_v2<39, 34, 3, kNT, true, TopoTocabi>, Lds2<39, 34, 3>::total_bytes, dwbc_cycle_kernel_v2w<39, 34, 3, kNT, true, TopoTocabi>,
     dwbc_cycle_kernel_v2<39, 34, 3, kNT, false, TopoTocabi, true>, dwbc_cycle_kernel_v2w<39, 34, 3, kNT, false, TopoTocabi>, Lds3<39, 34, 3>::total_bytes},
    {39, 34, 4, 1, dwbc_cycle_kernel_v2<39, 34, 4, kNT, true, TopoTocabi>, Lds2<39, 34, 4>::total_bytes, dwbc_cycle_kernel_v2w<39, 34, 4, kNT, true, TopoTocabi>,
     dwbc_cycle_kernel_v2<39, 34, 4, kNT, false, TopoTocabi, true>, dwbc_cycle_kernel_v2w<39, 34, 4, kNT, false, TopoTocabi>, Lds3<39, 34, 4>::total_bytes},
    {39, 34, 1, 0, dwbc_cycle_kernel_v2<39, 34, 1, kNT, true, TopoGeneric>, Lds2<39, 34, 1>::total_bytes, nullptr, nullptr, nullptr},
    {39, 34, 2, 0, dwbc_cycle_kernel_v2<39, 34, 2, kNT, true, TopoGeneric>, Lds2<39, 34, 2>::total_bytes, nullptr, nullptr, nullptr},
    {39, 34, 3, 0, dwbc_cycle_kernel_v2<39, 34, 3, kNT, true, TopoGeneric>, Lds2<39, 34, 3>::total_bytes, nullptr, nullptr, nullptr},
    {39, 34, 4, 0, dwbc_cycle_kernel_v2<39, 34, 4, kNT, true, TopoGeneric>, Lds2<39, 34, 4>::total_bytes, nullptr, nullptr, nullptr},
};
const KernelEntry kKernelsReduced[] = {
    {39, 34, 1, 1, dwbc_cycle_kernel_reduced<39, 34, 1, kNT, TopoTocabi>, LdsR<39, 34, 1>::total_bytes, nullptr, nullptr, nullptr},
    {39, 34, 2, 1, dwbc_cycle_kernel_reduced<39, 34, 2, kNT, TopoTocabi>, LdsR<39, 34, 2>::total_bytes, nullptr, nullptr, nullptr},
    {39, 34, 3, 1, dwbc_cycle_kernel_reduced<39, 34, 3, kNT, TopoTocabi>, LdsR<39, 34, 3>::total_bytes, nullptr, nullptr, nullptr},
    {39, 34, 4, 1, dwbc_cycle_kernel_reduced<39, 34, 4, kNT, TopoTocabi>, LdsR<39, 34, 4>::total_bytes, nullptr, nullptr, nullptr},
    {39, 34, 1, 0, dwbc_cycle_kernel_reduced<39, 34, 1, kNT, TopoGeneric>, LdsR<39, 34, 1>::total_bytes, nullptr, nullptr, nullptr},
    {39, 34, 2, 0, dwbc_cycle_kernel_reduced<39, 34, 2, kNT, TopoGeneric>, LdsR<39, 34, 2>::total_bytes, nullptr, nullptr, nullptr},
    {39, 34, 3, 0, dwbc_cycle_kernel_reduced<39, 34, 3, kNT, TopoGeneric>, LdsR<39, 34, 3>::total_bytes, nullptr, nullptr, nullptr},
    {39, 34, 4, 0, dwbc_cycle_kernel_reduced<39, 34, 4, kNT, TopoGeneric>, LdsR<39, 34, 4>::total_bytes, nullptr, nullptr, nullptr},
};
#endif
#ifndef DWBC_PACK_N
// which: 0 = full-model kernel, 2 = reduced dynamics
// topo: Setup::topo_kind of the loaded model -- an instantiation for that constant tree is preferred, else the generic one
inline const KernelEntry *lookup_kernel(int n, int nb, int nlv, int which, int topo) {
    for (int pass = 0; pass < 2; pass++) {
        const int want = pass == 0 ? topo : 0;
        if (which == 2) {
            for (const auto &k : kKernelsReduced)
                if (k.n == n && k.nb == nb && k.nlv == nlv && k.topo == want) return &k;
        } else {
            for (const auto &k : kKernels)
                if (k.n == n && k.nb == nb && k.nlv == nlv && k.topo == want) return &k;
        }
    }
    return nullptr;
}
#endif

// what a pack and the library that loads it must agree on (both are built from this header): the sizes of the shared structures
// and a hash of the kernel sources (Makefile: DWBC_SRC_HASH), so that a pack left over from older kernel code is refused
#ifndef DWBC_SRC_HASH
#define DWBC_SRC_HASH 0u
#endif
inline unsigned kernel_abi_tag() { return (unsigned)(DWBC_SRC_HASH) ^ (unsigned)(sizeof(Setup) * 2654435761u) ^ (unsigned)(sizeof(BatchIO) * 40503u) ^ (unsigned)(DG_COUNT * 97u) ^ (unsigned)sizeof(KernelEntry); }

}  // namespace dwbc

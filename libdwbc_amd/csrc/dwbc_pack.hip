// dwbc_pack.hip -- the fused cycle kernels for ONE model size other than TOCABI's, as a loadable pack:
//   hipcc ... -DDWBC_PACK_N=<system dof> -DDWBC_PACK_NB=<bodies> -shared -o ../libdwbc_pack_<N>_<NB>.so dwbc_pack.hip
// (`make pack N=.. NB=..`; libdwbc_amd.build_pack(model) from Python).  The reference is model-generic (any URDF RBDL reads:
// src/dwbc.cpp:140-277, tests/dof_test/*.urdf); here the kernels keep their matrices in registers, so the sizes are template
// arguments and each (N, NB) is compiled once.  TopoGeneric build (dense A^-1 sweep), fp64, full-model cycle; the reduced
// (centroidal) path is laid out for TOCABI's tree and is not part of a pack.  dwbc_batch_create() dlopens the pack of the
// loaded model's size from the directory of libdwbc_hip.so (or $DWBC_PACK_DIR).
#ifndef DWBC_PACK_N
#error "compile with -DDWBC_PACK_N=<system dof> -DDWBC_PACK_NB=<bodies>"
#endif
#include "dwbc_kernels.h"

using namespace dwbc;

static_assert(DWBC_PACK_N == DWBC_PACK_NB + 5, "floating base (6 dof) + one revolute joint per further body");
static_assert(DWBC_PACK_N <= 50, "one lane per column, and the wave QP keeps (N - 6) + 20 rows on 64 lanes");
static_assert(DWBC_PACK_NB <= kMaxBodies, "body table");

// with -DDWBC_PACK_PARENTS=... the pack is built for that one kinematic tree (TopoPack, dwbc_topo.h: tree-sparse A^-1 sweep, compile-time
// round counts) and is only used for a model with exactly that parent table; without it, for any tree of the size (TopoGeneric)
#ifdef DWBC_PACK_PARENTS
#define DWBC_PACK_TOPO TopoPack
#define DWBC_PACK_KIND 2
#else
#define DWBC_PACK_TOPO TopoGeneric
#define DWBC_PACK_KIND 0
#endif
#define DWBC_PACK_ENTRY(NLV)                                                                                                        \
    {DWBC_PACK_N, DWBC_PACK_NB, NLV, DWBC_PACK_KIND, dwbc_cycle_kernel_v2<DWBC_PACK_N, DWBC_PACK_NB, NLV, kNT, true, DWBC_PACK_TOPO>,  \
     Lds2<DWBC_PACK_N, DWBC_PACK_NB, NLV>::total_bytes, dwbc_cycle_kernel_v2w<DWBC_PACK_N, DWBC_PACK_NB, NLV, kNT, true, DWBC_PACK_TOPO>, \
     dwbc_cycle_kernel_v2<DWBC_PACK_N, DWBC_PACK_NB, NLV, kNT, false, DWBC_PACK_TOPO>,                                               \
     dwbc_cycle_kernel_v2w<DWBC_PACK_N, DWBC_PACK_NB, NLV, kNT, false, DWBC_PACK_TOPO>}
#ifdef DWBC_PACK_ONLY_NLV  // development: one level count only (a quarter of the compile time)
static const KernelEntry kPack[] = {DWBC_PACK_ENTRY(DWBC_PACK_ONLY_NLV)};
#else
static const KernelEntry kPack[] = {DWBC_PACK_ENTRY(1), DWBC_PACK_ENTRY(2), DWBC_PACK_ENTRY(3), DWBC_PACK_ENTRY(4)};
#endif

#ifdef DWBC_PACK_PARENTS
// the tree this pack was compiled for: the loader compares it with the model's
extern "C" const int *dwbc_pack_parents(int *nb) {
    *nb = TopoPack::nb;
    return TopoPack::parent;
}
#endif
// the general-contact kernel (dwbc_cycle_gc.h: up to three simultaneously active contacts) of this model size, when its QP rows fit one
// per lane ((N - 6) torque rows + 30 cone rows <= 64)
#if (DWBC_PACK_N - 6 + 10 * 3) <= 64
extern "C" const void *dwbc_pack_gc(int *lds_bytes) {
    *lds_bytes = LdsG<DWBC_PACK_N, DWBC_PACK_NB, kGcContacts>::total_bytes;
    return reinterpret_cast<const void *>(dwbc_cycle_kernel_gc<DWBC_PACK_N, DWBC_PACK_NB, kNT>);
}
#endif
extern "C" const KernelEntry *dwbc_pack_table(int *count, unsigned *abi_tag) {
    *count = (int)(sizeof(kPack) / sizeof(kPack[0]));
    *abi_tag = kernel_abi_tag();
    return kPack;
}

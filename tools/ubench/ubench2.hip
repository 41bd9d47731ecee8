// Second micro-benchmark round: issue cost of the fp64 building blocks measured in SHADER cycles (s_memtime) and in
// wall time (s_memrealtime, 100 MHz), with hand-written inner loops so that the ISA under test is what is timed.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench/ubench2.hip -o gpurun_out/ubench2 && gpurun_out/ubench2
// Every test: G single-wave workgroups (G = 1024 * waves_per_SIMD), two warm-up launches, then a launch long enough
// (>= 1 ms) for the clock to have settled.  Reported: shader cycles per instruction, the in-kernel clock
// (delta s_memtime / delta s_memrealtime * 100 MHz) and the whole-chip rate from HIP events.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct Stamp { unsigned long long cyc, rt; };
__device__ __forceinline__ Stamp stamp() {
    Stamp s;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(s.cyc), "=s"(s.rt)::"memory");
    return s;
}

// mode:
//  0  32 independent v_fma_f64 (8 accumulators x 4)            1  32 dependent v_fma_f64 (one chain)
//  2  32 independent v_fma_f32 (8 accumulators)                3  32 v_pk_fma_f32
//  4  32 x (2 v_readlane_b32 + v_fma_f64 with the SGPR pair)   5  64 v_readlane_b32 alone
//  6  32 x v_fma_f64 with an SGPR-pair operand (no readlane)   7  32 x (v_mov_b32_dpp x2 quad_perm + v_fma_f64)
//  8  32 x (ds_read_b64 broadcast + v_fma_f64), reads issued 8 ahead     9  16 x (ds_read_b128 broadcast + 2 v_fma_f64)
// 10  32 x (ds_bpermute_b32 x2 + v_fma_f64)                   11  32 s_nop 0
// 12  32 v_mul_f64                                             13  32 v_add_f64
// 14  32 v_cndmask_b32                                         15  32 v_mov_b32
// 16  v_fma_f64 with DPP row_newbcast? (not on gfx9) -> skipped
template <int MODE>
__global__ __launch_bounds__(64) void k_issue(const double* in, double* out, unsigned long long* cyc, int reps) {
    __shared__ double buf[128];
    const int lane = threadIdx.x;
    double a0 = in[lane], a1 = in[64 + lane], a2 = a0 * 1.5, a3 = a1 * 0.5, a4 = a0 + 1, a5 = a1 + 2, a6 = a0 - 1, a7 = a1 - 2;
    double x = in[lane] * 1e-3, y = 0.999999;
    float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3, f4_ = (float)a4, f5 = (float)a5, f6 = (float)a6, f7 = (float)a7, fx = 1e-3f;
    int idx = ((lane + 1) & 63) * 4;
    buf[lane] = x; buf[64 + lane] = y;
    __syncthreads();
    const unsigned ldsbase = (unsigned)(size_t)buf;
    Stamp s0 = stamp();
    for (int r = 0; r < reps; r++) {
        if constexpr (MODE == 0) {
            asm volatile(
                ".rept 4\n"
                "v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n"
                "v_fma_f64 %4, %8, %9, %4\n v_fma_f64 %5, %8, %9, %5\n v_fma_f64 %6, %8, %9, %6\n v_fma_f64 %7, %8, %9, %7\n"
                ".endr\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        }
        if constexpr (MODE == 1) {
            asm volatile(".rept 32\n v_fma_f64 %0, %1, %2, %0\n .endr\n" : "+v"(a0) : "v"(x), "v"(y));
        }
        if constexpr (MODE == 2) {
            asm volatile(
                ".rept 4\n"
                "v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"
                ".endr\n"
                : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4_), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fx), "v"(fx));
        }
        if constexpr (MODE == 3) {
            asm volatile(
                ".rept 4\n"
                "v_pk_fma_f32 %0, %8, %9, %0\n v_pk_fma_f32 %1, %8, %9, %1\n v_pk_fma_f32 %2, %8, %9, %2\n v_pk_fma_f32 %3, %8, %9, %3\n"
                "v_pk_fma_f32 %4, %8, %9, %4\n v_pk_fma_f32 %5, %8, %9, %5\n v_pk_fma_f32 %6, %8, %9, %6\n v_pk_fma_f32 %7, %8, %9, %7\n"
                ".endr\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        }
        if constexpr (MODE == 4) {
            asm volatile(
                ".rept 4\n"
                "v_readlane_b32 s20, %8, 3\n v_readlane_b32 s21, %9, 3\n s_nop 1\n v_fma_f64 %0, s[20:21], %10, %0\n"
                "v_readlane_b32 s22, %8, 5\n v_readlane_b32 s23, %9, 5\n s_nop 1\n v_fma_f64 %1, s[22:23], %10, %1\n"
                "v_readlane_b32 s20, %8, 7\n v_readlane_b32 s21, %9, 7\n s_nop 1\n v_fma_f64 %2, s[20:21], %10, %2\n"
                "v_readlane_b32 s22, %8, 9\n v_readlane_b32 s23, %9, 9\n s_nop 1\n v_fma_f64 %3, s[22:23], %10, %3\n"
                "v_readlane_b32 s20, %8, 11\n v_readlane_b32 s21, %9, 11\n s_nop 1\n v_fma_f64 %4, s[20:21], %10, %4\n"
                "v_readlane_b32 s22, %8, 13\n v_readlane_b32 s23, %9, 13\n s_nop 1\n v_fma_f64 %5, s[22:23], %10, %5\n"
                "v_readlane_b32 s20, %8, 15\n v_readlane_b32 s21, %9, 15\n s_nop 1\n v_fma_f64 %6, s[20:21], %10, %6\n"
                "v_readlane_b32 s22, %8, 17\n v_readlane_b32 s23, %9, 17\n s_nop 1\n v_fma_f64 %7, s[22:23], %10, %7\n"
                ".endr\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                : "v"(((int*)&x)[0]), "v"(((int*)&x)[1]), "v"(y) : "s20", "s21", "s22", "s23");
        }
        if constexpr (MODE == 5) {
            asm volatile(".rept 32\n v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n .endr\n" ::"v"(((int*)&x)[0]), "v"(((int*)&x)[1]) : "s20", "s21");
        }
        if constexpr (MODE == 6) {
            asm volatile(
                "v_readlane_b32 s20, %8, 3\n v_readlane_b32 s21, %9, 3\n s_nop 4\n"
                ".rept 4\n"
                "v_fma_f64 %0, s[20:21], %10, %0\n v_fma_f64 %1, s[20:21], %10, %1\n v_fma_f64 %2, s[20:21], %10, %2\n v_fma_f64 %3, s[20:21], %10, %3\n"
                "v_fma_f64 %4, s[20:21], %10, %4\n v_fma_f64 %5, s[20:21], %10, %5\n v_fma_f64 %6, s[20:21], %10, %6\n v_fma_f64 %7, s[20:21], %10, %7\n"
                ".endr\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                : "v"(((int*)&x)[0]), "v"(((int*)&x)[1]), "v"(y) : "s20", "s21");
        }
        if constexpr (MODE == 7) {
            int t0, t1;
            asm volatile(
                ".rept 4\n"
                "v_mov_b32_dpp %8, %10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %9, %11 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fma_f64 %0, %[t], %12, %0\n"
                "v_mov_b32_dpp %8, %10 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %9, %11 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fma_f64 %1, %[t], %12, %1\n"
                "v_mov_b32_dpp %8, %10 row_ror:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %9, %11 row_ror:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fma_f64 %2, %[t], %12, %2\n"
                "v_mov_b32_dpp %8, %10 row_ror:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %9, %11 row_ror:3 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fma_f64 %3, %[t], %12, %3\n"
                "v_mov_b32_dpp %8, %10 row_ror:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %9, %11 row_ror:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fma_f64 %4, %[t], %12, %4\n"
                "v_mov_b32_dpp %8, %10 row_ror:5 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %9, %11 row_ror:5 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fma_f64 %5, %[t], %12, %5\n"
                "v_mov_b32_dpp %8, %10 row_ror:6 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %9, %11 row_ror:6 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fma_f64 %6, %[t], %12, %6\n"
                "v_mov_b32_dpp %8, %10 row_ror:7 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %9, %11 row_ror:7 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_fma_f64 %7, %[t], %12, %7\n"
                ".endr\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(t0), "=&v"(t1)
                : "v"(((int*)&x)[0]), "v"(((int*)&x)[1]), "v"(y), [t] "v"(x));
            (void)t0; (void)t1;
        }
        if constexpr (MODE == 8) {
            double b0, b1, b2, b3, b4, b5, b6, b7;
            asm volatile(
                ".rept 4\n"
                "ds_read_b64 %8, %16\n ds_read_b64 %9, %16 offset:8\n ds_read_b64 %10, %16 offset:16\n ds_read_b64 %11, %16 offset:24\n"
                "ds_read_b64 %12, %16 offset:32\n ds_read_b64 %13, %16 offset:40\n ds_read_b64 %14, %16 offset:48\n ds_read_b64 %15, %16 offset:56\n"
                "s_waitcnt lgkmcnt(7)\n v_fma_f64 %0, %8, %17, %0\n s_waitcnt lgkmcnt(6)\n v_fma_f64 %1, %9, %17, %1\n"
                "s_waitcnt lgkmcnt(5)\n v_fma_f64 %2, %10, %17, %2\n s_waitcnt lgkmcnt(4)\n v_fma_f64 %3, %11, %17, %3\n"
                "s_waitcnt lgkmcnt(3)\n v_fma_f64 %4, %12, %17, %4\n s_waitcnt lgkmcnt(2)\n v_fma_f64 %5, %13, %17, %5\n"
                "s_waitcnt lgkmcnt(1)\n v_fma_f64 %6, %14, %17, %6\n s_waitcnt lgkmcnt(0)\n v_fma_f64 %7, %15, %17, %7\n"
                ".endr\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                  "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(b4), "=&v"(b5), "=&v"(b6), "=&v"(b7)
                : "v"(ldsbase), "v"(y));
        }
        if constexpr (MODE == 9) {
            asm volatile(
                ".rept 4\n"
                "ds_read_b128 v[100:103], %8\n ds_read_b128 v[104:107], %8 offset:16\n ds_read_b128 v[108:111], %8 offset:32\n ds_read_b128 v[112:115], %8 offset:48\n"
                "s_waitcnt lgkmcnt(3)\n v_fma_f64 %0, v[100:101], %9, %0\n v_fma_f64 %1, v[102:103], %9, %1\n"
                "s_waitcnt lgkmcnt(2)\n v_fma_f64 %2, v[104:105], %9, %2\n v_fma_f64 %3, v[106:107], %9, %3\n"
                "s_waitcnt lgkmcnt(1)\n v_fma_f64 %4, v[108:109], %9, %4\n v_fma_f64 %5, v[110:111], %9, %5\n"
                "s_waitcnt lgkmcnt(0)\n v_fma_f64 %6, v[112:113], %9, %6\n v_fma_f64 %7, v[114:115], %9, %7\n"
                ".endr\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                : "v"(ldsbase), "v"(y)
                : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115");
        }
        if constexpr (MODE == 10) {
            int t0, t1;
            asm volatile(
                ".rept 32\n"
                "ds_bpermute_b32 %1, %3, %4\n ds_bpermute_b32 %2, %3, %5\n s_waitcnt lgkmcnt(0)\n v_fma_f64 %0, %[t], %6, %0\n"
                ".endr\n"
                : "+v"(a0), "=&v"(t0), "=&v"(t1) : "v"(idx), "v"(((int*)&x)[0]), "v"(((int*)&x)[1]), "v"(y), [t] "v"(x));
            (void)t0; (void)t1;
        }
        if constexpr (MODE == 11) { asm volatile(".rept 32\n s_nop 0\n .endr\n"); }
        if constexpr (MODE == 12) {
            asm volatile(
                ".rept 4\n"
                "v_mul_f64 %0, %8, %0\n v_mul_f64 %1, %8, %1\n v_mul_f64 %2, %8, %2\n v_mul_f64 %3, %8, %3\n"
                "v_mul_f64 %4, %8, %4\n v_mul_f64 %5, %8, %5\n v_mul_f64 %6, %8, %6\n v_mul_f64 %7, %8, %7\n"
                ".endr\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));
        }
        if constexpr (MODE == 13) {
            asm volatile(
                ".rept 4\n"
                "v_add_f64 %0, %8, %0\n v_add_f64 %1, %8, %1\n v_add_f64 %2, %8, %2\n v_add_f64 %3, %8, %3\n"
                "v_add_f64 %4, %8, %4\n v_add_f64 %5, %8, %5\n v_add_f64 %6, %8, %6\n v_add_f64 %7, %8, %7\n"
                ".endr\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
        }
        if constexpr (MODE == 14) {
            asm volatile(".rept 32\n v_cndmask_b32 %0, %1, %2, vcc\n .endr\n" : "=&v"(idx) : "v"(lane), "v"(reps) : "vcc");
        }
        if constexpr (MODE == 15) {
            asm volatile(".rept 16\n v_mov_b32 %0, %2\n v_mov_b32 %1, %2\n .endr\n" : "=&v"(idx), "=&v"(f7) : "v"(lane));
        }
    }
    Stamp s1 = stamp();
    if (lane == 0) { cyc[2 * blockIdx.x] = s1.cyc - s0.cyc; cyc[2 * blockIdx.x + 1] = s1.rt - s0.rt; }
    out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4_ + f5 + f6 + f7 + idx;
}

// MFMA issue rates.  mode 0: f64 16x16x4, nacc independent accumulators; 1: f64 4x4x4 (4 blocks); 2: f32 16x16x4 (calibration:
// the guide measures 32 cycles per instruction)
template <int mode, int nacc>
__global__ __launch_bounds__(64) void k_mfma(const double* in, double* out, unsigned long long* cyc, int reps) {
    const int lane = threadIdx.x;
    const double a = in[lane], b = in[64 + lane];
    d4 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = d4{0, 0, 0, 0};
    f4 facc[4];
    for (int i = 0; i < 4; i++) facc[i] = f4{0, 0, 0, 0};
    double acc1[4] = {0, 0, 0, 0};
    const float fa = (float)a, fb = (float)b;
    Stamp s0 = stamp();
    for (int r = 0; r < reps; r++) {
        if (mode == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (i < nacc) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        } else if (mode == 1) {
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (i < nacc) acc1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (i < nacc) facc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, facc[i], 0, 0, 0);
        }
    }
    Stamp s1 = stamp();
    if (lane == 0) { cyc[2 * blockIdx.x] = s1.cyc - s0.cyc; cyc[2 * blockIdx.x + 1] = s1.rt - s0.rt; }
    double s = 0;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 4; i++) s += facc[i][0] + facc[i][3] + acc1[i];
    out[blockIdx.x * 64 + lane] = s;
}

// MFMA beside VALU in ONE wave: per iteration nm independent f64 16x16x4 MFMAs and nv independent v_fma_f64 -- do they overlap?
template <int nv>
__global__ __launch_bounds__(64) void k_mix(const double* in, double* out, unsigned long long* cyc, int reps) {
    const int lane = threadIdx.x;
    const double a = in[lane], b = in[64 + lane];
    d4 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = d4{0, 0, 0, 0};
    double a0 = a, a1 = b, a2 = a + 1, a3 = b + 1, a4 = a - 1, a5 = b - 1, a6 = a * 2, a7 = b * 2, x = 1e-3, y = 0.99999;
    Stamp s0 = stamp();
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            if (nv >= 8)
                asm volatile("v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n"
                             "v_fma_f64 %4, %8, %9, %4\n v_fma_f64 %5, %8, %9, %5\n v_fma_f64 %6, %8, %9, %6\n v_fma_f64 %7, %8, %9, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
            else if (nv >= 4)
                asm volatile("v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %2, %4, %5, %2\n v_fma_f64 %3, %4, %5, %3\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));
        }
    }
    Stamp s1 = stamp();
    if (lane == 0) { cyc[2 * blockIdx.x] = s1.cyc - s0.cyc; cyc[2 * blockIdx.x + 1] = s1.rt - s0.rt; }
    double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 64 + lane] = s;
}

static void med(std::vector<unsigned long long>& c, int G, double* cyc, double* rt) {
    std::vector<unsigned long long> a(G), b(G);
    for (int i = 0; i < G; i++) { a[i] = c[2 * i]; b[i] = c[2 * i + 1]; }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    *cyc = (double)a[G / 2]; *rt = (double)b[G / 2];
}

int main() {
    const int G0 = 1024;
    double *din, *dout; unsigned long long* dcyc;
    CK(hipMalloc(&din, 1 << 16)); CK(hipMalloc(&dout, 8 * G0 * 64 * 8)); CK(hipMalloc(&dcyc, 8 * G0 * 16));
    std::vector<double> h(256);
    for (int l = 0; l < 256; l++) h[l] = 1.0 + 1e-3 * ((l * 37) % 101);
    CK(hipMemcpy(din, h.data(), 256 * 8, hipMemcpyHostToDevice));
    std::vector<unsigned long long> c(2 * 8 * G0);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* nm[] = {"v_fma_f64 x32, 8 independent acc", "v_fma_f64 x32, one dependent chain", "v_fma_f32 x32, 8 independent acc",
                        "v_pk_fma_f32 x32, 8 independent acc", "(2 v_readlane + s_nop 1 + v_fma_f64 sgpr) x32", "v_readlane_b32 x64",
                        "v_fma_f64 (SGPR-pair operand) x32", "(2 v_mov_dpp + s_nop 1 + v_fma_f64) x32", "(ds_read_b64 bcast + v_fma_f64) x32, 8 in flight",
                        "(ds_read_b128 bcast + 2 v_fma_f64) x16, 4 in flight", "(2 ds_bpermute + wait + v_fma_f64) x32 dependent", "s_nop 0 x32",
                        "v_mul_f64 x32", "v_add_f64 x32", "v_cndmask_b32 x32", "v_mov_b32 x32"};
    const int per[] = {32, 32, 32, 32, 32, 64, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32};
    typedef void (*issue_fn)(const double*, double*, unsigned long long*, int);
    const issue_fn ifn[16] = {k_issue<0>, k_issue<1>, k_issue<2>, k_issue<3>, k_issue<4>, k_issue<5>, k_issue<6>, k_issue<7>,
                              k_issue<8>, k_issue<9>, k_issue<10>, k_issue<11>, k_issue<12>, k_issue<13>, k_issue<14>, k_issue<15>};
    printf("== issue cost (1024 x w single-wave workgroups), shader cycles per instruction group ==\n");
    for (int w : {1, 2, 4, 8}) {
        const int G = G0 * w;
        printf("-- %d wave(s) per SIMD\n", w);
        for (int mode = 0; mode < 16; mode++) {
            const int reps = 20000;
            for (int i = 0; i < 2; i++) ifn[mode]<<<G, 64>>>(din, dout, dcyc, reps);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); ifn[mode]<<<G, 64>>>(din, dout, dcyc, reps); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(c.data(), dcyc, G * 16, hipMemcpyDeviceToHost));
            double cy, rt; med(c, G, &cy, &rt);
            printf("%-58s %7.2f cyc each | clock %.2f GHz | %.2f ms\n", nm[mode], cy / ((double)reps * per[mode]), cy / rt * 0.1, ms);
        }
    }
    printf("== MFMA issue ==\n");
    typedef void (*mfma_fn)(const double*, double*, unsigned long long*, int);
    struct MF { int mode, nacc; mfma_fn fn; };
    const MF mf[] = {{0, 1, k_mfma<0, 1>}, {0, 2, k_mfma<0, 2>}, {0, 4, k_mfma<0, 4>}, {0, 8, k_mfma<0, 8>},
                     {1, 1, k_mfma<1, 1>}, {1, 2, k_mfma<1, 2>}, {1, 4, k_mfma<1, 4>},
                     {2, 1, k_mfma<2, 1>}, {2, 2, k_mfma<2, 2>}, {2, 4, k_mfma<2, 4>}};
    for (int w : {1, 2, 4}) {
        const int G = G0 * w;
        for (const MF& m : mf) {
            const int reps = 20000;
            for (int i = 0; i < 2; i++) m.fn<<<G, 64>>>(din, dout, dcyc, reps);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); m.fn<<<G, 64>>>(din, dout, dcyc, reps); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(c.data(), dcyc, G * 16, hipMemcpyDeviceToHost));
            double cy, rt; med(c, G, &cy, &rt);
            const double fl = (m.mode == 1 ? 2.0 * 256 : 2.0 * 1024) * m.nacc * reps * (double)G;
            printf("%d w/SIMD %-18s %d acc: %7.1f cyc per MFMA | clock %.2f GHz | %.2f ms | %.1f TFLOP/s\n", w,
                   m.mode == 0 ? "mfma_f64_16x16x4" : m.mode == 1 ? "mfma_f64_4x4x4" : "mfma_f32_16x16x4", m.nacc, cy / ((double)reps * m.nacc), cy / rt * 0.1, ms, fl / ms * 1e-9);
        }
    }
    printf("== one wave: 4 x (mfma_f64_16x16x4 + nv v_fma_f64) per iteration ==\n");
    const mfma_fn xf[3] = {k_mix<0>, k_mix<4>, k_mix<8>};
    for (int xi = 0; xi < 3; xi++) {
        const int reps = 20000;
        for (int i = 0; i < 2; i++) xf[xi]<<<G0, 64>>>(din, dout, dcyc, reps);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); xf[xi]<<<G0, 64>>>(din, dout, dcyc, reps); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(c.data(), dcyc, G0 * 16, hipMemcpyDeviceToHost));
        double cy, rt; med(c, G0, &cy, &rt);
        printf("nv = %d: %7.1f cyc per (MFMA + nv FMA) | clock %.2f GHz | %.2f ms\n", xi * 4, cy / ((double)reps * 4), cy / rt * 0.1, ms);
    }
    return 0;
}

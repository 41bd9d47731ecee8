// dwbc_wave.h -- wave-synchronous programming helpers (one 64-lane wavefront = one robot instance).
//
// Device build: a "per-lane" variable is an ordinary register variable, LANES{...} is a plain block executed by all
// 64 lanes in lockstep, cross-lane traffic uses v_readlane / ds_bpermute.
// Host emulation build (tests/emu, DWBC_HOST_EMU): a per-lane variable is an array of 64, LANES{...} loops over the
// lanes, cross-lane primitives index that array.  The emulation is exact as long as a LANES block never reads a
// per-lane value that another lane writes inside the same block -- the same rule the hardware needs a barrier or a
// cross-lane instruction for.
#pragma once
#include <math.h>

#include "dwbc_types.h"

#ifdef DWBC_HOST_EMU
#define PL(type, x) type x[64]
#define PLA(type, x, n) type x[64][n]
#define LV(x) x[lane]
#define LANES for (int lane = 0; lane < 64; ++lane)
#define WSYNC() ((void)0)
#define DWBC_LANE_DECL ((void)0)
#define DWBC_LANE_OPAQUE(name) const int name = lane
#define DWBC_LANE_OPAQUE_IF(name, cond) const int name = lane
#define DWBC_FLAG_VGPR(f) ((void)0)
#define DWBC_FLAG_UNIFORM(f) (f)
#define BCAST(x, src) ((x)[(src)])                 /* uniform value of per-lane scalar x in lane src */
#define BCASTA(x, j, src) ((x)[(src)][(j)])        /* uniform value of per-lane array element x[j] in lane src */
#define SHFLA(x, j, src) ((x)[(src)][(j)])         /* inside LANES: x[j] of lane `src` (src may differ per lane) */
#define SHFL(x, src) ((x)[(src)])
#define DWBC_WDEV inline
#define PLA_REF(type, name, n) type (&name)[64][n]
#define PL_REF(type, name) type (&name)[64]
#else
#define PL(type, x) type x
#define PLA(type, x, n) type x[n]
#define LV(x) x
#define LANES
#ifdef DWBC_BLOCK_BARRIER
#define WSYNC() __syncthreads()
#else
#define WSYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")  /* see DWBC_SYNC in dwbc_cycle.h */
#endif
// (the lane inside the wavefront: the paired kernel of dwbc_cycle2p.h runs two wavefronts per workgroup)
#define DWBC_LANE_DECL const int lane = (int)(threadIdx.x & 63u)
// a copy of the lane index the optimiser cannot identify with `lane`: comparisons against it are not merged with (and kept
// alive across) the same comparisons elsewhere -- 39 hoisted `lane == K` masks are 78 SGPRs, i.e. spills through v_writelane
#define DWBC_LANE_OPAQUE(name) int name = lane; asm volatile("" : "+v"(name))
#define DWBC_LANE_OPAQUE_IF(name, cond) int name = lane; if constexpr (cond) { asm volatile("" : "+v"(name)); }
// a wave-uniform flag kept in a VGPR while it is accumulated over unrolled steps (as lane masks, N pending compare results
// are 2 N SGPRs), and read back as a scalar at the end
#define DWBC_FLAG_VGPR(f) asm volatile("" : "+v"(f))
#define DWBC_FLAG_UNIFORM(f) __builtin_amdgcn_readfirstlane(f)
#define BCAST(x, src) dwbc::readlane_f64((x), (src))
#define BCASTA(x, j, src) dwbc::readlane_f64((x)[(j)], (src))
#define SHFLA(x, j, src) __shfl((x)[(j)], (src), 64)
#define SHFL(x, src) __shfl((x), (src), 64)
#define DWBC_WDEV __device__ __forceinline__
#define PLA_REF(type, name, n) type (&name)[n]
#define PL_REF(type, name) type &name
#endif

namespace dwbc {

#ifndef DWBC_HOST_EMU
__device__ __forceinline__ double readlane_f64(double v, int srclane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), srclane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), srclane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float readlane_f64(float v, int srclane) {  // fp32 build: one v_readlane
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), srclane));
}
__device__ __forceinline__ int readlane_i32(int v, int srclane) { return __builtin_amdgcn_readlane(v, srclane); }
#define BCASTI(x, src) dwbc::readlane_i32((x), (src))
#else
#define BCASTI(x, src) ((x)[(src)])
#endif

// reciprocal without the IEEE division fix-up sequence: hardware estimate + two Newton steps (rel. error ~1e-16 for
// normal, non-huge arguments -- the pivots and norms it is used on)
DWBC_WDEV double fast_rcp(double d) {
#ifdef DWBC_HOST_EMU
    return 1.0 / d;
#else
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, e, r);
#endif
}
DWBC_WDEV float fast_rcp(float d) {
#ifdef DWBC_HOST_EMU
    return 1.0f / d;
#else
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(r, e, r);
#endif
}

// 1/sqrt(d) for positive normal d: hardware estimate + two Newton steps
DWBC_WDEV double fast_rsqrt(double d) {
#ifdef DWBC_HOST_EMU
    return 1.0 / sqrt(d);
#else
    double r = __builtin_amdgcn_rsq(d);
    double h = 0.5 * d;
    r = __builtin_fma(r, __builtin_fma(-h * r, r, 0.5), r);
    r = __builtin_fma(r, __builtin_fma(-h * r, r, 0.5), r);
    return r;
#endif
}
DWBC_WDEV float fast_rsqrt(float d) {
#ifdef DWBC_HOST_EMU
    return 1.0f / sqrtf(d);
#else
    float r = __builtin_amdgcn_rsqf(d);
    const float h = 0.5f * d;
    return __builtin_fmaf(r, __builtin_fmaf(-h * r, r, 0.5f), r);
#endif
}
// sin and cos of one angle in the arithmetic type of the build
#ifndef DWBC_HOST_EMU
// (device; the host emulation keeps the C library's.)  sin and cos of a joint angle without the large-argument path of the library routine: quadrant by Cody-Waite reduction with a
// two-part pi/2 (exact products for |x| < 1e5 rad: the first part has 33 significant bits), then the Taylor polynomials on
// |r| <= pi/4 (sin to r^17, cos to r^16: truncation < 5e-17 relative; max error 1.3e-16 against long double over +-33 rad).
// 0.7 us of the 107 us cycle against ocml's sincos (which carries the Payne-Hanek branch)
DWBC_WDEV void sincos_r(double x, double *s, double *c) {
    const double kd = __builtin_rint(x * 6.36619772367581382433e-01);
    const int q = (int)kd & 3;
    double r = __builtin_fma(-kd, 1.57079632673412561417e+00, x);
    r = __builtin_fma(-kd, 6.07710050650619224932e-11, r);
    const double z = r * r;
    double ps = 2.81145725434552060e-15;                       // 1/17!
    ps = __builtin_fma(ps, z, -7.64716373181981641e-13);       // -1/15!
    ps = __builtin_fma(ps, z, 1.60590438368216133e-10);        // 1/13!
    ps = __builtin_fma(ps, z, -2.50521083854417202e-08);       // -1/11!
    ps = __builtin_fma(ps, z, 2.75573192239858925e-06);        // 1/9!
    ps = __builtin_fma(ps, z, -1.98412698412698413e-04);       // -1/7!
    ps = __builtin_fma(ps, z, 8.33333333333333322e-03);        // 1/5!
    ps = __builtin_fma(ps, z, -1.66666666666666657e-01);       // -1/3!
    const double sn = __builtin_fma(ps * z, r, r);
    double pc = 4.77947733238738525e-14;                       // 1/16!
    pc = __builtin_fma(pc, z, -1.14707455977297245e-11);       // -1/14!
    pc = __builtin_fma(pc, z, 2.08767569878680990e-09);        // 1/12!
    pc = __builtin_fma(pc, z, -2.75573192239858883e-07);       // -1/10!
    pc = __builtin_fma(pc, z, 2.48015873015873016e-05);        // 1/8!
    pc = __builtin_fma(pc, z, -1.38888888888888894e-03);       // -1/6!
    pc = __builtin_fma(pc, z, 4.16666666666666644e-02);        // 1/4!
    pc = __builtin_fma(pc, z, -0.5);
    const double cs = __builtin_fma(pc, z, 1.0);
    const double a = (q & 1) ? cs : sn, b = (q & 1) ? sn : cs;
    *s = (q & 2) ? -a : a;
    *c = ((q + 1) & 2) ? -b : b;
}
#else
DWBC_WDEV void sincos_r(double x, double *s, double *c) { sincos(x, s, c); }
#endif
DWBC_WDEV void sincos_r(float x, float *s, float *c) { sincosf(x, s, c); }

// element `lane` of a uniform 12-array (avoids dynamic register indexing on the device)
DWBC_WDEV real_t pick12(const real_t *a, int lane) {
    real_t v = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) v = (lane == i) ? a[i] : v;
    return v;
}

}  // namespace dwbc

// wave-wide arg-min over a per-lane (value, key): results are uniform.  Ties go to the smaller key.
#ifdef DWBC_HOST_EMU
#define WAVE_ARGMIN(val, key, out_v, out_k)                                                        \
    do {                                                                                           \
        out_v = (val)[0];                                                                          \
        out_k = (key)[0];                                                                          \
        for (int l_ = 1; l_ < 64; l_++)                                                            \
            if ((val)[l_] < out_v || ((val)[l_] == out_v && (key)[l_] < out_k)) { out_v = (val)[l_]; out_k = (key)[l_]; } \
    } while (0)
// lane of the smallest value after rounding to float (first such lane); the selection rules that use it (most violated
// row, QR pivot column) tolerate a 1e-7 relative tie-break
#define WAVE_ARGMIN_F32(val, out_lane)                                                             \
    do {                                                                                           \
        float m_ = (float)(val)[0];                                                                \
        out_lane = 0;                                                                              \
        for (int l_ = 1; l_ < 64; l_++)                                                            \
            if ((float)(val)[l_] < m_) { m_ = (float)(val)[l_]; out_lane = l_; }                   \
    } while (0)
// inclusive prefix sum over the lanes of element j of a per-lane array (in place)
#define WAVE_PREFIX_A(arr, j)                                                                      \
    do {                                                                                           \
        for (int l_ = 1; l_ < 64; l_++) (arr)[l_][(j)] += (arr)[l_ - 1][(j)];                      \
    } while (0)
// sum of a per-lane value over the 64 lanes (uniform result)
#define WAVE_SUM(val, out_v)                                                                       \
    do {                                                                                           \
        out_v = (val)[0];                                                                          \
        for (int l_ = 1; l_ < 64; l_++) out_v += (val)[l_];                                        \
    } while (0)
// exact arg-min over lanes 16..31 only
#define WAVE_ARGMIN_ROW1(val, key, out_v, out_k)                                                   \
    do {                                                                                           \
        out_v = (val)[16];                                                                         \
        out_k = (key)[16];                                                                         \
        for (int l_ = 17; l_ < 32; l_++)                                                           \
            if ((val)[l_] < out_v || ((val)[l_] == out_v && (key)[l_] < out_k)) { out_v = (val)[l_]; out_k = (key)[l_]; } \
    } while (0)
#else
namespace dwbc {
// min over the wave of a 64-bit unsigned key with DPP row shifts / row broadcasts (gfx9 family), result uniform
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long x) {
#define DWBC_DPP_STEP(ctrl, rmask)                                                                         \
    {                                                                                                      \
        const int lo_ = (int)(unsigned)(x & 0xffffffffull), hi_ = (int)(unsigned)(x >> 32);                \
        const int olo_ = __builtin_amdgcn_update_dpp(lo_, lo_, ctrl, rmask, 0xf, false);                   \
        const int ohi_ = __builtin_amdgcn_update_dpp(hi_, hi_, ctrl, rmask, 0xf, false);                   \
        const unsigned long long o_ = ((unsigned long long)(unsigned)ohi_ << 32) | (unsigned)olo_;         \
        x = o_ < x ? o_ : x;                                                                               \
    }
    DWBC_DPP_STEP(0x111, 0xf)  // row_shr:1
    DWBC_DPP_STEP(0x112, 0xf)  // row_shr:2
    DWBC_DPP_STEP(0x114, 0xf)  // row_shr:4
    DWBC_DPP_STEP(0x118, 0xf)  // row_shr:8   -> lane 15 of each row holds the row minimum
    DWBC_DPP_STEP(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
    DWBC_DPP_STEP(0x143, 0xc)  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave minimum
#undef DWBC_DPP_STEP
    const int lo = __builtin_amdgcn_readlane((int)(unsigned)(x & 0xffffffffull), 63);
    const int hi = __builtin_amdgcn_readlane((int)(unsigned)(x >> 32), 63);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
// same reduction restricted to DPP row 1 (lanes 16..31); the result is read from lane 31
__device__ __forceinline__ unsigned long long row1_min_u64(unsigned long long x) {
#define DWBC_DPP_STEP(ctrl)                                                                                \
    {                                                                                                      \
        const int lo_ = (int)(unsigned)(x & 0xffffffffull), hi_ = (int)(unsigned)(x >> 32);                \
        const int olo_ = __builtin_amdgcn_update_dpp(lo_, lo_, ctrl, 0xf, 0xf, false);                     \
        const int ohi_ = __builtin_amdgcn_update_dpp(hi_, hi_, ctrl, 0xf, 0xf, false);                     \
        const unsigned long long o_ = ((unsigned long long)(unsigned)ohi_ << 32) | (unsigned)olo_;         \
        x = o_ < x ? o_ : x;                                                                               \
    }
    DWBC_DPP_STEP(0x111)
    DWBC_DPP_STEP(0x112)
    DWBC_DPP_STEP(0x114)
    DWBC_DPP_STEP(0x118)
#undef DWBC_DPP_STEP
    const int lo = __builtin_amdgcn_readlane((int)(unsigned)(x & 0xffffffffull), 31);
    const int hi = __builtin_amdgcn_readlane((int)(unsigned)(x >> 32), 31);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
// lane of the wave minimum of a value rounded to float: six single-register DPP steps and a ballot
__device__ __forceinline__ int wave_argmin_f32(real_t v) {
    const float f = (float)v;
    float m = f;
#define DWBC_DPP_STEP(ctrl, rmask)                                                                         \
    {                                                                                                      \
        const int o_ = __builtin_amdgcn_update_dpp(__float_as_int(m), __float_as_int(m), ctrl, rmask, 0xf, false); \
        m = fminf(m, __int_as_float(o_));                                                                  \
    }
    DWBC_DPP_STEP(0x111, 0xf)
    DWBC_DPP_STEP(0x112, 0xf)
    DWBC_DPP_STEP(0x114, 0xf)
    DWBC_DPP_STEP(0x118, 0xf)
    DWBC_DPP_STEP(0x142, 0xa)
    DWBC_DPP_STEP(0x143, 0xc)
#undef DWBC_DPP_STEP
    const float mall = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63));
    const unsigned long long b = __ballot(f == mall);
    return b ? (int)__builtin_ctzll(b) : 0;
}
// sum over the wave of a per-lane value (DPP row shifts + row broadcasts; lanes without a source add zero), result uniform
__device__ __forceinline__ double wave_sum_f64(double x) {
#define DWBC_DPP_STEP(ctrl, rmask)                                                                         \
    {                                                                                                      \
        const long long b_ = __double_as_longlong(x);                                                      \
        const int lo_ = (int)(unsigned)(b_ & 0xffffffffll), hi_ = (int)(unsigned)((unsigned long long)b_ >> 32); \
        const int olo_ = __builtin_amdgcn_update_dpp(0, lo_, ctrl, rmask, 0xf, true);                      \
        const int ohi_ = __builtin_amdgcn_update_dpp(0, hi_, ctrl, rmask, 0xf, true);                      \
        x += __longlong_as_double((long long)(((unsigned long long)(unsigned)ohi_ << 32) | (unsigned)olo_)); \
    }
    DWBC_DPP_STEP(0x111, 0xf)
    DWBC_DPP_STEP(0x112, 0xf)
    DWBC_DPP_STEP(0x114, 0xf)
    DWBC_DPP_STEP(0x118, 0xf)  // lane 15 of each row: the row sum
    DWBC_DPP_STEP(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
    DWBC_DPP_STEP(0x143, 0xc)  // row_bcast:31 into rows 2 and 3 -> lane 63: the wave sum
#undef DWBC_DPP_STEP
    return readlane_f64(x, 63);
}
// inclusive prefix sum over the lanes (Hillis-Steele inside each DPP row of 16, then the row totals: row_bcast:15 into rows 1 and 3,
// row_bcast:31 into rows 2 and 3)
__device__ __forceinline__ double wave_prefix_f64(double x) {
#define DWBC_DPP_STEP(ctrl, rmask)                                                                         \
    {                                                                                                      \
        const long long b_ = __double_as_longlong(x);                                                      \
        const int lo_ = (int)(unsigned)(b_ & 0xffffffffll), hi_ = (int)(unsigned)((unsigned long long)b_ >> 32); \
        const int olo_ = __builtin_amdgcn_update_dpp(0, lo_, ctrl, rmask, 0xf, true);                      \
        const int ohi_ = __builtin_amdgcn_update_dpp(0, hi_, ctrl, rmask, 0xf, true);                      \
        x += __longlong_as_double((long long)(((unsigned long long)(unsigned)ohi_ << 32) | (unsigned)olo_)); \
    }
    DWBC_DPP_STEP(0x111, 0xf)
    DWBC_DPP_STEP(0x112, 0xf)
    DWBC_DPP_STEP(0x114, 0xf)
    DWBC_DPP_STEP(0x118, 0xf)
    DWBC_DPP_STEP(0x142, 0xa)
    DWBC_DPP_STEP(0x143, 0xc)
#undef DWBC_DPP_STEP
    return x;
}
__device__ __forceinline__ float wave_prefix_f64(float x) {
#define DWBC_DPP_STEP(ctrl, rmask) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rmask, 0xf, true));
    DWBC_DPP_STEP(0x111, 0xf)
    DWBC_DPP_STEP(0x112, 0xf)
    DWBC_DPP_STEP(0x114, 0xf)
    DWBC_DPP_STEP(0x118, 0xf)
    DWBC_DPP_STEP(0x142, 0xa)
    DWBC_DPP_STEP(0x143, 0xc)
#undef DWBC_DPP_STEP
    return x;
}
__device__ __forceinline__ float wave_sum_f64(float x) {
#define DWBC_DPP_STEP(ctrl, rmask) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rmask, 0xf, true));
    DWBC_DPP_STEP(0x111, 0xf)
    DWBC_DPP_STEP(0x112, 0xf)
    DWBC_DPP_STEP(0x114, 0xf)
    DWBC_DPP_STEP(0x118, 0xf)
    DWBC_DPP_STEP(0x142, 0xa)
    DWBC_DPP_STEP(0x143, 0xc)
#undef DWBC_DPP_STEP
    return readlane_f64(x, 63);
}
// order-preserving map double -> u64, low 7 bits replaced by the lane so that keys are unique
__device__ __forceinline__ unsigned long long argmin_key(real_t v, int lane) {
    unsigned long long b = (unsigned long long)__double_as_longlong((double)v);
    b = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    return (b & ~127ull) | (unsigned)lane;
}
}  // namespace dwbc
// values are compared with their 7 lowest mantissa bits dropped (1.4e-14 relative); the winner's exact value is returned
#define WAVE_ARGMIN(val, key, out_v, out_k)                                                        \
    do {                                                                                           \
        const unsigned long long m_ = dwbc::wave_min_u64(dwbc::argmin_key((val), lane));           \
        const int wl_ = (int)(m_ & 127ull);                                                        \
        out_v = dwbc::readlane_f64((val), wl_);                                                    \
        out_k = dwbc::readlane_i32((key), wl_);                                                    \
    } while (0)
#define WAVE_ARGMIN_F32(val, out_lane) out_lane = dwbc::wave_argmin_f32(val)
#define WAVE_SUM(val, out_v) out_v = dwbc::wave_sum_f64(val)
#define WAVE_PREFIX_A(arr, j) (arr)[(j)] = dwbc::wave_prefix_f64((arr)[(j)])
#define WAVE_ARGMIN_ROW1(val, key, out_v, out_k)                                                   \
    do {                                                                                           \
        const unsigned long long m_ = dwbc::row1_min_u64(dwbc::argmin_key((val), lane));           \
        const int wl_ = (int)(m_ & 127ull);                                                        \
        out_v = dwbc::readlane_f64((val), wl_);                                                    \
        out_k = dwbc::readlane_i32((key), wl_);                                                    \
    } while (0)
#endif

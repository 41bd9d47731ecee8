// Issue-cost table of single instructions for ONE wave per SIMD (the regime of BASELINE configs[1]: 1024 instances on 1024
// SIMDs): shader cycles (s_memtime) per instruction of a 32-instruction straight-line group, loop overhead included (< 0.3).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench/ubench3.hip -o /tmp/ubench3 && /tmp/ubench3
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void k(const double* in, double* out, unsigned long long* cyc, int reps) {
    __shared__ __attribute__((aligned(16))) double buf[128];
    const int lane = threadIdx.x;
    double a0 = in[lane], a1 = in[64 + lane], x = a0 * 1e-3, y = 0.999;
    int i0 = lane, i1 = lane + 1, i2 = lane + 2, i3 = lane + 3;
    buf[lane] = x; buf[64 + lane] = y;
    __syncthreads();
    const unsigned lb = (unsigned)(size_t)buf;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int r = 0; r < reps; r++) {
        if constexpr (MODE == 0) asm volatile(".rept 8\n v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n .endr" : "=&v"(i0), "=&v"(i1), "=&v"(i2), "=&v"(i3) : "v"(lane), "v"(reps) : "vcc");
        if constexpr (MODE == 1) asm volatile(".rept 8\n v_cndmask_b32_e64 %0, %4, %5, s[20:21]\n v_cndmask_b32_e64 %1, %4, %5, s[20:21]\n v_cndmask_b32_e64 %2, %4, %5, s[20:21]\n v_cndmask_b32_e64 %3, %4, %5, s[20:21]\n .endr" : "=&v"(i0), "=&v"(i1), "=&v"(i2), "=&v"(i3) : "v"(lane), "v"(reps) : "s20", "s21");
        if constexpr (MODE == 2) asm volatile(".rept 32\n v_cmp_eq_u32 vcc, %0, %1\n .endr" ::"v"(lane), "v"(reps) : "vcc");
        if constexpr (MODE == 3) asm volatile(".rept 32\n v_cmp_eq_u32_e64 s[20:21], %0, %1\n .endr" ::"v"(lane), "v"(reps) : "s20", "s21");
        if constexpr (MODE == 4) asm volatile(".rept 16\n v_cmp_eq_u32 vcc, %1, %2\n s_nop 1\n v_cndmask_b32 %0, %1, %2, vcc\n .endr" : "=&v"(i0) : "v"(lane), "v"(reps) : "vcc");  // 48 instructions
        if constexpr (MODE == 5) asm volatile(".rept 32\n v_rcp_f64 %0, %1\n .endr" : "=&v"(a0) : "v"(x));
        if constexpr (MODE == 6) asm volatile(".rept 32\n v_xor_b32 %0, %1, %2\n .endr" : "=&v"(i0) : "v"(lane), "v"(reps));
        if constexpr (MODE == 7) asm volatile(".rept 32\n s_and_b64 s[20:21], s[22:23], exec\n .endr" ::: "s20", "s21", "s22", "s23", "scc");
        if constexpr (MODE == 8) asm volatile(".rept 32\n s_mov_b32 s20, s21\n .endr" ::: "s20", "s21");
        if constexpr (MODE == 9) asm volatile(".rept 32\n s_waitcnt lgkmcnt(0)\n .endr");
        if constexpr (MODE == 10) asm volatile(".rept 32\n ds_read_b128 v[100:103], %0\n .endr\n s_waitcnt lgkmcnt(0)" ::"v"(lb) : "v100", "v101", "v102", "v103");
        if constexpr (MODE == 11) asm volatile(".rept 32\n ds_read_b64 v[100:101], %0\n .endr\n s_waitcnt lgkmcnt(0)" ::"v"(lb) : "v100", "v101");
        if constexpr (MODE == 12) asm volatile(".rept 32\n ds_write_b64 %0, %1\n .endr\n s_waitcnt lgkmcnt(0)" ::"v"(lb + 8 * lane), "v"(x) : "memory");
        if constexpr (MODE == 13) asm volatile(".rept 32\n v_mov_b64 %0, %1\n .endr" : "=&v"(a0) : "v"(x));
        if constexpr (MODE == 14) asm volatile(".rept 32\n v_readfirstlane_b32 s20, %0\n .endr" ::"v"(lane) : "s20");
        if constexpr (MODE == 15) asm volatile(".rept 32\n s_cmp_eq_u32 s20, 0\n .endr" ::: "s20", "scc");
        if constexpr (MODE == 16) asm volatile(".rept 16\n s_cmp_eq_u32 s20, 12345\n s_cbranch_scc1 1f\n .endr\n 1:" ::: "s20", "scc");  // never taken
        if constexpr (MODE == 17) asm volatile(".rept 32\n ds_read_b128 v[100:103], %0\n s_waitcnt lgkmcnt(0)\n .endr" ::"v"(lb) : "v100", "v101", "v102", "v103");  // dependent LDS round trip
        if constexpr (MODE == 18) asm volatile(".rept 32\n v_fma_f64 %0, %2, %3, %0\n v_mov_b32 %1, %4\n .endr" : "+v"(a0), "=&v"(i1) : "v"(x), "v"(y), "v"(lane));  // 64 instructions, alternating
        if constexpr (MODE == 19) asm volatile(".rept 32\n v_fma_f64 %0, %1, %2, %0\n s_mov_b32 s20, s21\n .endr" : "+v"(a0) : "v"(x), "v"(y) : "s20", "s21");  // VALU + SALU alternating (64)
        if constexpr (MODE == 20) asm volatile(".rept 32\n v_fma_f64 %0, %1, %2, %0\n ds_read_b128 v[100:103], %3\n .endr\n s_waitcnt lgkmcnt(0)" : "+v"(a0) : "v"(x), "v"(y), "v"(lb) : "v100", "v101", "v102", "v103");  // VALU + LDS alternating (64)
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + lane] = a0 + a1 + i0 + i1 + i2 + i3;
}

int main() {
    const int G = 1024, reps = 4000;
    double *din, *dout; unsigned long long* dc;
    CK(hipMalloc(&din, 1 << 16)); CK(hipMalloc(&dout, G * 64 * 8)); CK(hipMalloc(&dc, G * 8));
    std::vector<double> h(256);
    for (int l = 0; l < 256; l++) h[l] = 1.0 + 1e-3 * ((l * 37) % 101);
    CK(hipMemcpy(din, h.data(), 256 * 8, hipMemcpyHostToDevice));
    typedef void (*fn_t)(const double*, double*, unsigned long long*, int);
    const fn_t fn[] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>, k<12>, k<13>, k<14>, k<15>, k<16>, k<17>, k<18>, k<19>, k<20>};
    const char* nm[] = {"v_cndmask_b32 (vcc), 4 dests", "v_cndmask_b32_e64 (sgpr pair), 4 dests", "v_cmp_eq_u32 -> vcc", "v_cmp_eq_u32_e64 -> sgpr pair",
                        "v_cmp + s_nop 1 + v_cndmask (per triple)", "v_rcp_f64", "v_xor_b32", "s_and_b64", "s_mov_b32", "s_waitcnt lgkmcnt(0), nothing outstanding",
                        "ds_read_b128 broadcast, issue only", "ds_read_b64 broadcast, issue only", "ds_write_b64, issue only", "v_mov_b64", "v_readfirstlane_b32", "s_cmp_eq_u32",
                        "s_cmp + s_cbranch not taken (per pair)", "ds_read_b128 + wait: dependent LDS round trip", "v_fma_f64 + v_mov_b32 alternating (per pair)",
                        "v_fma_f64 + s_mov_b32 alternating (per pair)", "v_fma_f64 + ds_read_b128 alternating (per pair)"};
    const int per[] = {32, 32, 32, 32, 16, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 16, 32, 32, 32, 32};
    std::vector<unsigned long long> c(G);
    for (int m = 0; m < 21; m++) {
        for (int i = 0; i < 3; i++) fn[m]<<<G, 64>>>(din, dout, dc, reps);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(c.data(), dc, G * 8, hipMemcpyDeviceToHost));
        std::sort(c.begin(), c.end());
        printf("%-52s %7.2f cycles\n", nm[m], (double)c[G / 2] / ((double)reps * per[m]));
    }
    return 0;
}

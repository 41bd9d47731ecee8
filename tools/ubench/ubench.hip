// Micro-benchmarks for the building blocks of the cycle kernel (run on the GPU box):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I libdwbc_amd/csrc -I include tools/ubench/ubench.hip -o gpurun_out/ubench && gpurun_out/ubench
// Every test runs 1024 single-wave workgroups (one wave per SIMD, like the production launch) and reports the median
// wave time in shader clocks.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include "dwbc_types.h"
#include "dwbc_cycle.h"
#include "dwbc_qp_wave.h"
#include "dwbc_cycle2.h"
using namespace dwbc;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(64) void k_small_inv(const double* in, double* out, long long* cyc, int n, int reps) {
    __shared__ double A[144], O[144], cb[64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 144; i += 64) A[i] = in[i];
    __syncthreads();
    long long t0 = clock64();
    int ok = 1;
    for (int r = 0; r < reps; r++) ok &= spd_inverse_small(A, 12, n, O, 12, cb);
    long long t1 = clock64();
    if (lane == 0) cyc[blockIdx.x] = (t1 - t0) / reps;
    for (int i = lane; i < 144; i += 64) out[blockIdx.x * 144 + i] = O[i] + ok;
}

// dependent / independent FMA chains fed by v_readlane or by LDS broadcast
__global__ __launch_bounds__(64) void k_feed(const double* in, double* out, long long* cyc, int mode, int reps) {
    __shared__ double buf[64];
    const int lane = threadIdx.x;
    double a[8], x = in[lane];
    for (int i = 0; i < 8; i++) a[i] = in[64 + lane + i];
    buf[lane] = x;
    __syncthreads();
    long long t0 = clock64();
    for (int r = 0; r < reps; r++) {
        if (mode == 0) {          // 32 FMAs, operands via readlane, 8 independent accumulators
#pragma unroll
            for (int j = 0; j < 32; j++) a[j & 7] += readlane_f64(x, j) * a[(j + 1) & 7];
        } else if (mode == 1) {   // 32 FMAs, operands via LDS broadcast
#pragma unroll
            for (int j = 0; j < 32; j++) a[j & 7] += buf[j] * a[(j + 1) & 7];
        } else if (mode == 2) {   // 32 FMAs, register operands only, 8 independent accumulators
#pragma unroll
            for (int j = 0; j < 32; j++) a[j & 7] += x * a[(j + 1) & 7];
        } else if (mode == 4) {   // 32 FMAs, operands via ds_bpermute (per-lane source lane), 8 independent accumulators
#pragma unroll
            for (int j = 0; j < 32; j++) a[j & 7] += __shfl(x, (lane + j) & 63, 64) * a[(j + 1) & 7];
        } else {                  // 32 FMAs, one dependent chain
#pragma unroll
            for (int j = 0; j < 32; j++) a[0] += x * a[0];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    long long t1 = clock64();
    if (lane == 0) cyc[blockIdx.x] = (t1 - t0);
    double acc = 0;
    for (int i = 0; i < 8; i++) acc += a[i];
    out[blockIdx.x * 64 + lane] = acc;
}

typedef double d4 __attribute__((ext_vector_type(4)));
// MFMA f64 16x16x4: layout check (D = A B with asymmetric integer data) and issue rate (nacc independent accumulators)
__global__ __launch_bounds__(64) void k_mfma(const double* Am, const double* Bm, double* D, long long* cyc, int nacc, int reps) {
    const int lane = threadIdx.x;
    const double a = Am[(lane & 15) * 4 + (lane >> 4)];   // A[i = lane&15][k = lane>>4], A is 16 x 4 row-major
    const double b = Bm[(lane >> 4) * 16 + (lane & 15)];  // B[k = lane>>4][j = lane&15], B is 4 x 16 row-major
    d4 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = d4{0, 0, 0, 0};
    long long t0 = clock64();
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (i < nacc) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    if (lane == 0) cyc[blockIdx.x] = (t1 - t0);
    if (blockIdx.x == 0)
        for (int r = 0; r < 4; r++) D[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[0][r];  // row = (lane>>4) + 4r, col = lane&15
}

static long long median(std::vector<long long> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    const int G = 1024;
    double *din, *dout; long long* dcyc;
    CK(hipMalloc(&din, 1 << 20)); CK(hipMalloc(&dout, 8 * G * 144 * 8)); CK(hipMalloc(&dcyc, 8 * G * 8));  // sized for the 8-waves/SIMD runs
    std::vector<double> h(1 << 17, 0.0);
    std::vector<long long> c(G);
    // SPD 12x12
    for (int i = 0; i < 12; i++) for (int j = 0; j < 12; j++) h[i * 12 + j] = (i == j ? 5.0 : 0.0) + 1.0 / (1 + i + j);
    CK(hipMemcpy(din, h.data(), 144 * 8, hipMemcpyHostToDevice));
    for (int n : {6, 12}) {
        k_small_inv<<<G, 64>>>(din, dout, dcyc, n, 8); CK(hipDeviceSynchronize());
        CK(hipMemcpy(c.data(), dcyc, G * 8, hipMemcpyDeviceToHost));
        printf("spd_inverse_small n=%d: %lld cycles\n", n, median(c));
    }
    for (int l = 0; l < 128; l++) h[l] = 1.0 + 1e-9 * l;
    CK(hipMemcpy(din, h.data(), 128 * 8, hipMemcpyHostToDevice));
    const char* nm[] = {"readlane-fed, 8 acc", "LDS-broadcast-fed, 8 acc", "register, 8 acc", "register, 1 dependent chain", "ds_bpermute-fed, 8 acc"};
    for (int mode : {0, 1, 2, 3, 4}) {
        k_feed<<<G, 64>>>(din, dout, dcyc, mode, 64); CK(hipDeviceSynchronize());
        CK(hipMemcpy(c.data(), dcyc, G * 8, hipMemcpyDeviceToHost));
        printf("32 fp64 FMAs (%s): %.1f cycles per FMA\n", nm[mode], median(c) / (64.0 * 32));
    }
    // MFMA
    std::vector<double> A(64), B(64), Dh(256), Dref(256, 0.0);
    for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) A[i * 4 + k] = 1 + i + 17 * k;
    for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) B[k * 16 + j] = 2 + 3 * j + 101 * k + (j * j) % 7;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) for (int k = 0; k < 4; k++) Dref[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD;
    CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dD, 256 * 8));
    CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
    k_mfma<<<G, 64>>>(dA, dB, dD, dcyc, 1, 1); CK(hipDeviceSynchronize());
    CK(hipMemcpy(Dh.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
    double err = 0; for (int i = 0; i < 256; i++) err = std::max(err, std::abs(Dh[i] - Dref[i]));
    printf("mfma_f64_16x16x4 layout check: max err %.3g\n", err);
    for (int nacc : {1, 2, 4}) {
        k_mfma<<<G, 64>>>(dA, dB, dD, dcyc, nacc, 256); CK(hipDeviceSynchronize());
        CK(hipMemcpy(c.data(), dcyc, G * 8, hipMemcpyDeviceToHost));
        printf("mfma_f64_16x16x4, %d independent accumulators: %.1f cycles per MFMA\n", nacc, median(c) / (256.0 * nacc));
    }
    // throughput vs waves per SIMD: same per-wave work, G = 1024 * w single-wave workgroups
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w : {1, 2, 3, 4, 8}) {  // buffers above are sized for w <= 8
        float ms_f = 0, ms_m = 0;
        k_feed<<<G * w, 64>>>(din, dout, dcyc, 2, 1024); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); k_feed<<<G * w, 64>>>(din, dout, dcyc, 2, 1024); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_f, e0, e1));
        k_mfma<<<G * w, 64>>>(dA, dB, dD, dcyc, 4, 1024); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); k_mfma<<<G * w, 64>>>(dA, dB, dD, dcyc, 4, 1024); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_m, e0, e1));
        const double fl_f = 2.0 * 64 * 32 * 1024 * G * w, fl_m = 2.0 * 1024 * 4 * 1024 * G * w;
        printf("%d wave(s)/SIMD: VALU fp64 FMA %.1f TFLOP/s, MFMA f64 16x16x4 %.1f TFLOP/s\n", w, fl_f / ms_f * 1e-9, fl_m / ms_m * 1e-9);
    }
    return 0;
}

// One stage of the cycle kernel built both ways ("build, don't estimate, one MFMA stage"):
//     A^-1 N_c = A^-1 - Y^T Jbar       (39 x 39, rank-12 update; dwbc_cycle2_stage1.inc, "Jbar^T, A^-1 N_c update")
//   VALU   the kernel's form: lane j owns column j in registers, the 12 values of row i of Y^T arrive by broadcast ds_read_b128,
//          12 FMAs per row against the lane's own 12 values of Jbar[:, j]
//   MFMA   v_mfma_f64_16x16x4_f64 on the matrix padded to 48 x 48: 3 x 3 tiles, K = 12 = 3 steps; operands read from LDS in the
//          MFMA lane layout (A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15]); accumulator tile D[row = (l >> 4) + 4 r][col = l & 15]
//   MFMA+  the same plus what it costs to use it inside the kernel: the matrix goes from the column-per-lane registers to the
//          tile layout and back through LDS (the sweeps before and after this stage need column-per-lane)
// 1024 single-wave workgroups, shader cycles per stage (s_memtime), max difference from the VALU result.  Run it under
//   rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES -- /tmp/mfma_stage
// for the matrix-pipe counters of each kernel.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench/mfma_stage.hip -o /tmp/mfma_stage
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int N = 39, C = 12, NP = 48;
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// inputs: S (N x N symmetric, row-major), Yt (N x C), Jb (C x N).  out: N x N
__global__ __launch_bounds__(64) void k_valu(const double *S, const double *Yt, const double *Jb, double *out, unsigned long long *cyc, int reps) {
    __shared__ __attribute__((aligned(16))) double lY[NP * C];
    const int lane = threadIdx.x;
    for (int i = lane; i < NP * C; i += 64) lY[i] = i < N * C ? Yt[i] : 0.0;
    double s[N], jb[C];
#pragma unroll
    for (int i = 0; i < N; i++) s[i] = lane < N ? S[i * N + lane] : 0.0;
#pragma unroll
    for (int p = 0; p < C; p++) jb[p] = lane < N ? Jb[p * N + lane] : 0.0;
    __syncthreads();
    const unsigned long long t0 = now();
    for (int r = 0; r < reps; r++) {
        asm volatile("" ::: "memory");  // the operands are re-read from LDS every time, as in the kernel
#pragma unroll
        for (int i = 0; i < N; i++) {
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int p = 0; p < C; p += 2) { a0 += lY[i * C + p] * jb[p]; a1 += lY[i * C + p + 1] * jb[p + 1]; }
            s[i] -= a0 + a1;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = now();
    if (lane == 0) cyc[blockIdx.x] = (t1 - t0) / reps;
    if (blockIdx.x == 0 && lane < N)
#pragma unroll
        for (int i = 0; i < N; i++) out[i * N + lane] = s[i];
}

// MODE 0: matrix lives in the tile layout (no conversion);  MODE 1: column-per-lane registers -> LDS -> tiles -> MFMA -> LDS -> registers
template <int MODE>
__global__ __launch_bounds__(64) void k_mfma(const double *S, const double *Yt, const double *Jb, double *out, unsigned long long *cyc, int reps) {
    __shared__ __attribute__((aligned(16))) double lY[NP * C], lJ[C * NP], lS[NP * NP];
    const int lane = threadIdx.x;
    for (int i = lane; i < NP * C; i += 64) lY[i] = i < N * C ? Yt[i] : 0.0;
    for (int i = lane; i < C * NP; i += 64) { const int p = i / NP, j = i % NP; lJ[i] = j < N ? Jb[p * N + j] : 0.0; }
    for (int i = lane; i < NP * NP; i += 64) { const int a = i / NP, b = i % NP; lS[i] = (a < N && b < N) ? S[a * N + b] : 0.0; }
    double s[N];
#pragma unroll
    for (int i = 0; i < N; i++) s[i] = lane < N ? S[i * N + lane] : 0.0;
    __syncthreads();
    d4 acc[3][3];
    const int li = lane & 15, lk = lane >> 4;
    if (MODE == 0) {
#pragma unroll
        for (int ti = 0; ti < 3; ti++)
#pragma unroll
            for (int tj = 0; tj < 3; tj++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[ti][tj][r] = lS[(16 * ti + lk + 4 * r) * NP + 16 * tj + li];
    }
    const unsigned long long t0 = now();
    for (int rep = 0; rep < reps; rep++) {
        asm volatile("" ::: "memory");
        if (MODE == 1) {
            // column-per-lane registers -> LDS (row-major) -> accumulator tiles
            if (lane < N)
#pragma unroll
                for (int i = 0; i < N; i++) lS[i * NP + lane] = s[i];
            __syncthreads();
#pragma unroll
            for (int ti = 0; ti < 3; ti++)
#pragma unroll
                for (int tj = 0; tj < 3; tj++)
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[ti][tj][r] = lS[(16 * ti + lk + 4 * r) * NP + 16 * tj + li];
        }
        // D -= Y^T Jbar : A operand = -Y^T tile (16 x 4), B operand = Jbar tile (4 x 16)
#pragma unroll
        for (int ks = 0; ks < 3; ks++) {
            double a[3], b[3];
#pragma unroll
            for (int t = 0; t < 3; t++) {
                a[t] = -lY[(16 * t + li) * C + 4 * ks + lk];
                b[t] = lJ[(4 * ks + lk) * NP + 16 * t + li];
            }
#pragma unroll
            for (int ti = 0; ti < 3; ti++)
#pragma unroll
                for (int tj = 0; tj < 3; tj++) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
        }
        if (MODE == 1) {
            // tiles -> LDS -> column-per-lane registers
#pragma unroll
            for (int ti = 0; ti < 3; ti++)
#pragma unroll
                for (int tj = 0; tj < 3; tj++)
#pragma unroll
                    for (int r = 0; r < 4; r++) lS[(16 * ti + lk + 4 * r) * NP + 16 * tj + li] = acc[ti][tj][r];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < N; i++) s[i] = lS[i * NP + (lane < N ? lane : 0)];
            __syncthreads();
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = now();
    if (lane == 0) cyc[blockIdx.x] = (t1 - t0) / reps;
    if (blockIdx.x == 0) {
        if (MODE == 1) {
            if (lane < N)
#pragma unroll
                for (int i = 0; i < N; i++) out[i * N + lane] = s[i];
        } else {
#pragma unroll
            for (int ti = 0; ti < 3; ti++)
#pragma unroll
                for (int tj = 0; tj < 3; tj++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int row = 16 * ti + lk + 4 * r, col = 16 * tj + li;
                        if (row < N && col < N) out[row * N + col] = acc[ti][tj][r];
                    }
        }
    }
}

static unsigned long long median(std::vector<unsigned long long> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    const int G = 1024;
    std::vector<double> S(N * N), Yt(N * C), Jb(C * N), ref(N * N), got(N * N);
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) S[i * N + j] = (i == j ? 2.0 : 0.0) + 1.0 / (3.0 + i + j);
    for (int i = 0; i < N * C; i++) Yt[i] = 0.01 * ((i * 37) % 101) - 0.4;
    for (int i = 0; i < C * N; i++) Jb[i] = 0.02 * ((i * 53) % 89) - 0.7;
    double *dS, *dY, *dJ, *dout; unsigned long long *dc;
    CK(hipMalloc(&dS, N * N * 8)); CK(hipMalloc(&dY, N * C * 8)); CK(hipMalloc(&dJ, C * N * 8)); CK(hipMalloc(&dout, N * N * 8)); CK(hipMalloc(&dc, G * 8));
    CK(hipMemcpy(dS, S.data(), N * N * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dY, Yt.data(), N * C * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dJ, Jb.data(), C * N * 8, hipMemcpyHostToDevice));
    std::vector<unsigned long long> c(G);
    const char *nm[] = {"VALU, column per lane + broadcast LDS reads (the kernel's form)", "MFMA f64 16x16x4, matrix already in tile layout", "MFMA f64 16x16x4 incl. registers <-> tile layout through LDS"};
    for (int v = 0; v < 3; v++) {
        // reps = 1: the result of ONE update is compared; reps = 64: timing
        for (int pass = 0; pass < 2; pass++) {
            const int reps = pass == 0 ? 1 : 64;
            for (int it = 0; it < 2; it++) {
                if (v == 0) k_valu<<<G, 64>>>(dS, dY, dJ, dout, dc, reps);
                else if (v == 1) k_mfma<0><<<G, 64>>>(dS, dY, dJ, dout, dc, reps);
                else k_mfma<1><<<G, 64>>>(dS, dY, dJ, dout, dc, reps);
                CK(hipDeviceSynchronize());
            }
            if (pass == 0) {
                CK(hipMemcpy(got.data(), dout, N * N * 8, hipMemcpyDeviceToHost));
                if (v == 0) ref = got;
            } else {
                CK(hipMemcpy(c.data(), dc, G * 8, hipMemcpyDeviceToHost));
            }
        }
        double err = 0;
        for (int i = 0; i < N * N; i++) err = std::max(err, std::fabs(got[i] - ref[i]));
        printf("%-72s %7llu cycles per update | max |result - VALU| %.2e\n", nm[v], median(c), err);
    }
    printf("algorithmic FMAs: 39 x 39 x 12 = 18252; padded for MFMA: 48 x 48 x 12 = 27648 (27 instructions of 1024)\n");
    return 0;
}

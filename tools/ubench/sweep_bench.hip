// Prototype bench of the register-resident symmetric sweeps (the two largest single stages of the cycle kernel):
//   V0  the shipped sweep: pivot column broadcast with 2 v_readlane_b32 per double (sweep_inverse_tree, dwbc_cycle2.h)
//   V1  pivot column through LDS: by symmetry column K is row K across the lanes, so ONE ds_write_b64 per pivot publishes it
//       and broadcast ds_read_b128 feed the FMAs
//   V2  V1 software-pipelined: row K-1 is updated first and published while the rest of pivot K is applied
// Each variant inverts a seeded SPD matrix (33 x 33 dense = the W^+ sweep, 39 x 39 with the TOCABI tree pattern = the A^-1
// sweep) 2 * reps times in place (sweep o sweep = identity) on 1024 single-wave workgroups; reports shader cycles per sweep
// and the max difference of the result from V0.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I libdwbc_amd/csrc -I include tools/ubench/sweep_bench.hip -o /tmp/sweep_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>
#include "dwbc_types.h"
#include "dwbc_cycle.h"
#include "dwbc_qp_wave.h"
#include "dwbc_cycle2.h"
using namespace dwbc;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// ---- V1 / V2: LDS-fed pivots.  cb: 2 x 64 doubles; column K is read from buffer (K & 1)
template <class Topo, int NN, int K, bool PIPE>
__device__ __forceinline__ void lds_pivot(double (&s)[NN], double &dg, int &ok, double *cb) {
    const int lane = threadIdx.x;
    constexpr unsigned long long rel = Topo::relatives(K);
    double *col = cb + (K & 1) * 64;
    if (!PIPE) {
        col[lane] = s[K];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    double d = readlane_f64(dg, K);
    int pos = d > 0.0 ? 1 : 0;
    asm volatile("" : "+v"(pos));
    ok &= pos;
    if (!(d > 0.0)) d = 1.0;
    const double rp = fast_rcp(d);
    int lk = lane;
    asm volatile("" : "+v"(lk));
    const bool piv = lk == K;
    const double cj = s[K];
    const double h = piv ? (1.0 - rp) : cj * rp;
    if (PIPE && K > 0) {
        if ((rel >> (K - 1)) & 1ull) s[K - 1] -= col[K - 1] * h;
        cb[((K - 1) & 1) * 64 + lane] = s[K - 1];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
#pragma unroll
    for (int i = 0; i < NN; i++) {
        if (((rel >> i) & 1ull) && !(PIPE && K > 0 && i == K - 1)) s[i] -= col[i] * h;
    }
    dg = piv ? -rp : dg - cj * h;
    if constexpr (K > 0) lds_pivot<Topo, NN, K - 1, PIPE>(s, dg, ok, cb);
}

template <class Topo, int NN, bool PIPE>
__device__ __forceinline__ int sweep_lds(double (&s)[NN], double &dg, double *cb) {
    const int lane = threadIdx.x;
    int ok = 1;
    {
        int lp = lane;
        asm volatile("" : "+v"(lp));
#pragma unroll
        for (int i = 0; i < NN; i++) s[i] = (i == lp) ? dg - 1.0 : s[i];
    }
    if (PIPE) {
        cb[((NN - 1) & 1) * 64 + lane] = s[NN - 1];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    lds_pivot<Topo, NN, NN - 1, PIPE>(s, dg, ok, cb);
    ok = __builtin_amdgcn_readfirstlane(ok);
    {
        int le = lane;
        asm volatile("" : "+v"(le));
#pragma unroll
        for (int i = 0; i < NN; i++) s[i] = (i == le) ? -dg : -s[i];
        dg = -dg;
    }
    return ok;
}


// ---- V3 / V4: the same through hand-issued LDS traffic.  The compiler keeps only 2-3 broadcast reads in flight (V1 / V2:
// every FMA pair waits an LDS round trip); here ALL reads of a column are issued back to back into their own registers, one
// wait, then the FMAs.  V4 double-buffers registers and LDS: the reads of column K-1 are issued inside pivot K, right after
// row K-1 has been updated and published, and are consumed a whole pivot later.
typedef double d2v __attribute__((ext_vector_type(2)));
template <int P, int NP>
__device__ __forceinline__ void col_issue(d2v (&c)[NP], unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c[P]) : "v"(addr), "n"(16 * P));
    if constexpr (P + 1 < NP) col_issue<P + 1, NP>(c, addr);
}
template <int NP>
__device__ __forceinline__ void col_wait(d2v (&c)[NP]) {
    static_assert(NP <= 20, "operand count of one asm statement");
    if constexpr (NP == 17)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]),
                     "+v"(c[9]), "+v"(c[10]), "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), "+v"(c[14]), "+v"(c[15]), "+v"(c[16]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]),
                     "+v"(c[9]), "+v"(c[10]), "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), "+v"(c[14]), "+v"(c[15]), "+v"(c[16]), "+v"(c[17]), "+v"(c[18]), "+v"(c[19]));
}
__device__ __forceinline__ void lds_put(unsigned addr, double v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }

template <class Topo, int NN, int K, bool PIPE>
__device__ __forceinline__ void asm_pivot(double (&s)[NN], double &dg, int &ok, unsigned cb, d2v (&cur)[(NN + 1) / 2], d2v (&nxt)[(NN + 1) / 2]) {
    constexpr int NP = (NN + 1) / 2;
    const int lane = threadIdx.x;
    constexpr unsigned long long rel = Topo::relatives(K);
    if (!PIPE) {
        lds_put(cb + 8 * lane, s[K]);
        col_issue<0, NP>(cur, cb);
    }
    double d = readlane_f64(dg, K);
    int pos = d > 0.0 ? 1 : 0;
    asm volatile("" : "+v"(pos));
    ok &= pos;
    if (!(d > 0.0)) d = 1.0;
    const double rp = fast_rcp(d);
    int lk = lane;
    asm volatile("" : "+v"(lk));
    const bool piv = lk == K;
    const double cj = s[K];
    const double h = piv ? (1.0 - rp) : cj * rp;
    col_wait<NP>(cur);
    if (PIPE && K > 0) {
        if ((rel >> (K - 1)) & 1ull) s[K - 1] -= cur[(K - 1) / 2][(K - 1) & 1] * h;
        const unsigned nb = cb + (((K - 1) & 1) ? 512u : 0u);
        lds_put(nb + 8 * lane, s[K - 1]);
        col_issue<0, NP>(nxt, nb);
    }
#pragma unroll
    for (int i = 0; i < NN; i++) {
        if (((rel >> i) & 1ull) && !(PIPE && K > 0 && i == K - 1)) s[i] -= cur[i / 2][i & 1] * h;
    }
    dg = piv ? -rp : dg - cj * h;
    if constexpr (K > 0) asm_pivot<Topo, NN, K - 1, PIPE>(s, dg, ok, cb, nxt, cur);
}

template <class Topo, int NN, bool PIPE>
__device__ __forceinline__ int sweep_asm(double (&s)[NN], double &dg, double *cbp) {
    constexpr int NP = (NN + 1) / 2;
    const int lane = threadIdx.x;
    const unsigned cb = (unsigned)(size_t)cbp;
    int ok = 1;
    {
        int lp = lane;
        asm volatile("" : "+v"(lp));
#pragma unroll
        for (int i = 0; i < NN; i++) s[i] = (i == lp) ? dg - 1.0 : s[i];
    }
    d2v ca[NP], cbuf[NP];
    if (PIPE) {
        const unsigned nb = cb + (((NN - 1) & 1) ? 512u : 0u);
        lds_put(nb + 8 * lane, s[NN - 1]);
        col_issue<0, NP>(ca, nb);
    }
    asm_pivot<Topo, NN, NN - 1, PIPE>(s, dg, ok, cb, ca, cbuf);
    ok = __builtin_amdgcn_readfirstlane(ok);
    {
        int le = lane;
        asm volatile("" : "+v"(le));
#pragma unroll
        for (int i = 0; i < NN; i++) s[i] = (i == le) ? -dg : -s[i];
        dg = -dg;
    }
    return ok;
}


// ---- V5: V4 with the next column's reads INTERLEAVED with this pivot's FMAs (one ds_read_b128 per two FMAs, order pinned with
// sched_barrier): a broadcast ds_read_b128 costs 16 issue cycles of its own (tools/ubench/ubench3), which overlap with VALU
// issue only when the two are mixed.  The select of the multiplier is gone too: with the diagonal stored shifted by one,
// cj * rp is already 1 - 1/d in the pivot lane.
template <int P, int NP>
__device__ __forceinline__ void col_issue1(d2v (&c)[NP], unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c[P]) : "v"(addr), "n"(16 * P));
}
template <class Topo, int NN, int K, int P>
__device__ __forceinline__ void v5_rows(double (&s)[NN], const double h, unsigned nb, const d2v (&cur)[(NN + 1) / 2], d2v (&nxt)[(NN + 1) / 2]) {
    constexpr int NP = (NN + 1) / 2;
    constexpr unsigned long long rel = Topo::relatives(K);
    if (K > 0) col_issue1<P, NP>(nxt, nb);
#pragma unroll
    for (int i = 2 * P; i < 2 * P + 2 && i < NN; i++)
        if (((rel >> i) & 1ull) && !(K > 0 && i == K - 1)) s[i] -= cur[P][i & 1] * h;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (P + 1 < NP) v5_rows<Topo, NN, K, P + 1>(s, h, nb, cur, nxt);
}
template <class Topo, int NN, int K>
__device__ __forceinline__ void v5_pivot(double (&s)[NN], double &dg, int &ok, unsigned cb, d2v (&cur)[(NN + 1) / 2], d2v (&nxt)[(NN + 1) / 2]) {
    constexpr int NP = (NN + 1) / 2;
    const int lane = threadIdx.x;
    constexpr unsigned long long rel = Topo::relatives(K);
    double d = readlane_f64(dg, K);
    int pos = d > 0.0 ? 1 : 0;
    asm volatile("" : "+v"(pos));
    ok &= pos;
    if (!(d > 0.0)) d = 1.0;
    const double rp = fast_rcp(d);
    int lk = lane;
    asm volatile("" : "+v"(lk));
    const bool piv = lk == K;
    const double cj = s[K];
    const double h = cj * rp;  // pivot lane: (d - 1) / d = 1 - 1/d, the multiplier of the pivot column itself
    col_wait<NP>(cur);
    const unsigned nb = cb + (((K - 1) & 1) ? 512u : 0u);
    if (K > 0) {
        if ((rel >> (K - 1)) & 1ull) s[K - 1] -= cur[(K - 1) / 2][(K - 1) & 1] * h;
        lds_put(nb + 8 * lane, s[K - 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    v5_rows<Topo, NN, K, 0>(s, h, nb, cur, nxt);
    dg = piv ? -rp : dg - cj * h;
    if constexpr (K > 0) v5_pivot<Topo, NN, K - 1>(s, dg, ok, cb, nxt, cur);
}
template <class Topo, int NN>
__device__ __forceinline__ int sweep_v5(double (&s)[NN], double &dg, double *cbp) {
    constexpr int NP = (NN + 1) / 2;
    const int lane = threadIdx.x;
    const unsigned cb = (unsigned)(size_t)cbp;
    int ok = 1;
    {
        int lp = lane;
        asm volatile("" : "+v"(lp));
#pragma unroll
        for (int i = 0; i < NN; i++) s[i] = (i == lp) ? dg - 1.0 : s[i];
    }
    d2v ca[NP], cbuf[NP];
    const unsigned nb = cb + (((NN - 1) & 1) ? 512u : 0u);
    lds_put(nb + 8 * lane, s[NN - 1]);
    col_issue<0, NP>(ca, nb);
    v5_pivot<Topo, NN, NN - 1>(s, dg, ok, cb, ca, cbuf);
    ok = __builtin_amdgcn_readfirstlane(ok);
    {
        int le = lane;
        asm volatile("" : "+v"(le));
#pragma unroll
        for (int i = 0; i < NN; i++) s[i] = (i == le) ? -dg : -s[i];
        dg = -dg;
    }
    return ok;
}

template <class Topo, int NN, int V>
__global__ __launch_bounds__(64) void k_sweep(const double *A, double *out, unsigned long long *cyc, int reps) {
    __shared__ __attribute__((aligned(1024))) double cb[128];
    const int lane = threadIdx.x;
    double s[NN], dg;
#pragma unroll
    for (int i = 0; i < NN; i++) s[i] = lane < NN ? A[i * NN + lane] : 0.0;
    dg = lane < NN ? A[lane * NN + lane] : 1.0;
    int ok = 1;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int r = 0; r < 2 * reps + 1; r++) {  // odd count: the result is the inverse
        if (V == 0) ok &= sweep_inverse_tree<Topo, NN>(s, dg);
        else if (V == 1) ok &= sweep_lds<Topo, NN, false>(s, dg, cb);
        else if (V == 2) ok &= sweep_lds<Topo, NN, true>(s, dg, cb);
        else if (V == 3) ok &= sweep_asm<Topo, NN, false>(s, dg, cb);
        else if (V == 4) ok &= sweep_asm<Topo, NN, true>(s, dg, cb);
        else ok &= sweep_v5<Topo, NN>(s, dg, cb);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) cyc[blockIdx.x] = (t1 - t0) / (2 * reps + 1);
    if (blockIdx.x == 0 && lane < NN) {
#pragma unroll
        for (int i = 0; i < NN; i++) out[i * NN + lane] = s[i] + (ok ? 0.0 : 1e300);
    }
}

static unsigned long long median(std::vector<unsigned long long> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

template <class Topo, int NN>
int run(const char *name, bool tree) {
    const int G = 1024, reps = 20;
    std::vector<double> A(NN * NN, 0.0);
    if (!tree) {
        for (int i = 0; i < NN; i++)
            for (int j = 0; j < NN; j++) A[i * NN + j] = (i == j ? 3.0 + 0.1 * i : 0.0) + 1.0 / (2.0 + i + j);
    } else {
        // a mass-matrix-like SPD matrix with the tree's sparsity: sum over dofs b of v_b v_b^T, v_b supported on the ancestors of b
        for (int b = 0; b < NN; b++) {
            std::vector<double> v(NN, 0.0);
            for (int i = 0; i <= b; i++)
                if (((Topo::relatives(b) >> i) & 1ull)) v[i] = 0.3 + 0.01 * ((i * 7 + b * 3) % 11);
            v[b] = 1.5;
            for (int i = 0; i < NN; i++)
                for (int j = 0; j < NN; j++) A[i * NN + j] += v[i] * v[j];
        }
    }
    double *dA, *dout; unsigned long long *dc;
    CK(hipMalloc(&dA, NN * NN * 8)); CK(hipMalloc(&dout, NN * NN * 8)); CK(hipMalloc(&dc, G * 8));
    CK(hipMemcpy(dA, A.data(), NN * NN * 8, hipMemcpyHostToDevice));
    std::vector<double> ref(NN * NN), got(NN * NN);
    std::vector<unsigned long long> c(G);
    for (int v = 0; v < 6; v++) {
        for (int it = 0; it < 2; it++) {
            if (v == 0) k_sweep<Topo, NN, 0><<<G, 64>>>(dA, dout, dc, reps);
            else if (v == 1) k_sweep<Topo, NN, 1><<<G, 64>>>(dA, dout, dc, reps);
            else if (v == 2) k_sweep<Topo, NN, 2><<<G, 64>>>(dA, dout, dc, reps);
            else if (v == 3) k_sweep<Topo, NN, 3><<<G, 64>>>(dA, dout, dc, reps);
            else if (v == 4) k_sweep<Topo, NN, 4><<<G, 64>>>(dA, dout, dc, reps);
            else k_sweep<Topo, NN, 5><<<G, 64>>>(dA, dout, dc, reps);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(c.data(), dc, G * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(got.data(), dout, NN * NN * 8, hipMemcpyDeviceToHost));
        if (v == 0) ref = got;
        double err = 0, chk = 0;
        for (int i = 0; i < NN * NN; i++) err = std::max(err, std::fabs(got[i] - ref[i]));
        // check A * inv = I on the first variant
        for (int i = 0; i < NN; i++)
            for (int j = 0; j < NN; j++) {
                double acc = 0;
                for (int k = 0; k < NN; k++) acc += A[i * NN + k] * got[k * NN + j];
                chk = std::max(chk, std::fabs(acc - (i == j ? 1.0 : 0.0)));
            }
        printf("%-28s V%d: %8llu cycles per sweep | max |result - V0| %.2e | max |A inv - I| %.2e\n", name, v, median(c), err, chk);
    }
    return 0;
}

int main() {
    if (run<TopoDense<33>, 33>("33 x 33 dense (W^+)", false)) return 1;
    if (run<TopoTocabi, 39>("39 x 39 TOCABI tree (A^-1)", true)) return 1;
    if (run<TopoDense<39>, 39>("39 x 39 dense", false)) return 1;
    return 0;
}

// How many single-wave workgroups of a given dynamic-LDS size share a CU?  Launches 256 * w workgroups of fixed work and
// prints the time: a jump between w and w+1 marks the co-residency limit.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(64) void spin(double* out, int reps) {
    extern __shared__ double lds[];
    double a = threadIdx.x, b = 1.0000001;
    lds[threadIdx.x] = a;
    __syncthreads();
    for (int r = 0; r < reps; r++) a = a * b + lds[(threadIdx.x + r) & 63];
    out[blockIdx.x * 64 + threadIdx.x] = a;
}
int main() {
    double* out; CK(hipMalloc(&out, 256 * 16 * 64 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int bytes : {40960, 34560, 32768, 32520, 32256, 31744, 30720, 28672, 27136, 26624, 23040, 20480, 20224}) {
        printf("LDS %6d B:", bytes);
        for (int w = 4; w <= 9; w++) {
            spin<<<256 * w, 64, bytes>>>(out, 20000); CK(hipDeviceSynchronize());
            float ms; CK(hipEventRecord(e0)); spin<<<256 * w, 64, bytes>>>(out, 20000); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("  w=%d %.2f ms", w, ms);
        }
        printf("\n");
    }
    return 0;
}

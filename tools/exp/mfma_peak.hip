// FP64 MFMA ceiling: back-to-back v_mfma_f64_16x16x4_f64 on independent accumulators, no memory traffic
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4v __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
    double4v acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (double4v){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu, int iters) {
    double* out; hipMalloc(&out, sizeof(double) * 256 * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    k<NACC><<<grid, 256>>>(out, 10);
    hipEventRecord(e0);
    k<NACC><<<grid, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 /*waves*/ * iters * NACC * 2048.0;
    printf("acc=%d blocks/CU=%d: %.1f TFLOP/s (%.3f ms)\n", NACC, blocks_per_cu, flops / ms / 1e9, ms);
    hipFree(out);
}
int main() {
    run<4>(1, 20000); run<8>(1, 10000); run<16>(1, 5000); run<8>(2, 10000); run<8>(3, 10000); run<16>(2, 5000);
    return 0;
}

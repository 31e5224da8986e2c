// FP64 MFMA ceiling, third look: one block per CU, 1..4 waves per SIMD guaranteed co-resident
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4v __attribute__((ext_vector_type(4)));
template <int NACC, int THREADS>
__global__ __launch_bounds__(THREADS) void k(double* out, long long* clk, int iters) {
    double4v acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (double4v){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    const long long c0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const long long c1 = clock64(), r1 = wall_clock64();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 7) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}
template <int NACC, int THREADS>
void run(int iters) {
    double* out; (void)hipMalloc(&out, sizeof(double) * 1024 * 256 * 8);
    long long* clk; (void)hipMalloc(&clk, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256;
    k<NACC, THREADS><<<grid, THREADS>>>(out, clk, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<NACC, THREADS><<<grid, THREADS>>>(out, clk, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = (double)grid * (THREADS / 64) * iters * NACC * 2048.0;
    printf("acc=%2d waves/SIMD=%d: %5.1f TFLOP/s (%.3f ms)  in-kernel %.3f ms  cycles per MFMA per SIMD %.1f\n", NACC, THREADS / 256,
           flops / ms / 1e9, ms, h[1] / 1e5, (double)h[0] / ((double)iters * NACC) / (THREADS / 256));
    (void)hipFree(out); (void)hipFree(clk);
}
int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run<4, 256>(8000); run<4, 512>(8000); run<4, 768>(8000); run<4, 1024>(8000);
        run<8, 512>(4000); run<8, 1024>(4000); run<2, 1024>(16000); run<1, 1024>(32000);
    }
    return 0;
}

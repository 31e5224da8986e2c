// Phase timing of chol_diag_kernel (and the panel solve) on one 128x128 block.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -DSTBA_DIAG_TS -Islam-tricks_amd/csrc tools/exp/diag_timing.hip -o /tmp/diag_timing
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../../slam-tricks_amd/csrc/dense_chol.hip"
namespace stba { thread_local std::string g_last_error; }
int main() {
    const int lda = 1024, n = 1000;
    std::vector<double> h((size_t)lda * lda, 0.0);
    for (int i = 0; i < lda; ++i) for (int j = 0; j <= i; ++j) h[(size_t)i * lda + j] = (i == j) ? lda + 1.0 : std::sin(0.37 * i + 0.11 * j);
    double *A, *dinv; int* flag;
    hipMalloc((void**)&A, h.size() * 8); hipMalloc((void**)&dinv, 2048 * 8 + 128 * 128 * 8); hipMalloc((void**)&flag, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        hipMemset(flag, 0, 4);
        hipDeviceSynchronize();
        long long w0 = 0, w1 = 0;
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(stba::chol_diag_kernel, dim3(1), dim3(512), 0, 0, A, lda, 0, n, flag, dinv);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("diag kernel %.2f us\n", ms * 1e3);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(stba::chol_trsm_kernel, dim3(7 * 8 + 8), dim3(64), 0, 0, A, lda, 0, 7 * 8, dinv + 2048, dinv);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        printf("trsm kernel (64 groups) %.2f us\n", ms * 1e3);
        (void)w0; (void)w1;
    }
    long long ts[8][16][6];
    hipMemcpyFromSymbol(ts, HIP_SYMBOL(stba::g_diag_ts), sizeof ts);
    printf("total cycles wave0: %lld\n", ts[0][15][5] - ts[0][0][0]);
    for (int w = 0; w < 8; w += 1) {
        printf("wave %d\n step:  e2prev   c      bar    e1+f    a+bar | total\n", w);
        for (int s = 0; s < 16; ++s) {
            printf("  %2d: ", s);
            for (int k = 0; k < 5; ++k) printf("%6lld ", ts[w][s][k + 1] - ts[w][s][k]);
            printf("| %6lld\n", ts[w][s][5] - ts[w][s][0]);
        }
    }
    return 0;
}

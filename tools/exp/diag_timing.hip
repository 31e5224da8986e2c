// Phase timing of chol_diag_kernel (and the panel solve) on one 128x128 block.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -Islam-tricks_amd/csrc tools/exp/diag_timing.hip -o /tmp/diag_timing
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
    long long ts[8][12];
    hipMemcpyFromSymbol(ts, HIP_SYMBOL(stba::g_diag_ts), sizeof ts);
    printf("row thread 127, cycles per tile column:\n Jt  publish  Bp    c0     B1   local    Bq    c1     B2   store    B3   | total\n");
    for (int J = 0; J < 8; ++J) {
        printf(" %d ", J);
        for (int k = 0; k < 10; ++k) printf("%6lld ", ts[J][k + 1] - ts[J][k]);
        printf("| %6lld\n", ts[J][10] - ts[J][0]);
    }
    long long t2[8][8];
    hipMemcpyFromSymbol(t2, HIP_SYMBOL(stba::g_diag_ts2), sizeof t2);
    printf("matrix-core wave 2 (tile rows 1,7):\n Jt    Bp  e(prev,1)rest  B1+Bq   e(Jt,0)    B2   e1+publish   B3\n");
    for (int J = 0; J < 7; ++J) {
        printf(" %d ", J);
        for (int k = 0; k < 7; ++k) printf("%7lld ", t2[J][k + 1] - t2[J][k]);
        printf("\n");
    }
    return 0;
}

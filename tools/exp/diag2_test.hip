// Correctness + timing of the second diagonal-block design (diag_block2) against the first (diag_block).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -Islam-tricks_amd/csrc -Iinclude tools/exp/diag2_test.hip -o tools/exp/diag2_test.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../../slam-tricks_amd/csrc/dense_chol.hip"
namespace stba { thread_local std::string g_last_error; }
int main() {
    const int lda = 1024, n = 1000;
    std::vector<double> h((size_t)lda * lda, 0.0);
    // SPD-ish 128x128 leading block: diagonally dominant with structure
    for (int i = 0; i < lda; ++i) for (int j = 0; j <= i; ++j) h[(size_t)i * lda + j] = (i == j) ? 200.0 + 0.01 * i : std::sin(0.37 * i + 0.11 * j) + 0.3 * std::cos(0.05 * (i - j));
    double *A1, *A2, *dinv1, *dinv2; int* flag;
    hipMalloc((void**)&A1, h.size() * 8); hipMalloc((void**)&A2, h.size() * 8);
    hipMalloc((void**)&dinv1, 2048 * 8); hipMalloc((void**)&dinv2, 4096 * 8); hipMalloc((void**)&flag, 4);
    long long* cyc; hipMalloc((void**)&cyc, 16);
    hipFuncSetAttribute(reinterpret_cast<const void*>(stba::chol_diag2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds2 = sizeof(stba::Diag2Smem) + 64;
    for (int rep = 0; rep < 4; ++rep) {
        hipMemcpy(A1, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        hipMemcpy(A2, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        hipMemset(flag, 0, 4); hipMemset(dinv1, 0, 2048 * 8); hipMemset(dinv2, 0, 2048 * 8);
        hipDeviceSynchronize();
        float ms1, ms2;
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(stba::chol_diag_kernel, dim3(1), dim3(512), 0, 0, A1, lda, 0, n, flag, dinv1);
        hipEventRecord(e1, 0); hipDeviceSynchronize(); hipEventElapsedTime(&ms1, e0, e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(stba::chol_diag2_kernel, dim3(1), dim3(512), lds2, 0, A2, lda, 0, n, flag, dinv2, 256, cyc);
        hipEventRecord(e1, 0);
        hipError_t err = hipDeviceSynchronize(); hipEventElapsedTime(&ms2, e0, e1);
        int fl = 0; hipMemcpy(&fl, flag, 4, hipMemcpyDeviceToHost);
        long long hc = 0; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        printf("rep %d: diag v1 %.2f us (kernel incl. launch)   diag v2 warm, in-kernel %.2f us   (err %d flag %d)\n", rep, ms1 * 1e3, hc / 100.0, (int)err, fl);
    }
    std::vector<double> r1(h.size()), r2(h.size()), d1(2048), d2(2048);
    hipMemcpy(r1.data(), A1, h.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), A2, h.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(d1.data(), dinv1, 2048 * 8, hipMemcpyDeviceToHost); hipMemcpy(d2.data(), dinv2, 2048 * 8, hipMemcpyDeviceToHost);
    // reference: host Cholesky of the block
    std::vector<double> L(128 * 128, 0.0);
    for (int j = 0; j < 128; ++j) {
        double d = h[(size_t)j * lda + j];
        for (int k = 0; k < j; ++k) d -= L[j * 128 + k] * L[j * 128 + k];
        L[j * 128 + j] = std::sqrt(d);
        for (int i = j + 1; i < 128; ++i) {
            double v = h[(size_t)i * lda + j];
            for (int k = 0; k < j; ++k) v -= L[i * 128 + k] * L[j * 128 + k];
            L[i * 128 + j] = v / L[j * 128 + j];
        }
    }
    double e1h = 0, e2h = 0, e12 = 0, ed = 0, emax_up = 0;
    for (int i = 0; i < 128; ++i) for (int j = 0; j <= i; ++j) {
        e1h = std::fmax(e1h, std::fabs(r1[(size_t)i * lda + j] - L[i * 128 + j]));
        e2h = std::fmax(e2h, std::fabs(r2[(size_t)i * lda + j] - L[i * 128 + j]));
        e12 = std::fmax(e12, std::fabs(r1[(size_t)i * lda + j] - r2[(size_t)i * lda + j]));
    }
    for (int k = 0; k < 2048; ++k) ed = std::fmax(ed, std::fabs(d1[k] - d2[k]));
    // rows below the block and everything else must be untouched by v2
    for (int i = 128; i < lda; ++i) for (int j = 0; j < lda; ++j) emax_up = std::fmax(emax_up, std::fabs(r2[(size_t)i * lda + j] - h[(size_t)i * lda + j]));
#ifdef STBA_DIAG_TS
    {
        static long long ev[8][160][2]; int nev[8];
        hipMemcpyFromSymbol(ev, HIP_SYMBOL(stba::g_d2_ev), sizeof ev);
        hipMemcpyFromSymbol(nev, HIP_SYMBOL(stba::g_d2_nev), sizeof nev);
        // merge the two factor waves: per step tt: Gp made (1tt), behind the barrier (2tt)
        long long T[33][2] = {{0}}, t0 = 1LL << 62;
        for (int w = 0; w < 2; ++w) for (int k = 0; k < nev[w] && k < 160; ++k) {
            const long long c = ev[w][k][0];
            if (c >= 100 && c < 300) T[c % 100][c / 100 - 1] = ev[w][k][1];
            if (ev[w][k][1] < t0) t0 = ev[w][k][1];
        }
        printf("factor chain (cycles): step | Gp made -> behind the barrier | barrier -> next Gp made\n");
        for (int tt = 1; tt < 32; ++tt) printf(" %2d | %5lld | %5lld\n", tt, T[tt][1] - T[tt][0], tt < 31 ? T[tt + 1][0] - T[tt][1] : 0LL);
        printf("chain total %lld cycles\n", T[31][1] - T[1][0]);
        // bulk waves: per step, relative to the factor wave's time behind the barrier: behind the barrier | updates done | step done
        for (int w : {2, 6, 5}) {
            printf("bulk wave %d: step | behind barrier | updates done | step done   (relative to the factor wave behind the barrier)\n", w);
            long long B[32][3] = {{0}};
            for (int k = 0; k < nev[w] && k < 160; ++k) { const long long c = ev[w][k][0]; if (c >= 300 && c < 600) B[c % 100][c / 100 - 3] = ev[w][k][1]; }
            for (int tt = 0; tt < 12; ++tt) printf(" %2d | %5lld | %5lld | %5lld\n", tt, B[tt][0] - T[tt][1], B[tt][1] - T[tt][1], B[tt][2] - T[tt][1]);
        }
    }
#endif
    printf("L: |v1 - host| %.3g  |v2 - host| %.3g  |v1 - v2| %.3g   dinv |v1 - v2| %.3g   outside block changed by %.3g\n", e1h, e2h, e12, ed, emax_up);
    return 0;
}

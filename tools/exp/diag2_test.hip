// Correctness + timing of the second diagonal-block design (diag_block2) against the first (diag_block).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -Islam-tricks_amd/csrc -Iinclude tools/exp/diag2_test.hip -o tools/exp/diag2_test.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../../slam-tricks_amd/csrc/dense_chol.hip"
namespace stba { thread_local std::string g_last_error; }
// ---- micro-kernels: the factor wave alone, and factor wave + follower, on synthetic tiles (timing only)
namespace stba {
__device__ __forceinline__ long long mk_clock(double dep) { long long c; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c) : "v"(dep) : "memory"); return c; }
__global__ __launch_bounds__(512) void d2_micro_kernel(long long* out, int mode) {
    extern __shared__ __attribute__((aligned(16))) double sm2[];
    Diag2Smem& sm = *reinterpret_cast<Diag2Smem*>(sm2);
    const int t = threadIdx.x, w = t >> 6, lane = t & 63, n = lane & 15, g = lane >> 4;
    {
        unsigned long long* p = reinterpret_cast<unsigned long long*>(sm2);
        constexpr int NSENT = 36 * 4 * 64 + 32 * 64 + 8 * 2 * 4 * 64;
        for (int e = t; e < NSENT; e += 512) p[e] = D2_SENTINEL;
        if (t == 0) sm.abortf = 0u;
    }
    __syncthreads();
    double mk[10];
    { const int idx = (n < 4 && g <= n) ? n * (n + 1) / 2 + g : -1;
#pragma unroll
      for (int e = 0; e < 10; ++e) mk[e] = (idx == e) ? 1.0 : 0.0; }
    int badv = 0;
    if (w == 0) {
        // factor wave: 8 well-conditioned diagonal tiles in a row (fresh tile per row, as after a hand-over)
        double chk = 0.0;
        long long t0 = mk_clock(chk), tt[9];
        tt[0] = t0;
#pragma unroll
        for (int I = 0; I < 8; ++I) {
            double4v accD;
#pragma unroll
            for (int r = 0; r < 4; ++r) accD[r] = ((n == 4 * r + g) ? 50.0 + I : 0.01 * (n + 4 * r + g)) + chk * 1e-30;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double gp = d2_factor_gp(accD, s, mk, badv, 0, 1 << 30);
                sm.Gp[4 * I + s][lane] = gp;
                const double l = d2_factor_update(accD, s, gp);
                sm.Lsl[d2_tix(I, I)][s][lane] = l;
                chk += l;
            }
            tt[I + 1] = mk_clock(chk);
        }
        if (lane == 0) { for (int I = 0; I < 9; ++I) out[I] = tt[I] - t0; out[20] = (long long)chk + badv; }
    } else if (w == 1 && mode >= 1) {
        // follower: for every tile column J the two tiles of row J+1
        double chk = 0.0;
        long long tt[9];
        tt[0] = mk_clock(chk);
#pragma unroll
        for (int J = 0; J < 8; ++J) {
            double4v accS, accD;
#pragma unroll
            for (int r = 0; r < 4; ++r) { accS[r] = 0.01 * (n - g + r) + chk * 1e-30; accD[r] = (n == 4 * r + g) ? 60.0 : 0.02; }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double gp = d2_take(&sm.Gp[4 * J + s][lane], &sm.abortf);
                const double l = mfma_l(gp, accS[s]);
                sm.Lsl[d2_tix(7, J == 7 ? 6 : J)][s][lane] = l;       // (any free slot: row 7's)
                accD = __builtin_amdgcn_mfma_f64_16x16x4f64(-l, l, accD, 0, 0, 0);
                if (s < 3) {
                    const double lj = d2_take(&sm.Lsl[d2_tix(J, J)][s][lane], &sm.abortf);
                    accS = __builtin_amdgcn_mfma_f64_16x16x4f64(-lj, l, accS, 0, 0, 0);
                }
            }
            chk += accD[0] + accD[1] + accD[2] + accD[3];
            tt[J + 1] = mk_clock(chk);
        }
        if (lane == 0) { for (int I = 0; I < 9; ++I) out[32 + I] = tt[I]; out[52] = (long long)chk; }
    } else if (mode >= 2 && w >= 2) {
        // pollers: the other six waves poll a slot that is published last (LDS polling traffic as in the real task)
        (void)d2_take(&sm.Lsl[d2_tix(7, 7)][3][lane], &sm.abortf);
    }
    if (w == 0 && lane == 0) out[31] = 0;
}
}
int main() {
    {
        long long* out; hipMalloc((void**)&out, 64 * 8);
        hipFuncSetAttribute(reinterpret_cast<const void*>(stba::d2_micro_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        for (int mode = 0; mode < 3; ++mode) {
            long long h[64];
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(out, 0, 64 * 8);
                hipLaunchKernelGGL(stba::d2_micro_kernel, dim3(1), dim3(512), sizeof(stba::Diag2Smem) + 64, 0, out, mode);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
            printf("micro mode %d (0: factor wave alone, 1: + follower, 2: + six pollers): factor-wave cycles per tile:", mode);
            for (int I = 0; I < 8; ++I) printf(" %lld", h[I + 1] - h[I]);
            printf("  total %lld\n", h[8]);
            if (mode >= 1) {
                printf("   follower column end minus factor-wave tile end:");
                // both clocks are s_memtime: compare absolute values (factor wave's t0 unknown here -> print follower deltas)
                for (int J = 0; J < 8; ++J) printf(" %lld", h[32 + J + 1] - h[32 + J]);
                printf("\n");
            }
        }
        hipFree(out);
    }

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
        long long t0 = 1LL << 62;
        for (int w = 0; w < 8; ++w) for (int k = 0; k < nev[w] && k < 160; ++k) t0 = ev[w][k][1] < t0 ? ev[w][k][1] : t0;
        printf("event trace (cycles since the first event).  codes: 900 start, 1xx bulk got Gp(step), 2xx bulk step done, 999 bulk handed over,\n"
               "  3II migrated tiles of row II received, 4xx follower step done, 5xx factor step done (l there), 6II inverse tile of row II done, 7II tile column stored\n");
        for (int w = 0; w < 8; ++w) {
            printf("wave %d:", w);
            for (int k = 0; k < nev[w] && k < 160; ++k) printf(" %lld@%lld", ev[w][k][0], ev[w][k][1] - t0);
            printf("\n");
        }
    }
#endif
    printf("L: |v1 - host| %.3g  |v2 - host| %.3g  |v1 - v2| %.3g   dinv |v1 - v2| %.3g   outside block changed by %.3g\n", e1h, e2h, e12, ed, emax_up);
    return 0;
}

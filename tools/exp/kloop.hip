// What does the K loop of the trailing update lose its 15 % to?  The loop's SHAPE in isolation: one 512-thread workgroup per CU,
// operand fragments re-read from LDS every k-step, v_mfma_f64_16x16x4_f64 on the wave's accumulator tiles, no global traffic at all.
// Variants (template parameters): MB x NBK MFMA tiles per wave; READS: 0 none (operands stay in registers), 1 one ds_read_b64 per
// fragment, 2 one ds_read_b128 per fragment PAIR (two k-steps of one operand); BAR: 0 no barrier, 1 a barrier every CH k-steps,
// 2 barrier + the staging stores (4 ds_write_b64 pairs per 4 k-steps and wave, as the task writes the next chunk); THREADS 512 / 768 / 1024.
// build: hipcc -O3 --offload-arch=gfx950 tools/exp/kloop.hip -o tools/exp/kloop.bin ; run: tools/exp/kloop.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4v __attribute__((ext_vector_type(4)));
typedef double double2v __attribute__((ext_vector_type(2)));

template <int MB, int NBK, int READS, int BAR, int CH, int THREADS>
__global__ __launch_bounds__(THREADS) void kloop(double* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int e = t; e < 16384; e += THREADS) smem[e] = 1e-3 * (e & 255);
    __syncthreads();
    double4v acc[MB][NBK];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NBK; ++n) acc[m][n] = (double4v){0, 0, 0, 0};
    double a[MB], b[NBK];
#pragma unroll
    for (int m = 0; m < MB; ++m) a[m] = 1e-3 * (lane + m);
#pragma unroll
    for (int n = 0; n < NBK; ++n) b[n] = 1.0 + 1e-4 * (lane + n);
    const double* sA = smem + (w & 1) * 256;
    const double* sB = smem + 8192 + (w >> 1) * 16;
    for (int it = 0; it < iters; ++it) {
        const int buf = (it & 1) * 4096;
#pragma unroll
        for (int kq = 0; kq < CH; ++kq) {
            if (READS == 1) {
#pragma unroll
                for (int m = 0; m < MB; ++m) a[m] = sA[buf + ((kq * 8 + m) << 6) + lane];
#pragma unroll
                for (int n = 0; n < NBK; ++n) b[n] = sB[buf + ((kq * 8 + n) << 6) + lane];
            }
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NBK; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
            if (BAR == 3 && kq == CH / 2 - 1) {      // the staging stores in the MIDDLE of the chunk: done long before the barrier
                double* d = smem + (buf ^ 4096) + 8192 * (w & 1) + lane;
#pragma unroll
                for (int h = 0; h < 4; ++h) { d[((w & 7) * 4 + h) * 64] = 1e-3 * (lane + h); d[((w & 7) * 4 + h) * 64 + 16] = 1e-3 * h; }
            }
        }
        if (BAR == 2) {
            double* d = smem + (buf ^ 4096) + 8192 * (w & 1) + lane;
#pragma unroll
            for (int h = 0; h < 4; ++h) { d[((w & 7) * 4 + h) * 64] = 1e-3 * (lane + h); d[((w & 7) * 4 + h) * 64 + 16] = 1e-3 * h; }
        }
        if (BAR >= 1) __syncthreads();
    }
    double s = 0;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NBK; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[(size_t)blockIdx.x * THREADS + t] = s;
}

// READS == 2 as its own kernel (the fragment pair in one ds_read_b128)
template <int MB, int NBK, int BAR, int CH, int THREADS>
__global__ __launch_bounds__(THREADS) void kloop128(double* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int e = t; e < 16384; e += THREADS) smem[e] = 1e-3 * (e & 255);
    __syncthreads();
    double4v acc[MB][NBK];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NBK; ++n) acc[m][n] = (double4v){0, 0, 0, 0};
    const double2v* sA = reinterpret_cast<const double2v*>(smem) + (w & 1) * 128;
    const double2v* sB = reinterpret_cast<const double2v*>(smem + 8192) + (w >> 1) * 8;
    for (int it = 0; it < iters; ++it) {
        const int buf = (it & 1) * 2048;
#pragma unroll
        for (int kp = 0; kp < CH / 2; ++kp) {
            double2v a2[MB], b2[NBK];
#pragma unroll
            for (int m = 0; m < MB; ++m) a2[m] = sA[buf + ((kp * 8 + m) << 6) + lane];
#pragma unroll
            for (int n = 0; n < NBK; ++n) b2[n] = sB[buf + ((kp * 8 + n) << 6) + lane];
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NBK; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[m].x, b2[n].x, acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NBK; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[m].y, b2[n].y, acc[m][n], 0, 0, 0);
        }
        if (BAR == 2) {
            double* d = smem + ((it & 1) ^ 1) * 4096 + 8192 * (w & 1) + lane;
#pragma unroll
            for (int h = 0; h < 4; ++h) { d[((w & 7) * 4 + h) * 64] = 1e-3 * (lane + h); d[((w & 7) * 4 + h) * 64 + 16] = 1e-3 * h; }
        }
        if (BAR >= 1) __syncthreads();
    }
    double s = 0;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NBK; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[(size_t)blockIdx.x * THREADS + t] = s;
}

// fragments of k-step q + 1 requested before the MFMAs of k-step q (second register set); PIN: sched_barrier(0) around the MFMA groups
template <int MB, int NBK, int BAR, int CH, int PIN, int THREADS>
__global__ __launch_bounds__(THREADS) void kloop_pipe(double* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int e = t; e < 16384; e += THREADS) smem[e] = 1e-3 * (e & 255);
    __syncthreads();
    double4v acc[MB][NBK];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NBK; ++n) acc[m][n] = (double4v){0, 0, 0, 0};
    const double* sA = smem + (w & 1) * 256;
    const double* sB = smem + 8192 + (w >> 1) * 16;
    for (int it = 0; it < iters; ++it) {
        const int buf = (it & 1) * 4096;
        double a[2][MB], b[2][NBK];
        auto frags = [&](int kq, int set) {
#pragma unroll
            for (int m = 0; m < MB; ++m) a[set][m] = sA[buf + ((kq * 8 + m) << 6) + lane];
#pragma unroll
            for (int n = 0; n < NBK; ++n) b[set][n] = sB[buf + ((kq * 8 + n) << 6) + lane];
        };
        frags(0, 0);
#pragma unroll
        for (int kq = 0; kq < CH; ++kq) {
            if (kq + 1 < CH) frags(kq + 1, (kq + 1) & 1);
            if (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NBK; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kq & 1][m], b[kq & 1][n], acc[m][n], 0, 0, 0);
            if (PIN) __builtin_amdgcn_sched_barrier(0);
        }
        if (BAR == 2) {
            double* d = smem + (buf ^ 4096) + 8192 * (w & 1) + lane;
#pragma unroll
            for (int h = 0; h < 4; ++h) { d[((w & 7) * 4 + h) * 64] = 1e-3 * (lane + h); d[((w & 7) * 4 + h) * 64 + 16] = 1e-3 * h; }
        }
        if (BAR >= 1) __syncthreads();
    }
    double s = 0;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NBK; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[(size_t)blockIdx.x * THREADS + t] = s;
}

template <typename K>
static void time_it(const char* name, K kern, int threads, int mfma_per_iter, int iters) {
    double* out; (void)hipMalloc(&out, sizeof(double) * 1024 * 256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 131072, 0, out, 10);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 131072, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = 256.0 * (threads / 64) * (double)iters * mfma_per_iter * 2048.0;
    printf("%-64s %6.1f TFLOP/s = %.3f of 78.6 (%.3f ms)\n", name, flops / best / 1e9, flops / best / 1e9 / 78.6, best);
    (void)hipFree(out);
}

int main() {
    const int IT = 4000;
    time_it("4x2, no reads, no barrier", kloop<4, 2, 0, 0, 4, 512>, 512, 32, IT);
    time_it("4x2, no reads, barrier / 4 k-steps", kloop<4, 2, 0, 1, 4, 512>, 512, 32, IT);
    time_it("4x2, b64 reads, no barrier", kloop<4, 2, 1, 0, 4, 512>, 512, 32, IT);
    time_it("4x2, b64 reads, barrier / 4 k-steps", kloop<4, 2, 1, 1, 4, 512>, 512, 32, IT);
    time_it("4x2, b64 reads, barrier + staging stores / 4  (the task's shape)", kloop<4, 2, 1, 2, 4, 512>, 512, 32, IT);
    time_it("4x2, b64 reads, barrier / 4, staging stores after k-step 1 of 4", kloop<4, 2, 1, 3, 4, 512>, 512, 32, IT);
    time_it("4x4, b64 reads, barrier / 4, staging stores after k-step 1 of 4", kloop<4, 4, 1, 3, 4, 512>, 512, 64, IT / 2);
    time_it("4x2, PIPELINED b64 reads (pinned), barrier + stores / 4", kloop_pipe<4, 2, 2, 4, 1, 512>, 512, 32, IT);
    time_it("4x2, PIPELINED b64 reads (compiler order), barrier + stores / 4", kloop_pipe<4, 2, 2, 4, 0, 512>, 512, 32, IT);
    time_it("4x2, PIPELINED b64 reads (pinned), no barrier", kloop_pipe<4, 2, 0, 4, 1, 512>, 512, 32, IT);
    time_it("4x4, PIPELINED b64 reads (pinned), barrier + stores / 4", kloop_pipe<4, 4, 2, 4, 1, 512>, 512, 64, IT / 2);
    time_it("4x2, b64 reads, barrier + staging stores / 8 k-steps", kloop<4, 2, 1, 2, 8, 512>, 512, 64, IT / 2);
    time_it("4x2, b128 reads (fragment pairs), barrier + stores / 4", kloop128<4, 2, 2, 4, 512>, 512, 32, IT);
    time_it("4x2, b128 reads, no barrier", kloop128<4, 2, 0, 4, 512>, 512, 32, IT);
    time_it("4x4, b64 reads, barrier + stores / 4", kloop<4, 4, 1, 2, 4, 512>, 512, 64, IT / 2);
    time_it("4x4, b128 reads, barrier + stores / 4", kloop128<4, 4, 2, 4, 512>, 512, 64, IT / 2);
    time_it("2x2, b64 reads, barrier + stores / 4, 1024 threads", kloop<2, 2, 1, 2, 4, 1024>, 1024, 16, IT);
    time_it("2x4, b64 reads, barrier + stores / 4, 1024 threads", kloop<2, 4, 1, 2, 4, 1024>, 1024, 32, IT);
    time_it("4x2, b64 reads, barrier + stores / 4, 768 threads", kloop<4, 2, 1, 2, 4, 768>, 768, 32, IT);
    time_it("2x2, b128 reads, barrier + stores / 4, 1024 threads", kloop128<2, 2, 2, 4, 1024>, 1024, 16, IT);
    return 0;
}

// Latency / issue-rate microbenchmarks of the primitives on the critical chain of the diagonal-block task:
// FP64 FMA (dependent, independent), v_rsq_f64, v_rcp_f64, dependent v_mfma_f64_16x16x4_f64, LDS write->read
// round trip inside one wave, s_barrier with 8 waves, ds_bpermute, v_readlane.  One workgroup; s_memtime cycles.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/exp/lat_f64.hip -o tools/exp/lat_f64.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4v __attribute__((ext_vector_type(4)));

// s_memtime behind a volatile asm; FENCE(x) makes the value x an input+output of a volatile asm, so the chain that
// produces / consumes it cannot move across the time stamps (volatile asms keep their order)
__device__ __forceinline__ long long cyc() { long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
#define FENCE(x) asm volatile("" : "+v"(x) :: "memory")

__global__ __launch_bounds__(512) void k(double* out, long long* res, int waves_active) {
    __shared__ double lds[4096];
    const int t = threadIdx.x, w = t >> 6;
    double x = 1.0 + t * 1e-6, y = 0.5, acc = 0.0;
    long long c0, c1;
    const int N = 256;
    if (w >= waves_active) { out[t] = 0; return; }
    // 1. dependent FMA chain
    FENCE(x); c0 = cyc(); FENCE(x);
#pragma unroll
    for (int i = 0; i < N; ++i) x = fma(x, y, 0.25);
    FENCE(x); c1 = cyc();
    if (t == 0) res[0] = (c1 - c0);
    acc += x;
    // 2. 8 independent FMA chains
    double v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 1.0 + j + t * 1e-6;
    for (int j = 0; j < 8; ++j) FENCE(v[j]);
    c0 = cyc();
    for (int j = 0; j < 8; ++j) FENCE(v[j]);
#pragma unroll
    for (int i = 0; i < N / 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fma(v[j], y, 0.25);
    for (int j = 0; j < 8; ++j) FENCE(v[j]);
    c1 = cyc();
    if (t == 0) res[1] = (c1 - c0);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j];
    // 3. dependent rsq chain
    x = 2.0 + t * 1e-6;
    FENCE(x); c0 = cyc(); FENCE(x);
#pragma unroll
    for (int i = 0; i < 64; ++i) x = __builtin_amdgcn_rsq(x) + 1.0;
    FENCE(x); c1 = cyc();
    if (t == 0) res[2] = (c1 - c0);     // 64 x (rsq + add)
    acc += x;
    // 4. dependent rcp chain
    x = 2.0 + t * 1e-6;
    FENCE(x); c0 = cyc(); FENCE(x);
#pragma unroll
    for (int i = 0; i < 64; ++i) x = __builtin_amdgcn_rcp(x) + 1.0;
    FENCE(x); c1 = cyc();
    if (t == 0) res[3] = (c1 - c0);
    acc += x;
    // 5. dependent MFMA chain (same accumulator)
    double4v a4 = {0, 0, 0, 0};
    FENCE(x); c0 = cyc(); FENCE(x);
#pragma unroll
    for (int i = 0; i < 64; ++i) a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a4, 0, 0, 0);
    double q5 = a4[0] + a4[1] + a4[2] + a4[3];
    FENCE(q5); c1 = cyc();
    acc += q5;
    if (t == 0) res[4] = (c1 - c0);
    // 5b. MFMA whose B operand is the previous result (panel-solve pattern)
    a4 = (double4v){1.0, 0.5, 0.25, 0.125};
    FENCE(y); c0 = cyc(); FENCE(y);
#pragma unroll
    for (int i = 0; i < 64; ++i) a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, a4[i & 3], (double4v){0, 0, 0, 0}, 0, 0, 0);
    q5 = a4[0] + a4[1] + a4[2] + a4[3];
    FENCE(q5); c1 = cyc();
    acc += q5;
    if (t == 0) res[5] = (c1 - c0);
    // 6. LDS write -> read round trip in one wave (dependent through memory)
    x = 1.0 + t;
    c0 = cyc();
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        lds[w * 64 + (t & 63)] = x;
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
        x = lds[w * 64 + ((t + 1) & 63)] + 1.0;
    }
    c1 = cyc();
    if (t == 0) res[6] = (c1 - c0);
    acc += x;
    // 7. s_barrier, all active waves
    __syncthreads();
    c0 = cyc();
#pragma unroll
    for (int i = 0; i < 32; ++i) __builtin_amdgcn_s_barrier();
    c1 = cyc();
    if (t == 0) res[7] = (c1 - c0);
    // 8. LDS write | barrier | read (cross-wave hand-off)
    c0 = cyc();
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        lds[t] = x;
        __syncthreads();
        x = lds[(t + 64) & 511] + 1.0;
        __syncthreads();
    }
    c1 = cyc();
    if (t == 0) res[8] = (c1 - c0);
    acc += x;
    // 9. ds_bpermute dependent chain (64-bit = two 32-bit permutes)
    int iv = t;
    FENCE(iv); c0 = cyc(); FENCE(iv);
#pragma unroll
    for (int i = 0; i < 64; ++i) iv = __builtin_amdgcn_ds_bpermute(((iv + 1) & 63) << 2, iv);
    FENCE(iv); c1 = cyc();
    if (t == 0) res[9] = (c1 - c0);
    acc += iv;
    // 10. v_readlane (readfirstlane-dependent chain)
    iv = t;
    FENCE(iv); c0 = cyc(); FENCE(iv);
#pragma unroll
    for (int i = 0; i < 64; ++i) iv = __builtin_amdgcn_readlane(iv, 5) + (t & 1);
    FENCE(iv); c1 = cyc();
    if (t == 0) res[10] = (c1 - c0);
    acc += iv;
    // 11. DPP row broadcast-ish: quad_perm dependent chain
    iv = t;
    FENCE(iv); c0 = cyc(); FENCE(iv);
#pragma unroll
    for (int i = 0; i < 64; ++i) iv = __builtin_amdgcn_update_dpp(0, iv, 0x1b, 0xf, 0xf, false) + 1;
    FENCE(iv); c1 = cyc();
    if (t == 0) res[11] = (c1 - c0);
    acc += iv;
    // 12. full Halley rsqrt + scale + update step, dependent (the pivot chain as written in mini_chol_solve)
    double d = 3.0 + t * 1e-6, o = 0.1, p = 0.2;
    FENCE(d); c0 = cyc(); FENCE(d);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const double yy = __builtin_amdgcn_rsq(d);
        const double e = fma(-d * yy, yy, 1.0);
        const double pp = fma(0.375, e, 0.5);
        const double yr = fma(yy * e, pp, yy);
        const double l = o * yr;
        d = fma(-l, l, p + d);
    }
    FENCE(d); c1 = cyc();
    if (t == 0) res[12] = (c1 - c0);
    acc += d;
    out[t] = acc;
}

int main() {
    double* out; long long* res;
    (void)hipMalloc(&out, 8 * 512); (void)hipMalloc(&res, 8 * 16);
    for (int wa : {1, 2, 8}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, res, wa);
            (void)hipDeviceSynchronize();
        }
        long long h[16];
        (void)hipMemcpy(h, res, sizeof h, hipMemcpyDeviceToHost);
        printf("active waves %d (cycles per op):\n", wa);
        printf("  dependent v_fma_f64            %6.1f\n", h[0] / 256.0);
        printf("  independent v_fma_f64 (8 ch)   %6.1f\n", h[1] / 256.0);
        printf("  dependent rsq_f64 + add        %6.1f\n", h[2] / 64.0);
        printf("  dependent rcp_f64 + add        %6.1f\n", h[3] / 64.0);
        printf("  dependent mfma_f64_16x16x4 (C) %6.1f\n", h[4] / 64.0);
        printf("  dependent mfma (B <- prev D)   %6.1f\n", h[5] / 64.0);
        printf("  LDS write->wait->read (1 wave) %6.1f\n", h[6] / 32.0);
        printf("  s_barrier                      %6.1f\n", h[7] / 32.0);
        printf("  write|barrier|read|barrier     %6.1f\n", h[8] / 32.0);
        printf("  ds_bpermute dependent          %6.1f\n", h[9] / 64.0);
        printf("  v_readlane dependent + add     %6.1f\n", h[10] / 64.0);
        printf("  dpp quad_perm dependent + add  %6.1f\n", h[11] / 64.0);
        printf("  pivot step (rsq+Halley+mul+fma)%6.1f\n", h[12] / 32.0);
    }
    return 0;
}

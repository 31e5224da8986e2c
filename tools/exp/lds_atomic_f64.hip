// Throughput of LDS FP64 atomic adds (ds_add_f64) on MI355X, the instruction the Schur pair kernel accumulates with.
// Patterns: (a) every lane its own address, conflict-free stride; (b) the Schur kernel's pattern: a lane adds the 36
// entries of one 6x6 block (block stride 37 doubles), blocks random per lane; (c) all lanes of a wave into ONE block
// (worst case); (d) plain ds_write_b64 of the same addresses as a reference.  One 512-thread workgroup per CU.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics tools/exp/lds_atomic_f64.hip -o tools/exp/lds_atomic_f64.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int NSLOT = 330, LD = 37;
template <int MODE>
__global__ __launch_bounds__(512) void k(const int* __restrict__ slots, double* out, int iters) {
    __shared__ double acc[NSLOT * LD + 64];
    const int t = threadIdx.x;
    for (int e = t; e < NSLOT * LD + 64; e += 512) acc[e] = 0.0;
    __syncthreads();
    double v = 1.0 + t * 1e-3;
    for (int it = 0; it < iters; ++it) {
        const int slot = (MODE == 2) ? ((it * 7 + (t >> 6)) % NSLOT) : slots[(it * 512 + t) & 65535];
        double* blk = acc + slot * LD;
        if (MODE == 0) {
#pragma unroll
            for (int k2 = 0; k2 < 36; ++k2) unsafeAtomicAdd(&acc[((t + 64 * k2) % (NSLOT * LD))], v);     // distinct addresses, consecutive lanes
        } else if (MODE == 4 || MODE == 5) {
            unsigned long long* ua = reinterpret_cast<unsigned long long*>(MODE == 4 ? blk : acc);
            const unsigned long long uv = (unsigned long long)(long long)(v * 1048576.0);
#pragma unroll
            for (int k2 = 0; k2 < 36; ++k2) atomicAdd(MODE == 4 ? &ua[k2] : &ua[(t + 64 * k2) % (NSLOT * LD)], uv);   // ds_add_u64
        } else if (MODE == 3) {
#pragma unroll
            for (int k2 = 0; k2 < 36; ++k2) blk[k2] = v + k2;                                             // plain stores, Schur addresses
        } else {
#pragma unroll
            for (int k2 = 0; k2 < 36; ++k2) unsafeAtomicAdd(&blk[k2], v);
        }
    }
    __syncthreads();
    double s = 0; for (int e = t; e < NSLOT * LD; e += 512) s += acc[e];
    out[blockIdx.x * 512 + t] = s;
}
template <int MODE> void run(const char* name, const int* d_slots, double* d_out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d_slots, d_out, 4);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d_slots, d_out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = 256.0 * 512 * 36.0 * iters;
    printf("%-46s %8.3f ms  %7.1f G lane-ops/s  = %5.2f lane-ops per cycle per CU (2.4 GHz)\n", name, ms, ops / ms / 1e6, ops / (ms * 1e-3) / 256 / 2.4e9);
}
int main() {
    std::vector<int> h(65536);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (int)((s >> 8) % NSLOT); }
    int* d_slots; double* d_out;
    hipMalloc((void**)&d_slots, h.size() * 4); hipMalloc((void**)&d_out, 256 * 512 * 8);
    hipMemcpy(d_slots, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("ds_add_f64, distinct addresses, lane-consecutive", d_slots, d_out, 400);
        run<1>("ds_add_f64, Schur pattern (random 6x6 block/lane)", d_slots, d_out, 400);
        run<2>("ds_add_f64, one block per wave (all lanes collide)", d_slots, d_out, 100);
        run<3>("ds_write_b64, Schur addresses (reference)", d_slots, d_out, 400);
        run<4>("ds_add_u64, Schur pattern", d_slots, d_out, 400);
        run<5>("ds_add_u64, distinct addresses, lane-consecutive", d_slots, d_out, 400);
    }
    return 0;
}

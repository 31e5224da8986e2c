// dense_chol.hip -- blocked right-looking Cholesky of the dense reduced camera system on gfx950,
// FP64, with the trailing update on the FP64 matrix cores (v_mfma_f64_16x16x4_f64).
//
// Replaces, on the hot path, what Ceres' SPARSE_SCHUR / g2o's LinearSolverCSparse do with the
// reduced camera matrix (st20-g2o/src/include/test_ceres.h:145, test_g2o.h:95-100) and the
// in-tree `hMat.ldlt().solve(gMat)` (st17-ceres/src/include/solver.hpp:438,
// st3-calibration/src/src/calib.cpp:393).
//
// Layout: A is lda x lda row-major, lda a multiple of NB = 128 with at least one spare row.
// Only the lower triangle is referenced.  The LAST row (lda-1) carries rhs^T: because the panel
// solve and the trailing update are applied to every row below the diagonal block, that row is
// forward-substituted for free (it ends up holding y = L^-1 rhs).  Only the backward
// substitution L^T x = y needs its own kernels.
//
// Production schedule: ONE persistent kernel per factorisation (chol_mega_kernel): the four stages of
// a blocked right-looking Cholesky are tasks of a static dataflow graph, pulled from per-XCD ticket
// queues by one 512-thread workgroup per CU and synchronised with device-side flags:
//   D   diag_block     the 128x128 diagonal block in MFMA accumulator layout, 8-column block steps:
//                      row threads run the sqrt/scale chain, the other waves the rank-8 MFMA updates
//   T   trsm_task512   panel solve of 128 rows on the matrix cores (transposed form: an accumulator
//                      tile is the next B operand), L11 staged in LDS in operand order
//   TU  tu_task512     the critical hand-off: panel solve of block row b+1 fused with the update of
//                      the next diagonal tile
//   U   syrk_tile512   128x128 trailing tile -= P_i P_j^T, K = 128 through double-buffered LDS in
//                      fragment order (conflict-free ds_read_b64 / ds_write_b64); Uq: 32-row pieces
// Diagnostic schedule (stba_cholesky_profile): the same stages as one kernel each
// (chol_diag_kernel / chol_trsm_kernel / chol_syrk_kernel), timed per class with hipEvents.
// Backward substitution: one small kernel per block (GEMV with the inverse transposes the panel
// solve produced).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <queue>
#include <string>
#include <vector>

#include "common.hpp"

namespace stba {

constexpr int NB = CHOL_NB;
#ifndef MEGA_WAIT_SLEEP
#define MEGA_WAIT_SLEEP 1          // (the waits INSIDE a task -- for the diagonal block, for the siblings: 4 measured the same)
#endif
#ifndef MEGA_POLL_SLEEP
#define MEGA_POLL_SLEEP 16         // x 64 cycles between two looks at a parked ticket's flags.  Round 5, n = 6000, medians of six interleaved runs against 8:
                                   // 2: +0.7 %, 4: +0.4 %, 16: -1.0 / -0.2 %, 24: -0.8 %, 32: -0.3 %, 48: -0.2 % -- fewer polls, less traffic in front of the loads
#endif

typedef double double4v __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
__global__ void chol_pad_kernel(double* __restrict__ A, int lda, int n, const double* __restrict__ rhs) {
    // rows [n, lda): zero, unit diagonal; last row: rhs^T
    const int r = n + blockIdx.x;
    if (r >= lda) return;
    double* row = A + (size_t)r * lda;
    const bool last = (r == lda - 1);
    for (int c = threadIdx.x; c < lda; c += blockDim.x) {
        double v = 0.0;
        if (c == r) v = 1.0;
        else if (last && c < n && rhs) v = rhs[c];
        row[c] = v;
    }
}

int chol_prepare_padding_dev(double* A_dev, int lda, int n, const double* rhs_dev, hipStream_t st) {
    hipLaunchKernelGGL(chol_pad_kernel, dim3(lda - n), dim3(256), 0, st, A_dev, lda, n, rhs_dev);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

// ------------------------------------------------------------------------------------------
// fast reciprocal square root: v_rsq_f64 + ONE third-order (Halley) correction.  With e = 1 - d y0^2,
// 1/sqrt(d) = y0 (1 - e)^(-1/2) = y0 (1 + e/2 + 3e^2/8 + O(e^3)): four dependent FMA-class operations
// after the seed instead of six for two Newton steps (this sits on the pivot chain of the diagonal
// block, 6000 times per factorisation), and cubic convergence takes the >= 2^-20 seed below 2^-60.
__device__ __forceinline__ double fast_rsqrt(double d) {
    const double y = __builtin_amdgcn_rsq(d);
    const double e = fma(-d * y, y, 1.0);
    const double p = fma(0.375, e, 0.5);
    return fma(y * e, p, y);
}
__device__ __forceinline__ double sqrt_from_rsqrt(double d, double y) {
    double g = d * y;
    return fma(fma(-g, g, d), 0.5 * y, g);
}

// ------------------------------------------------------------------------------------------
// Global-memory stores.  WT = true: agent-scope relaxed atomic stores, i.e. stores with the sc1 bit,
// which are written through the XCD-private L2 to memory.  The persistent kernel uses them for
// FINAL data (finished blocks of L), which workgroups on other XCDs read; everything else in it
// stays in the L2 of the XCD that owns the tile (see chol_mega_kernel).  No cache-wide
// buffer_wbl2 / buffer_inv fences anywhere: with one fence pair per task hand-off the factorisation
// took 14 ms instead of 4.6 ms (every fence flushes and invalidates a whole L2).
template <bool WT> __device__ __forceinline__ void gst(double* p, double v) {
    if constexpr (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool WT> __device__ __forceinline__ void gst4(double* p, double4v v) {
    if constexpr (WT) {
        // two 16-byte write-through stores (the compiler only emits sc1 on <= 8-byte atomic stores; the
        // caller's s_waitcnt vmcnt(0) before raising the flag covers these)
        typedef double double2v __attribute__((ext_vector_type(2)));
        const double2v lo = {v[0], v[1]}, hi = {v[2], v[3]};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1"
                     :: "v"(p), "v"(lo), "v"(hi) : "memory");
    } else *reinterpret_cast<double4v*>(p) = v;
}
__device__ __forceinline__ double4v gld4(const double* p) { return *reinterpret_cast<const double4v*>(p); }
// ------------------------------------------------------------------------------------------
#define PHASE_STAMP(k) do { if (ph && t == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ph[k] = wall_clock64(); } } while (0)
// workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() also waits for its
// outstanding GLOBAL stores (s_waitcnt vmcnt(0)), which would put a store wave's ~1500-cycle global
// writes on the critical chain of the waves that wait for it at the barrier
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ------------------------------------------------------------------------------------------
// Diagonal block (128 x 128, 512 threads): everything on the matrix cores, 4 columns per step.  (A first design -- 128 row
// threads factoring 8 x 8 mini-blocks redundantly on the VALU, four waves of rank-8 MFMA updates -- took 27.8 us inside the
// persistent kernel against 21.3 us for this one and is gone; DESIGN.md 4 has its anatomy.)
//
// Measured on MI355X (tools/exp/lat_f64.hip): a dependent v_fma_f64 costs 4.2 cycles (= its issue time), the whole
// pivot step rsq + Halley + scale + update 50 cycles, a dependent v_mfma_f64_16x16x4 66-81 cycles, an LDS write ->
// barrier -> read hand-off 150-190, a POLLED LDS hand-off 250-500.  The first design spent 250 cycles per pivot
// because every row thread factored an 8x8 mini-block redundantly (FP64 latency = issue time, so redundant work IS
// latency) and because FP64 MFMAs of the SIMD partner stall a wave's FP64 VALU.  Here:
//   * the 128x128 block is 8x8 tiles of 16x16, lower tiles only, each tile in the accumulator layout of
//     v_mfma_f64_16x16x4_f64 holding the TRANSPOSE: lane (n, g) register r = M[n][4r + g].  Register s of a tile
//     is then directly the B operand "columns 4s..4s+3 of the tile" (lane (n, g) = M[n][4s + g]).
//   * per 4-column step the FACTOR wave broadcasts the 4x4 pivot block of its diagonal tile with v_readlane,
//     factors it and inverts the factor on the VALU (uniform in all lanes, branch-free), forms the operand
//     Gp = [Ginv; 0] (lane (n, g) = Ginv[n][g], n < 4) and then
//                l = mfma(Gp, tile.reg[s])[0]        lane (n, g) = L[n][4s + g]   (panel: 1 MFMA)
//                tile = mfma(-l, l, tile)             rank-4 update                (1 MFMA)
//     Every other tile (I, J') does the same two MFMAs with Gp and the l operands of rows I and J' taken from LDS:
//     tile(I, J') = mfma(-l_J', l_I, tile(I, J')).
//   * every Gp and every l has its own slot in LDS, written once.  A step has three phases separated by two
//     s_barriers: (1) the factor wave makes Gp | (2) every row computes and publishes its l | (3) trailing updates.
//     The updates of step t overlap the factor wave's VALU work of step t + 1.
//   * roles (waves w and w + 4 share a SIMD):
//       wave 0   the factor wave.  Nothing else runs FP64 MFMAs on its SIMD: wave 4 only moves the finished tile columns
//                of L from LDS to memory.
//       wave 1   the FOLLOWER: it carries the two tiles (J+1, J), (J+1, J+1) of the next row through tile column J and
//                hands the finished diagonal tile to wave 0 through LDS.
//       waves 5 | 2, 6 | 3, 7   bulk: tile rows {5, 2} | {7}, {3} | {6}, {4} through tile column I-2 (14 MFMAs per step
//                and SIMD in the first tile column, fewer later); the two tiles of row I that the follower needs then
//                migrate to it through LDS.  Waves 6 / 7 also compute the inverses of the finished diagonal tiles of the
//                even / odd rows (the same two MFMAs on an identity tile; the panel-solve tasks multiply by them).
struct Diag2Smem {
    double Lsl[36][4][64];          // l operands: tile (I, J) at I(I+1)/2 + J, step s, lane
    double Gp[32][64];              // Gp operands: global step 4J + s, lane
    double Mig[2][2][4][64];        // tiles (I, I-1) and (I, I) of row I on their way to the follower, slot I & 1: register r, lane
    double Dg[4][64];               // the next diagonal tile on its way from the follower to the factor wave
};
static_assert(sizeof(Diag2Smem) <= 128 * 1024, "D2 must fit the persistent kernel's LDS");


__device__ __forceinline__ double readlane_f64(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane), hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double mfma_l(double gp, double b) {       // l = (Gp . tile.reg[s])[0]
    const double4v z = {0.0, 0.0, 0.0, 0.0};
    return __builtin_amdgcn_mfma_f64_16x16x4f64(gp, b, z, 0, 0, 0)[0];
}
__device__ __forceinline__ int d2_tix(int I, int J) { return I * (I + 1) / 2 + J; }

// tile (I, J) of the block -> registers; the diagonal tiles are made symmetric from their lower triangle
// The block enters through LDS.  Read tile by tile straight into the accumulator layout, a load instruction touches 16
// rows at 32 bytes each and every 128-byte line four times: the factor wave, its own tile long loaded, stood 3.1 us at
// the first barrier waiting for the bulk rows' 32 such instructions per lane (tools/mega_trace.py, stamps behind the
// first two barriers).  Here all 512 threads fetch the lower-triangle tiles with full-line requests (8 lanes x 16 B per
// row) and drop them into the -- still empty -- slots of the panel values, tile (I, J) at Lsl[tix(I, J)], element
// (row a, column c) at c * 16 + (a + c') mod 16, c' = c with bit 0 cleared: the accumulator layout then reads as four
// contiguous 512-byte rows per tile (each rotated by a constant: conflict-free), and the rotation spreads the staging
// writes, whose lanes differ in c' by multiples of two, over the banks.
__device__ __forceinline__ void d2_stage_block(const double* __restrict__ Ab, int lda, Diag2Smem& sm, int t) {
    const int h = t & 7, a8 = (t >> 3) & 7, w = t >> 6;            // 16-byte piece of the row, row within the half tile
    double2 v[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int item = k * 8 + w;                                // 72 half tiles over 8 waves
        const int tile = item >> 1, a = (item & 1) * 8 + a8;
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= tile) ++I;
        const int J = tile - I * (I + 1) / 2;
        v[k] = *reinterpret_cast<const double2*>(&Ab[(size_t)(16 * I + a) * lda + 16 * J + 2 * h]);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int item = k * 8 + w;
        const int tile = item >> 1, a = (item & 1) * 8 + a8;
        double* dst = &sm.Lsl[tile][0][0];
        dst[(2 * h) * 16 + ((a + 2 * h) & 15)] = v[k].x;
        dst[(2 * h + 1) * 16 + ((a + 2 * h) & 15)] = v[k].y;
    }
}

// tile (I, J) of the staged block -> registers; the diagonal tiles are made symmetric from their lower triangle
__device__ __forceinline__ double4v d2_load_tile(const Diag2Smem& sm, int I, int J, int n, int g) {
    const double* src = &sm.Lsl[d2_tix(I, J)][0][0];
    double4v v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = 4 * r + g;
        const int row = (I == J && c > n) ? c : n, col = (I == J && c > n) ? n : c;
        v[r] = src[col * 16 + ((row + (col & ~1)) & 15)];
    }
    return v;
}

// final L tile (I, J): LDS slots -> global, 32 contiguous bytes per lane (lane (n, g') writes L[n][4g' .. 4g'+3])
template <bool WT>
__device__ __forceinline__ void d2_store_tile(double* __restrict__ A, int lda, int k0, const Diag2Smem& sm, int I, int J, int lane) {
    const int n = lane & 15, gq = lane >> 4;
    const int tix = d2_tix(I, J);
    double4v v;
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
        v[gg] = sm.Lsl[tix][gq][n + 16 * gg];
        if (I == J && 4 * gq + gg > n) v[gg] = 0.0;          // right of the diagonal: not part of L
    }
    gst4<WT>(A + (size_t)(k0 + 16 * I + n) * lda + k0 + 16 * J + 4 * gq, v);
}

// one factor step on the diagonal tile accD (see above): pivot block s -> Gp operand
__device__ __forceinline__ double d2_factor_gp(const double4v& accD, int s, const double (&mk)[10], int& badv, int col0, int n_real) {
    const double b = accD[s];
    // the 4x4 pivot block M[4s+a][4s+c] sits in lane (n = 4s+a, g = c), register s
    const double a00 = readlane_f64(b, 4 * s + 0);
    const double a10 = readlane_f64(b, 4 * s + 1), a11 = readlane_f64(b, 4 * s + 1 + 16);
    const double a20 = readlane_f64(b, 4 * s + 2), a21 = readlane_f64(b, 4 * s + 2 + 16), a22 = readlane_f64(b, 4 * s + 2 + 32);
    const double a30 = readlane_f64(b, 4 * s + 3), a31 = readlane_f64(b, 4 * s + 3 + 16), a32 = readlane_f64(b, 4 * s + 3 + 32),
                 a33 = readlane_f64(b, 4 * s + 3 + 48);
    // a failed pivot of a real column is recorded (reported at the end) and replaced by 1; no branches
    auto pivot = [&](double d, int c) -> double {
        const bool ok = d > 0.0;
        badv = (!ok && col0 + c < n_real && badv == 0) ? col0 + c + 1 : badv;
        return fast_rsqrt(ok ? d : 1.0);
    };
    const double y0 = pivot(a00, 0);
    const double g10 = a10 * y0, g20 = a20 * y0, g30 = a30 * y0;
    const double y1 = pivot(fma(-g10, g10, a11), 1);
    const double g21 = fma(-g20, g10, a21) * y1, g31 = fma(-g30, g10, a31) * y1;
    const double y2 = pivot(fma(-g21, g21, fma(-g20, g20, a22)), 2);
    const double g32 = fma(-g31, g21, fma(-g30, g20, a32)) * y2;
    const double y3 = pivot(fma(-g32, g32, fma(-g31, g31, fma(-g30, g30, a33))), 3);
    // inverse of the 4x4 factor
    const double i10 = -(g10 * y0) * y1, i21 = -(g21 * y1) * y2, i32 = -(g32 * y2) * y3;
    const double i20 = -fma(g21, i10, g20 * y0) * y2, i31 = -fma(g32, i21, g31 * y1) * y3;
    const double i30 = -fma(g32, i20, fma(g31, i10, g30 * y0)) * y3;
    // Gp: lane (n, g) = Ginv[n][g] (n < 4), as a sum with per-lane 0/1 weights (exact; no divergent code),
    // the entries that depend on the last pivot last
    double gp = mk[0] * y0;
    gp = fma(mk[1], i10, gp); gp = fma(mk[2], y1, gp); gp = fma(mk[3], i20, gp); gp = fma(mk[4], i21, gp);
    gp = fma(mk[5], y2, gp); gp = fma(mk[6], i30, gp); gp = fma(mk[7], i31, gp); gp = fma(mk[8], i32, gp);
    gp = fma(mk[9], y3, gp);
    return gp;
}

// Every role runs the same schedule of barriers: ONE per 4-column step.  Behind barrier t = (tile column J, step s):
//   * the factor wave, whose Gp(t) was published before the barrier, computes its panel values l(t), updates its
//     tile and goes straight on to the pivot block of step t + 1 (published before barrier t + 1);
//   * every other row first applies the rank-4 update of step t - 1 (all its operands were published before this
//     barrier), then computes and publishes l(t) = mfma(Gp(t), tile.reg[s]).
// So the updates run one step behind, no wave ever needs a second barrier inside a step, and the factor wave's
// chain per step is: panel MFMA, update MFMA, pivot block on the VALU, one LDS store, one barrier.
// (One function per role: their registers are allocated independently; a single loop with all roles inside needed
// 256 registers and spilled.)
#define D2_BARRIER() lds_barrier()      // LDS traffic only: a wave's global stores must not hold the barrier up

// the factor wave
__device__ __forceinline__ void d2_factor_wave(const double* __restrict__ Ab, int lda, int k0, int n_real, int* __restrict__ flag,
                                               Diag2Smem& sm, int lane, long long* ph) {
    const int n = lane & 15, g = lane >> 4;
    double4v accD = d2_load_tile(sm, 0, 0, n, g);
    double mk[10];          // 0/1 weights of the ten entries of the 4x4 inverse for this lane's operand slot
    {
        const int idx = (n < 4 && g <= n) ? n * (n + 1) / 2 + g : -1;
#pragma unroll
        for (int e = 0; e < 10; ++e) mk[e] = (idx == e) ? 1.0 : 0.0;
    }
    int badv = 0;
    double gp = d2_factor_gp(accD, 0, mk, badv, k0, n_real);
    sm.Gp[0][lane] = gp;
    if (ph && lane == 0) ph[1] = wall_clock64();
#pragma unroll 1
    for (int J = 0; J < 8; ++J) {
        if (ph && lane == 0 && (J == 1 || J == 4)) ph[J == 1 ? 2 : 3] = wall_clock64();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int tt = 4 * J + s;
            D2_BARRIER();
            const double l = mfma_l(gp, accD[s]);
            sm.Lsl[d2_tix(J, J)][s][lane] = l;
            if (s < 3) {
                accD = __builtin_amdgcn_mfma_f64_16x16x4f64(-l, l, accD, 0, 0, 0);
                gp = d2_factor_gp(accD, s + 1, mk, badv, k0 + 16 * J + 4 * (s + 1), n_real);
                sm.Gp[tt + 1][lane] = gp;
            }
        }
        if (J < 7) {
            D2_BARRIER();                               // X: the follower has published the next diagonal tile
#pragma unroll
            for (int r = 0; r < 4; ++r) accD[r] = sm.Dg[r][lane];
            gp = d2_factor_gp(accD, 0, mk, badv, k0 + 16 * (J + 1), n_real);
            sm.Gp[4 * J + 4][lane] = gp;
        }
    }
    D2_BARRIER();                                       // (the panel values of the last step are published)
    if (badv != 0 && lane == 0) atomicCAS(flag, 0, badv);
}

// the follower: row J + 1 through tile column J
__device__ __forceinline__ void d2_follower_wave(const double* __restrict__ Ab, int lda, Diag2Smem& sm, int lane) {
    const int n = lane & 15, g = lane >> 4;
    double4v accS = d2_load_tile(sm, 1, 0, n, g), accD = d2_load_tile(sm, 1, 1, n, g);
    double lprev = 0.0;
#pragma unroll 1
    for (int J = 0; J < 8; ++J) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int tt = 4 * J + s;
            D2_BARRIER();
            if (J < 7) {
                if (s == 0 && J >= 1) {
                    // the two tiles of row J + 1 arrive from the bulk wave that carried them, with every update but the
                    // last one of the previous tile column: that one is applied here (lprev: row J's values of step
                    // (J-1, 3), this wave's own)
                    const double li = sm.Lsl[d2_tix(J + 1, J - 1)][3][lane];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { accS[r] = sm.Mig[(J + 1) & 1][0][r][lane]; accD[r] = sm.Mig[(J + 1) & 1][1][r][lane]; }
                    accS = __builtin_amdgcn_mfma_f64_16x16x4f64(-lprev, li, accS, 0, 0, 0);
                    accD = __builtin_amdgcn_mfma_f64_16x16x4f64(-li, li, accD, 0, 0, 0);
                }
                if (s >= 1) accS = __builtin_amdgcn_mfma_f64_16x16x4f64(-sm.Lsl[d2_tix(J, J)][s - 1][lane], lprev, accS, 0, 0, 0);
                const double l = mfma_l(sm.Gp[tt][lane], accS[s]);
                sm.Lsl[d2_tix(J + 1, J)][s][lane] = l;
                accD = __builtin_amdgcn_mfma_f64_16x16x4f64(-l, l, accD, 0, 0, 0);
                lprev = l;
                if (s == 3) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sm.Dg[r][lane] = accD[r];
                }
            }
        }
        if (J < 7) D2_BARRIER();                        // X
    }
    D2_BARRIER();
}

// Bulk rows.  A wave carries ONE tile row R with its tiles indexed by their distance from the diagonal: acc[d] = tile
// (R, R - d).  In tile column J = R - C the live tiles are d = 0 .. C and the leading tile is acc[C]: with C a template
// parameter every register index is static, so a tile column is straight-line code with C + 2 MFMAs per step and not
// one branch; the code of the columns C = 7 .. 2 is shared by all rows (row R enters at C = R and leaves after C = 2,
// when its last two tiles migrate to the follower).  Per step, behind the barrier:
//   * all LDS operands are fetched in one batch (the loads must not sit behind the wave's own LDS stores, which the
//     compiler cannot tell apart from them);
//   * the LEADING tile gets the previous step's update first, and the row's panel values of this step are published
//     at once (every row below waits for them at the next barrier);
//   * then the rest of the previous step's update (nothing waits for it before the next step).
template <int C>
__device__ __forceinline__ void d2_bulk_column(Diag2Smem& sm, int R, double4v (&acc)[8], double& lprev, int lane) {
    const int J = R - C;
    // LDS slots of the operands: l of row R - d in the current tile column (for steps 1..3) and in the previous one
    const double* ljp[C + 1];
#pragma unroll
    for (int d = 1; d <= C; ++d) ljp[d] = &sm.Lsl[d2_tix(R - d, J)][0][lane];
    double* mine = &sm.Lsl[d2_tix(R, J)][0][lane];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        D2_BARRIER();
        const bool upd = s > 0 || J > 0;        // (the very first step has no previous one)
        double lj[C + 1];
        const double gp = sm.Gp[4 * J + s][lane];
        if (upd) {
#pragma unroll
            for (int d = C; d >= 1; --d) lj[d] = (s > 0) ? ljp[d][(s - 1) * 64] : ljp[d][3 * 64 - 4 * 64];   // (previous column: one tile slot back)
            acc[C] = __builtin_amdgcn_mfma_f64_16x16x4f64(-lj[C], lprev, acc[C], 0, 0, 0);
        }
        const double l = mfma_l(gp, acc[C][s]);
        mine[s * 64] = l;
        if (s == 3) D2_BARRIER();               // X (early: the factor wave works on the next pivot block meanwhile)
        if (upd) {
#pragma unroll
            for (int d = C - 1; d >= 1; --d) acc[d] = __builtin_amdgcn_mfma_f64_16x16x4f64(-lj[d], lprev, acc[d], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-lprev, lprev, acc[0], 0, 0, 0);
        }
        lprev = l;
    }
    if (C == 2) {
        // the row's last bulk column: its two remaining tiles leave for the follower WITHOUT the update of step 3, which
        // the follower applies itself
#pragma unroll
        for (int r = 0; r < 4; ++r) { sm.Mig[R & 1][0][r][lane] = acc[1][r]; sm.Mig[R & 1][1][r][lane] = acc[0][r]; }
    }
}

// tile row R (2 .. 7) through the tile columns 0 .. R - 2; R + 1 barriers short of ... no: 5 barriers per tile column
__device__ __forceinline__ void d2_bulk_row(const double* __restrict__ Ab, int lda, Diag2Smem& sm, int lane, int R) {
    const int n = lane & 15, g = lane >> 4;
    double4v acc[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) if (d <= R) acc[d] = d2_load_tile(sm, R, R - d, n, g);
    double lprev = 0.0;
    switch (R) {
        case 7: d2_bulk_column<7>(sm, R, acc, lprev, lane); [[fallthrough]];
        case 6: d2_bulk_column<6>(sm, R, acc, lprev, lane); [[fallthrough]];
        case 5: d2_bulk_column<5>(sm, R, acc, lprev, lane); [[fallthrough]];
        case 4: d2_bulk_column<4>(sm, R, acc, lprev, lane); [[fallthrough]];
        case 3: d2_bulk_column<3>(sm, R, acc, lprev, lane); [[fallthrough]];
        default: d2_bulk_column<2>(sm, R, acc, lprev, lane);
    }
}

// bulk wave: its row, then -- idle otherwise -- the inverses of finished diagonal tiles: the same two MFMAs on an identity
// tile below the diagonal tile; its panel values are X = L^-T, lane (n, g) of step s = X[n][4s+g] = Inv[4s+g][n].
// einv 0: the inverse of row J - 1 during every tile column J the wave is idle in (rows 1 .. 6 for tile row 3);
// einv 1: row 0 during the first idle column, row 7 at the end.
template <bool WT>
__device__ __forceinline__ void d2_bulk_wave(const double* __restrict__ Ab, int lda, Diag2Smem& sm, int lane, int R, int einv,
                                             double* __restrict__ dinv) {
    const int n = lane & 15, g = lane >> 4;
    d2_bulk_row(Ab, lda, sm, lane, R);
    double4v accE;
    auto inverse_step = [&](int I, int s) {
        const double le = mfma_l(sm.Gp[4 * I + s][lane], accE[s]);
        gst<WT>(&dinv[(I * 16 + 4 * s + g) * 16 + n], le);
        if (s < 3) accE = __builtin_amdgcn_mfma_f64_16x16x4f64(-sm.Lsl[d2_tix(I, I)][s][lane], le, accE, 0, 0, 0);
    };
#pragma unroll 1
    for (int J = R - 1; J < 8; ++J) {
        const int Ie = einv == 0 ? J - 1 : (einv == 1 && J == R - 1) ? 0 : -1;
        if (Ie >= 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) accE[r] = (n == 4 * r + g) ? 1.0 : 0.0;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            D2_BARRIER();
            if (Ie >= 0) inverse_step(Ie, s);
        }
        if (J < 7) D2_BARRIER();                        // X
    }
    D2_BARRIER();
    if (einv == 1) {                            // the last diagonal tile's inverse (row 7)
#pragma unroll
        for (int r = 0; r < 4; ++r) accE[r] = (n == 4 * r + g) ? 1.0 : 0.0;
#pragma unroll
        for (int s = 0; s < 4; ++s) inverse_step(7, s);
    }
}

// wave 4, the SIMD mate of the factor wave: tile row 2 during the first tile column (4 MFMAs per step next to the
// factor wave, which is not the bottleneck there), then all global stores of L.  The tile column finished one column
// ago goes to memory two tiles per step (all at once would make this wave late for the next barrier, and every wave
// waits there).
template <bool WT>
__device__ __forceinline__ void d2_store_wave(double* __restrict__ A, const double* __restrict__ Ab, int lda, int k0, Diag2Smem& sm, int lane) {
    d2_bulk_row(Ab, lda, sm, lane, 2);
#pragma unroll 1
    for (int J = 1; J < 8; ++J) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            D2_BARRIER();
            // (behind barrier (J, 0) every panel value of tile column J - 1 has been published)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int Ip = (J - 1) + 2 * s + h;
                if (Ip < 8) d2_store_tile<WT>(A, lda, k0, sm, Ip, J - 1, lane);
            }
        }
        if (J < 7) D2_BARRIER();                        // X
    }
    D2_BARRIER();
    d2_store_tile<WT>(A, lda, k0, sm, 7, 7, lane);      // the last tile column
}

// all 512 threads of the workgroup; smem = sizeof(Diag2Smem); ends with the results in flight to global memory
// (the caller waits for them: s_waitcnt vmcnt(0) + barrier)
template <bool WT>
__device__ __forceinline__ void diag_block2(double* __restrict__ A, int lda, int k0, int n_real,
                                            int* __restrict__ flag, double* __restrict__ dinv, double* smem, int t,
                                            long long* ph = nullptr) {
    Diag2Smem& sm = *reinterpret_cast<Diag2Smem*>(smem);
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const double* Ab = A + (size_t)k0 * lda + k0;
    PHASE_STAMP(0);
    d2_stage_block(Ab, lda, sm, t);
    D2_BARRIER();
    if (w == 0) d2_factor_wave(Ab, lda, k0, n_real, flag, sm, lane, ph);
    else if (w == 1) d2_follower_wave(Ab, lda, sm, lane);
    else if (w == 4) d2_store_wave<WT>(A, Ab, lda, k0, sm, lane);
    else d2_bulk_wave<WT>(Ab, lda, sm, lane, w == 5 ? 5 : w == 2 ? 7 : w == 6 ? 3 : w == 3 ? 6 : 4, w == 6 ? 0 : w == 7 ? 1 : -1, dinv);
}

// the diagonal-block task as a kernel of its own (stage-kernel schedule); dynamic LDS: sizeof(Diag2Smem)
__global__ __launch_bounds__(512) void chol_diag_kernel(double* __restrict__ A, int lda, int k0, int n_real,
                                                         int* __restrict__ flag, double* __restrict__ dinv) {
    extern __shared__ __attribute__((aligned(16))) double sm2[];
    diag_block2<false>(A, lda, k0, n_real, flag, dinv, sm2, threadIdx.x);
}

// ------------------------------------------------------------------------------------------
// Panel solve X = A21 * L11^-T on the matrix cores, one wave per 16 panel rows, computed in the
// transposed form  Y = X^T = L11^-1 * A21^T  so that a finished 16x16 tile Y_I, sitting in the
// accumulator layout, IS the B operand of the next v_mfma_f64_16x16x4_f64 (k-step q <-> register q):
// no LDS, no shuffles, no barriers.
//   for I = 0..7:   Y_I  = Inv_II * W_I                      (4 MFMAs; Inv_II from the diagonal kernel)
//                   W_J -= L_JI * Y_I   for J > I            (4 MFMAs per tile, independent chains)
// Row order inside a tile: accumulator register r of lane group g holds LOGICAL row 4g + r (physical
// MFMA row g + 4r), so a lane's four registers are four consecutive matrix columns of X: one 32 B
// load/store per lane per tile, and the A operands (L_JI, Inv_II) are one 32 B load per lane too.
// `ident`: the same solve on 16 rows of the identity, X = I * L11^-T = the inverse transpose of the
// diagonal block (rows >= nv of the block, i.e. the right-hand-side row, count as identity rows);
// the backward substitution multiplies by it.
template <bool WT>
__device__ __forceinline__ void trsm_group(double* __restrict__ rowp, bool ident, int ident_row0, int nv,
                                           const double* __restrict__ Lb, int lda,
                                           const double* __restrict__ dinv, int lane) {
    const int n = lane & 15, g = lane >> 4;
    const int pm = 4 * (n & 3) + (n >> 2);          // logical tile row this lane feeds as an A operand
    double4v W[8];
#pragma unroll
    for (int J = 0; J < 8; ++J) {
        if (ident) {
#pragma unroll
            for (int r = 0; r < 4; ++r) W[J][r] = (16 * J + 4 * g + r == ident_row0 + n) ? 1.0 : 0.0;
        } else {
            W[J] = gld4(rowp + 16 * J + 4 * g);
        }
    }
    auto load_l = [&](int J, int I) -> double4v {
        const int row = 16 * J + pm;
        double4v v = gld4(Lb + (size_t)row * lda + 16 * I + 4 * g);
        if (row >= nv) v = double4v{0.0, 0.0, 0.0, 0.0};
        return -v;
    };
    auto load_inv = [&](int J) -> double4v {
        double4v v = gld4(dinv + (J * 16 + pm) * 16 + 4 * g);
        if (16 * J + pm >= nv) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = (4 * g + q == pm) ? 1.0 : 0.0;
        }
        return v;
    };
    double4v Lc[8], Ln[8], iv, ivn;
    iv = load_inv(0);
#pragma unroll
    for (int J = 1; J < 8; ++J) Lc[J] = load_l(J, 0);
#pragma unroll
    for (int I = 0; I < 8; ++I) {
        if (I + 1 < 8) {
            ivn = load_inv(I + 1);
#pragma unroll
            for (int J = I + 2; J < 8; ++J) Ln[J] = load_l(J, I + 1);
        }
        double4v y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q) y = __builtin_amdgcn_mfma_f64_16x16x4f64(iv[q], W[I][q], y, 0, 0, 0);
        W[I] = y;
#pragma unroll
        for (int J = I + 1; J < 8; ++J)
#pragma unroll
            for (int q = 0; q < 4; ++q) W[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(Lc[J][q], y[q], W[J], 0, 0, 0);
        iv = ivn;
#pragma unroll
        for (int J = I + 2; J < 8; ++J) Lc[J] = Ln[J];
    }
#pragma unroll
    for (int J = 0; J < 8; ++J) gst4<WT>(rowp + 16 * J + 4 * g, W[J]);
}

// one wave per workgroup: groups [0, n_groups) are the rows below the block, the last 8 the identity
__global__ __launch_bounds__(64) void chol_trsm_kernel(double* __restrict__ A, int lda, int k0, int n_groups,
                                                       double* __restrict__ Xinv, const double* __restrict__ dinv) {
    const int lane = threadIdx.x, grp = blockIdx.x;
    const bool ident = grp >= n_groups;
    const int e = grp - n_groups;
    const int nv = ident ? min(NB, (lda - 1) - k0) : NB;
    double* rowp = ident ? Xinv + (size_t)(16 * e + (lane & 15)) * NB
                         : A + (size_t)(k0 + NB + 16 * grp + (lane & 15)) * lda + k0;
    trsm_group<false>(rowp, ident, 16 * e, nv, A + (size_t)k0 * lda + k0, lda, dinv, lane);
}

// ------------------------------------------------------------------------------------------
// Trailing update C(i,j) -= P_i P_j^T over the lower-triangle 128x128 tiles, P = panel columns
// [k0, k0+128).  256 threads = 4 waves in a 2x2 grid, each wave 64x64 = 4x4 MFMA tiles.
// Tile shapes: TM x 128 outputs per workgroup (256 threads = 4 waves).
//   TM = 128: waves 2x2, each 64x64 = 4x4 MFMA tiles; 64 KB LDS -> 2 workgroups per CU.  Best for
//             very large grids (measured 89 % of the FP64 MFMA peak at 1128 tiles).
//   TM =  64: waves 1x4, each 64x32 = 4x2 MFMA tiles; 48 KB LDS -> 3 workgroups per CU, twice as
//             many (half-size) tiles: less tail quantisation and better phase overlap on the
//             mid-size and small trailing matrices that dominate the step count.
// tile_mode 0: every lower-triangle tile; 1: only the first tile column (the next panel, look-ahead);
// 2: everything except the first tile column.
template <int TM>
__global__ __launch_bounds__(256) void chol_syrk_kernel(double* __restrict__ A, int lda, int k0, int tile_mode) {
    constexpr int RBA = TM / 16;            // 16-row blocks of the A tile
    constexpr int WR = TM / 64;             // wave grid rows (2 or 1)
    constexpr int WC = 4 / WR;              // wave grid cols (2 or 4)
    constexpr int MB = 4;                   // MFMA row blocks per wave (64 rows)
    constexpr int NBK = 8 / WC;             // MFMA col blocks per wave (4 or 2)
    constexpr int PA = TM / 64;             // staging passes for the A tile
    __shared__ __attribute__((aligned(16))) double sA[2][TM * 16];
    __shared__ __attribute__((aligned(16))) double sB[2][2048];
    const int sub = (TM == 64) ? (blockIdx.x & 1) : 0;
    const int id = (TM == 64) ? (blockIdx.x >> 1) : blockIdx.x;
    int ti, tj;
    if (tile_mode == 1) {
        ti = id; tj = 0;
    } else {
        ti = (int)((sqrt(8.0 * (double)id + 1.0) - 1.0) * 0.5);
        while (ti * (ti + 1) / 2 > id) --ti;
        while ((ti + 1) * (ti + 2) / 2 <= id) ++ti;
        tj = id - ti * (ti + 1) / 2;
        if (tile_mode == 2) { ++ti; ++tj; }
    }
    const int r0 = k0 + NB;
    const int row_i = r0 + ti * NB + sub * 64, row_j = r0 + tj * NB;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = w / WC, wc = w % WC;

    // accumulators start from C (C/D layout of v_mfma_f64_16x16x4_f64: col = lane&15,
    // row = (lane>>4) + 4*reg); the A fragment is negated, so the epilogue is a plain store
    double4v acc[MB][NBK];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NBK; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row_i + wr * (MB * 16) + m * 16 + (lane >> 4) + 4 * r;
                const int col = row_j + wc * (NBK * 16) + n * 16 + (lane & 15);
                acc[m][n][r] = A[(size_t)row * lda + col];
            }

    // staging map: pass p, half h -> row = (lane&15) + 16*(w + 4p), k = 2*((lane>>4) + 4h)
    double2 ga[PA][2], gb[2][2];
    const int lrow = lane & 15, lkp = lane >> 4;
    auto gload = [&](int kc) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = lrow + 16 * (w + 4 * p);
                const int k = 2 * (lkp + 4 * h);
                if (p < PA) ga[p < PA ? p : 0][h] = *reinterpret_cast<const double2*>(&A[(size_t)(row_i + row) * lda + k0 + kc * 16 + k]);
                gb[p][h] = *reinterpret_cast<const double2*>(&A[(size_t)(row_j + row) * lda + k0 + kc * 16 + k]);
            }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rb = w + 4 * p;                 // 16-row block index
                const int k = 2 * (lkp + 4 * h);
                if (p < PA) {
                    const int posa = (((k >> 2) * RBA + rb) << 6) + ((k & 3) << 4) + lrow;
                    sA[buf][posa] = ga[p < PA ? p : 0][h].x;
                    sA[buf][posa + 16] = ga[p < PA ? p : 0][h].y;   // k+1: (k&3) is even so +1 -> +16
                }
                const int posb = (((k >> 2) * 8 + rb) << 6) + ((k & 3) << 4) + lrow;
                sB[buf][posb] = gb[p][h].x;
                sB[buf][posb + 16] = gb[p][h].y;
            }
    };
    gload(0);
    lstore(0);
    __syncthreads();
    constexpr int KC = NB / 16;
    for (int kc = 0; kc < KC; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < KC) gload(kc + 1);
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            double a[MB], b[NBK];
#pragma unroll
            for (int m = 0; m < MB; ++m) a[m] = -sA[buf][((kq * RBA + wr * MB + m) << 6) + lane];
#pragma unroll
            for (int n = 0; n < NBK; ++n) b[n] = sB[buf][((kq * 8 + wc * NBK + n) << 6) + lane];
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NBK; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        if (kc + 1 < KC) lstore(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NBK; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row_i + wr * (MB * 16) + m * 16 + (lane >> 4) + 4 * r;
                const int col = row_j + wc * (NBK * 16) + n * 16 + (lane & 15);
                A[(size_t)row * lda + col] = acc[m][n][r];
            }
}

// picks the tile shape by grid size (tiles128 = number of 128x128 tiles of the launch)
static void launch_syrk(double* A, int lda, int k0, int tile_mode, int tiles128, hipStream_t st) {
    // (TM = 64 for every grid size met so far: measured; TM = 128 pays from ~1000 tiles per launch)
    const bool use64 = tiles128 < 1000000;
    if (use64) hipLaunchKernelGGL(chol_syrk_kernel<64>, dim3(2 * tiles128), dim3(256), 0, st, A, lda, k0, tile_mode);
    else hipLaunchKernelGGL(chol_syrk_kernel<128>, dim3(tiles128), dim3(256), 0, st, A, lda, k0, tile_mode);
}

// ------------------------------------------------------------------------------------------
// The factorisation as a persistent DATAFLOW program: a static task list in dataflow order, executed by workgroups that
// stay resident for the whole factorisation.  Launching the steps as separate kernels costs ~3 us between dependent
// kernels and ~12 us per cross-stream hand-off (measured), and a diagonal-block workgroup cannot start on a CU that still
// holds trailing-update workgroups; with 47 dependent steps that was half of the solve time.  Here a workgroup takes the
// next ticket of its list (atomic counter), waits on the device-side flags of that task's inputs, runs it, publishes its
// outputs (release fence + flag) and goes on.  Tickets are handed out in an order that is a topological order of the task
// graph, and a workgroup only ever waits for tasks that come earlier in that order -- which are finished, held by a
// resident workgroup, or the next ticket of some list -- so the program cannot deadlock as long as every list has a
// resident workgroup (MEGA_NTU = 10 for the lists that carry the TU tasks).
//   D(b)          diagonal block b                        needs ver[b][b] == 4b
//   TU(b, q)      q = 0..9: rows of block row b+1 of the panel solve AND a 2 x 2-tile block of the update of the next
//                 diagonal tile, fused (the critical hand-off D(b) -> D(b+1), see tu_task512)
//                                                         needs D(b), ver[b+1][b] == ver[b+1][b+1] == 4b
//   T(b, i)       panel solve of the 128 rows of tile row i > b+1     needs D(b), ver[i][b] == 4b
//                 (in the last 30 panels as two half tasks of 64 rows: see mega_build_tasks)
//   TI(b)         inverse transpose of block b (for the backward substitution)   needs D(b)
//   U(b; i, j)    tile (i, j) -= L_ib L_jb^T       needs T(b,i), T(b,j), ver[i][j] == 4b;  ver += 4
//   Uq(b; i,q,j)  the same for 32 rows of a tile of the NEXT panel's column (j = b+1)       ver += 1
// (Round 3 tried TWO CLASSES of workgroups -- 512-thread "chain" workgroups on a few CUs of every XCD for D / TU / T / Uq, and
// 256-thread trailing-update workgroups, two per CU, on all the others, the device split with CU-masked streams.  It ran,
// bit-exact, and LOST: 2.61-2.96 ms against 2.44 ms, whatever the split.  Two update workgroups on a CU each take 45 us
// per 128-column pass against 24 us alone: the update task is bound by what the memory system delivers to ALL CUs
// together (~5.5 TB/s past the L2s at 8.2 flop/B), not by exposed latency inside one CU, so a second workgroup adds
// nothing, and the critical updates queue behind far ones in the in-order bulk lists.  DESIGN.md 4 has the table.)
// XCD ownership.  MI355X has eight XCDs with private, mutually non-coherent L2 caches.  Every tile
// (i, j) is owned by one XCD (by tile row, see mega_row_owner), and every task that WRITES the tile runs
// on a workgroup of that XCD (a workgroup reads HW_REG_XCC_ID to find its lists).  A tile under update therefore lives
// in one L2 only and needs nothing but an L1 invalidate (buffer_inv sc1: agent-scope invalidate, which leaves the L2's
// local-memory lines alone) per task.  FINAL data -- blocks of L, written once by D / T / TU and never modified
// again -- is stored write-through (sc1) and may then be cached by every other XCD: no other L2 can
// hold an older copy, because nobody but the owner ever touched those lines before.  Flags are
// agent-scope atomics.
// The ticket order comes from a list-scheduling simulation on the host (mega_build_tasks).
enum { TASK_D = 0, TASK_T = 1, TASK_TI = 2, TASK_U = 3, TASK_UQ = 4, TASK_TU = 5 };

constexpr int MEGA_MAX_Q = 16;         // XCDs
struct MegaArgs {
    double* A; int lda; int n; int nblk;
    const int4* tasks;      // the ticket lists, concatenated
    int nq;                 // number of XCD queues (= XCDs seen by the probe)
    int lstart[MEGA_MAX_Q + 1];   // list q (one per XCD) holds tasks [lstart[q], lstart[q+1]), taken in order
    signed char xcc_queue[16];   // HW_REG_XCC_ID -> queue
    int* sync;              // [0..16) tickets of the lists, [16] abort, then dflag[nblk], tuflag[nblk], tudone[nblk],
                            // tflag[nblk*nrow], ver[nrow*nblk] (nrow = nblk + 4 nwide)
    double* linv; size_t linv_stride;
    double* vbuf; int nwide;   // inverse transposes of the 512 x 512 diagonal blocks 0 .. nwide-1 (row-major, ld 512): see below
    long long spin_limit;   // give up waiting for a dependency after this many ticks of the 100 MHz clock
    int* flag;
    long long* trace;       // optional (STBA_MEGA_TRACE): per task {workgroup, t_ticket, t_ready, t_done}, 100 MHz clock
};
constexpr int MEGA_SYNC_HDR = MEGA_MAX_Q + 1;
constexpr int MEGA_ABORT = MEGA_MAX_Q;
constexpr int MEGA_SMEM_BYTES = 128 * 1024;   // X of the TU task: 8 waves x 8 chunks x 2 KB
constexpr int MEGA_PREDRAW_NB = 2;            // the next ticket is drawn at the start of a task only if the task spans at most
                                              // this many panels (a ticket held for 60-80 us can be a critical one: measured -12 %)

__device__ __forceinline__ int xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return (int)(v & 15u);
}
// which XCD does every workgroup of a launch land on (plan construction)
__global__ void xcc_probe_kernel(int* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

// C(TM x 128) -= P_i P_j^T, K = 128 panel columns, 512 threads; smem: 2*(TM+128)*16 doubles
// C(TM x 128) -= P_i P_j^T, K = 128 panel columns, 512 threads; smem: 2*(TM+128)*16 doubles
// (kchunks 16-column chunks of K: 8 for one panel; a batched trailing update runs several panels in one pass)
template <int TM, int KCH = 16>
__device__ __forceinline__ void syrk_tile512(double* __restrict__ A, int lda, int k0, int row_i, int row_j,
                                             double* smem, int t, int kchunks = NB / 16) {
    constexpr int WR = (TM == 128) ? 2 : 1, WC = 8 / WR;      // wave grid
    constexpr int MB = TM / (16 * WR);                        // MFMA row blocks per wave: 4 | 2
    constexpr int NBK = 8 / WC;                               // MFMA col blocks per wave: 2 | 1
    constexpr int RBA = TM / 16;                              // 16-row blocks of the A tile
    constexpr int SA = TM * KCH, SB = 128 * KCH, NH = KCH / 8;
    double* sA = smem;                   // [2][TM * KCH]
    double* sB = smem + 2 * SA;          // [2][128 * KCH]
    const int lane = t & 63, w = t >> 6;
    const int wr = w / WC, wc = w % WC;
    double4v acc[MB][NBK];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NBK; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row_i + wr * (MB * 16) + m * 16 + (lane >> 4) + 4 * r;
                const int col = row_j + wc * (NBK * 16) + n * 16 + (lane & 15);
                acc[m][n][r] = A[(size_t)row * lda + col];
            }
    // staging map: wave w, half h -> row = (lane&15) + 16*w, k = 2*((lane>>4) + 4h)
    double2 ga[NH], gb[NH];
    const int lrow = lane & 15, lkp = lane >> 4;
    // (TM = 128: all eight waves stage rows of the A tile.  Said at compile time: with the run-time test `w < RBA` the compiler put
    // an exec-mask branch around every A load and store of the K loop -- three s_cbranch per chunk between the MFMAs; without them
    // the factorisation is 1.4 % faster at n = 6000, round 4)
    const bool stage_a = (RBA == 8) ? true : (w < RBA);
    auto gload = [&](int kc) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int row = lrow + 16 * w;
            const int k = 2 * (lkp + 4 * h);
            if (stage_a) ga[h] = *reinterpret_cast<const double2*>(&A[(size_t)(row_i + row) * lda + k0 + kc * KCH + k]);
            gb[h] = *reinterpret_cast<const double2*>(&A[(size_t)(row_j + row) * lda + k0 + kc * KCH + k]);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int k = 2 * (lkp + 4 * h);
            if (stage_a) {
                const int posa = (((k >> 2) * RBA + w) << 6) + ((k & 3) << 4) + lrow;
                // (the A operand is negated HERE, once per element, not at every fragment read: four v_xor per k-step less between
                // the MFMAs -- alone it measured +0.7 %, together with the branch-free staging -2.2 % at n = 6000, -3.8 % at 24 000)
                sA[buf * SA + posa] = -ga[h].x;
                sA[buf * SA + posa + 16] = -ga[h].y;     // k+1: (k&3) is even so +1 -> +16
            }
            const int posb = (((k >> 2) * 8 + w) << 6) + ((k & 3) << 4) + lrow;
            sB[buf * SB + posb] = gb[h].x;
            sB[buf * SB + posb + 16] = gb[h].y;
        }
    };
    gload(0);
    lstore(0);
    __syncthreads();
    const int KC = kchunks * 16 / KCH;
    auto chunk = [&](int kc, int buf) {
        if (kc + 1 < KC) gload(kc + 1);
#pragma unroll
        for (int kq = 0; kq < KCH / 4; ++kq) {
            double a[MB], b[NBK];
#pragma unroll
            for (int m = 0; m < MB; ++m) a[m] = sA[buf * SA + (((kq * RBA + wr * MB + m) << 6) + lane)];
#pragma unroll
            for (int n = 0; n < NBK; ++n) b[n] = sB[buf * SB + (((kq * 8 + wc * NBK + n) << 6) + lane)];
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NBK; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        if (kc + 1 < KC) lstore(buf ^ 1);
        __syncthreads();
    };
    // (measured and not kept, round 4: the loop unrolled by two so that the LDS offsets become immediates: +4 % at n = 6000, +8 % at
    // 24 000 -- the compiler then hoists the operand reads of both chunks and the staging stores drift; KCH = 32, half the barriers
    // at 128 KB of LDS: +1.5 % / +1.2 %; s_setprio around the MFMAs: +1.8 %)
    for (int kc = 0; kc < KC; ++kc) chunk(kc, kc & 1);
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NBK; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row_i + wr * (MB * 16) + m * 16 + (lane >> 4) + 4 * r;
                const int col = row_j + wc * (NBK * 16) + n * 16 + (lane & 15);
                gst<false>(&A[(size_t)row * lda + col], acc[m][n][r]);
            }
}

// The same with the three operands anywhere (tiles of the wide inverse blocks live outside A; the hot task above keeps
// the single base + offsets form: the general one costs 1.6 us more per tile in address arithmetic).
// C (TM x 128) -= Pi Pj^T with K = 128: C at Cb (leading dimension ldc), Pi rows at Pi (ldi), Pj rows at Pj (ldj).
// c_zero: C counts as zero on entry (the first update of a tile of an inverse block, see the wide inverse blocks).
template <int TM>
__device__ __forceinline__ void syrk_tile512_gen(double* __restrict__ Cb, int ldc, const double* __restrict__ Pi, size_t ldi,
                                             const double* __restrict__ Pj, size_t ldj, bool c_zero, double* smem, int t, int kchunks = NB / 16) {
    constexpr int WR = (TM == 128) ? 2 : 1, WC = 8 / WR;      // wave grid
    constexpr int MB = TM / (16 * WR);                        // MFMA row blocks per wave: 4 | 2
    constexpr int NBK = 8 / WC;                               // MFMA col blocks per wave: 2 | 1
    constexpr int RBA = TM / 16;                              // 16-row blocks of the A tile
    double* sA = smem;                   // [2][TM * 16]
    double* sB = smem + 2 * TM * 16;     // [2][2048]
    const int lane = t & 63, w = t >> 6;
    const int wr = w / WC, wc = w % WC;
    double4v acc[MB][NBK];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NBK; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wr * (MB * 16) + m * 16 + (lane >> 4) + 4 * r;
                const int col = wc * (NBK * 16) + n * 16 + (lane & 15);
                acc[m][n][r] = c_zero ? 0.0 : Cb[(size_t)row * ldc + col];
            }
    // staging map: wave w, half h -> row = (lane&15) + 16*w, k = 2*((lane>>4) + 4h)
    double2 ga[2], gb[2];
    const int lrow = lane & 15, lkp = lane >> 4;
    const bool stage_a = (RBA == 8) ? true : (w < RBA);   // see syrk_tile512
    auto gload = [&](int kc) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = lrow + 16 * w;
            const int k = 2 * (lkp + 4 * h);
            if (stage_a) ga[h] = *reinterpret_cast<const double2*>(&Pi[(size_t)row * ldi + (size_t)kc * 16 + k]);
            gb[h] = *reinterpret_cast<const double2*>(&Pj[(size_t)row * ldj + (size_t)kc * 16 + k]);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = 2 * (lkp + 4 * h);
            if (stage_a) {
                const int posa = (((k >> 2) * RBA + w) << 6) + ((k & 3) << 4) + lrow;
                sA[buf * TM * 16 + posa] = -ga[h].x;     // negated at staging, see syrk_tile512
                sA[buf * TM * 16 + posa + 16] = -ga[h].y;     // k+1: (k&3) is even so +1 -> +16
            }
            const int posb = (((k >> 2) * 8 + w) << 6) + ((k & 3) << 4) + lrow;
            sB[buf * 2048 + posb] = gb[h].x;
            sB[buf * 2048 + posb + 16] = gb[h].y;
        }
    };
    gload(0);
    lstore(0);
    __syncthreads();
    const int KC = kchunks;
    for (int kc = 0; kc < KC; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < KC) gload(kc + 1);
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            double a[MB], b[NBK];
#pragma unroll
            for (int m = 0; m < MB; ++m) a[m] = sA[buf * TM * 16 + (((kq * RBA + wr * MB + m) << 6) + lane)];
#pragma unroll
            for (int n = 0; n < NBK; ++n) b[n] = sB[buf * 2048 + (((kq * 8 + wc * NBK + n) << 6) + lane)];
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NBK; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        if (kc + 1 < KC) lstore(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NBK; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wr * (MB * 16) + m * 16 + (lane >> 4) + 4 * r;
                const int col = wc * (NBK * 16) + n * 16 + (lane & 15);
                gst<false>(&Cb[(size_t)row * ldc + col], acc[m][n][r]);
            }
}

// Panel solve of 128 rows (8 waves x 16 rows) inside the persistent kernel.  Same algorithm as
// trsm_group, but the 28 sub-diagonal 16x16 tiles of L11 are staged ONCE per workgroup into LDS, already
// in A-operand order (56 KB; every lane then reads its 32 B with two ds_read_b128), because coherent
// (sc1) global loads have ~2 us latency and a per-wave load chain made the task 50 us long.
// thread 0 only.  Polls with relaxed agent-scope loads; gives up (and raises the abort flag, so that
// every other workgroup gives up too) after `limit` ticks of the 100 MHz clock instead of hanging the device
// (MegaArgs::spin_limit: a multiple of the predicted makespan; the host then reruns the factorisation through the
// stage kernels, see chol_run).
__device__ __forceinline__ bool mega_wait(const int* p, int target, int* abortf, long long limit) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
    const long long t0 = wall_clock64();
    for (;;) {
        __builtin_amdgcn_s_sleep(MEGA_WAIT_SLEEP);
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        if (__hip_atomic_load(abortf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        if (wall_clock64() - t0 > limit) {
            __hip_atomic_store(abortf, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
}

// X = W * L11^-T for the 16 rows of this wave (see trsm_group), all 8 waves of the workgroup together:
// L11's 28 sub-diagonal tiles and the 8 inverse diagonal tiles are staged once into LDS (72 KB) in
// A-operand order.  W holds X on return.  Starts and ends without a barrier on smem.
// `dflag`: the diagonal block's flag is awaited HERE, after the wave's own rows have been requested (they
// do not depend on it), so their latency hides behind the wait.  Returns false on a wait time-out.
__device__ __forceinline__ bool trsm_compute512(double4v (&W)[8], const double* __restrict__ rowp, bool ident,
                                                int ident_row0, int nv, const double* __restrict__ Lb, int lda,
                                                const double* __restrict__ dinv, double* smem, int t, long long* ph,
                                                const int* dflag, int* abortf, int* s_ok, long long spin_limit, bool flag_known = false,
                                                bool active = true) {
    const int lane = t & 63;
    const int n = lane & 15, g = lane >> 4;
    // this wave's rows (issued first: they are not needed before the staging is done).  (active == false: a wave of a
    // HALF panel-solve task that has no rows -- it helps to stage the factor and sits the arithmetic out, so that the four
    // waves with rows have a SIMD each)
#pragma unroll
    for (int J = 0; J < 8; ++J) {
        if (!active) {
#pragma unroll
            for (int r = 0; r < 4; ++r) W[J][r] = 0.0;
        } else if (ident) {
#pragma unroll
            for (int r = 0; r < 4; ++r) W[J][r] = (16 * J + 4 * g + r == ident_row0 + n) ? 1.0 : 0.0;
        } else {
            W[J] = gld4(rowp + 16 * J + 4 * g);
        }
    }
    // (flag_known: the caller has SEEN the diagonal block's flag set -- in the same poll that found the task ready --, so the
    // factor's tiles are requested right behind the rows: one memory round trip in front of the first MFMA instead of two)
    if (!flag_known) {
        if (t == 0) {
            *s_ok = mega_wait(dflag, 1, abortf, spin_limit) ? 1 : 0;
            asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (!*s_ok) return false;
    }
    // stage tile (J, I), J > I, at index 7I - I(I-1)/2 + (J-I-1), then the inverse tiles at 28 + J;
    // element (r, c) of a tile goes to lane (n = pm(r), g = c >> 2), register c & 3, with
    // pm(r) = 4 (r & 3) + (r >> 2) (an involution)
    {
        double v[18];
#pragma unroll
        for (int s2 = 0; s2 < 18; ++s2) {
            const int e = t + 512 * s2;
            const int tile = e >> 8, r = (e >> 4) & 15, c = e & 15;
            if (s2 < 14) {
                const int I = (tile >= 27) ? 6 : (tile >= 25) ? 5 : (tile >= 22) ? 4 : (tile >= 18) ? 3 : (tile >= 13) ? 2 : (tile >= 7) ? 1 : 0;
                const int J = tile - (7 * I - I * (I - 1) / 2) + I + 1;
                const int row = 16 * J + r;
                v[s2] = -Lb[(size_t)row * lda + 16 * I + c];
                if (row >= nv) v[s2] = 0.0;
            } else {
                const int J = tile - 28;
                v[s2] = dinv[(J * 16 + r) * 16 + c];
                if (16 * J + r >= nv) v[s2] = (c == r) ? 1.0 : 0.0;
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 18; ++s2) {
            const int e = t + 512 * s2;
            const int tile = e >> 8, r = (e >> 4) & 15, c = e & 15;
            const int nn = 4 * (r & 3) + (r >> 2);
            smem[tile * 256 + (((c >> 2) << 4) | nn) * 4 + (c & 3)] = v[s2];
        }
    }
    __syncthreads();
    PHASE_STAMP(0);
    if (active) {
#pragma unroll
    for (int I = 0; I < 8; ++I) {
        const double4v iv = *reinterpret_cast<const double4v*>(smem + (28 + I) * 256 + lane * 4);
        double4v y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q) y = __builtin_amdgcn_mfma_f64_16x16x4f64(iv[q], W[I][q], y, 0, 0, 0);
        W[I] = y;
#pragma unroll
        for (int J = I + 1; J < 8; ++J) {
            const int tile = 7 * I - I * (I - 1) / 2 + (J - I - 1);
            const double4v l = *reinterpret_cast<const double4v*>(smem + tile * 256 + lane * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) W[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(l[q], y[q], W[J], 0, 0, 0);
        }
    }
    }
    PHASE_STAMP(1);
    return true;
}

// T / TI task: panel solve of 128 rows, result written through to memory (final data)
__device__ __forceinline__ bool trsm_task512(double* __restrict__ rowp, bool ident, int ident_row0, int nv,
                                             const double* __restrict__ Lb, int lda,
                                             const double* __restrict__ dinv, double* smem, int t, long long* ph,
                                             const int* dflag, int* abortf, int* s_ok, long long spin_limit, bool flag_known = false,
                                             bool active = true) {
    double4v W[8];
    if (!trsm_compute512(W, rowp, ident, ident_row0, nv, Lb, lda, dinv, smem, t, ph, dflag, abortf, s_ok, spin_limit, flag_known, active)) return false;
    const int g = (t & 63) >> 4;
    if (active) {
#pragma unroll
        for (int J = 0; J < 8; ++J) gst4<true>(rowp + 16 * J + 4 * g, W[J]);
    }
    PHASE_STAMP(2);
    return true;
}

// TU task, FOUR-workgroup form (the panels in front of MEGA's TU10_FROM, where workgroups are scarce and ten siblings
// would wait for each other): the critical hand-off D(b) -> D(b+1) in ONE task instead of a panel solve, a flag, and
// a trailing update.  Each of the four workgroups q = 0..3 solves the WHOLE block row b+1 of the panel
// (X = A[b+1, b] L_bb^-T, 128 x 128, redundantly), exchanges X between its waves through LDS (a wave's
// accumulators are already MFMA operand fragments: 128 KB), and applies rows 32q..32q+31 of the update
// A[b+1, b+1] -= X X^T (lower-triangle tiles only).  Workgroup q also writes rows 32q..32q+31 of X.
__device__ __forceinline__ bool tu4_task512(double* __restrict__ A, int lda, int k0, int rb, int q,
                                           const double* __restrict__ dinv, double* smem, int t, long long* ph,
                                           int* loaded, int* ver_diag, const int* dflag, int* abortf, int* s_ok, long long spin_limit,
                                           bool flag_known = false) {
    const int lane = t & 63, w = t >> 6;
    const int n = lane & 15, g = lane >> 4;
    double* rowp = A + (size_t)(rb * NB + 16 * w + n) * lda + k0;
    // output tiles (m, n') of the diagonal tile, m in {2q, 2q+1}, n' <= m: 4q + 3 of them, at most two
    // per wave (e = w and w + 8).  Their current values are requested first: they do not depend on D(b).
    const int ntile = 4 * q + 3;
    double* Cb = A + (size_t)(rb * NB) * lda + rb * NB;
    double4v acc[2];
    int tm[2], tn[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int e = w + 8 * s2;
        tm[s2] = (e < 2 * q + 1) ? 2 * q : 2 * q + 1;
        tn[s2] = (e < 2 * q + 1) ? e : e - (2 * q + 1);
        if (e < ntile) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[s2][r] = Cb[(size_t)(16 * tm[s2] + g + 4 * r) * lda + 16 * tn[s2] + n];
        }
    }
    double4v W[8];
    if (!trsm_compute512(W, rowp, false, 0, NB, A + (size_t)k0 * lda + k0, lda, dinv, smem, t, ph, dflag, abortf, s_ok, spin_limit, flag_known)) return false;
    __syncthreads();                          // every wave is done with the L11 tiles in smem and has consumed its rows
    // the four TU workgroups of this step all READ the whole block row and each WRITES 32 rows of it in
    // place: count the readers, and store only once all four have their copy (see below)
    if (t == 0) __hip_atomic_fetch_add(loaded, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int J = 0; J < 8; ++J) *reinterpret_cast<double4v*>(smem + (w * 8 + J) * 256 + lane * 4) = W[J];
    __syncthreads();
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        if (w + 8 * s2 < ntile) {
            const int m = tm[s2], np = tn[s2];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const double4v fa = *reinterpret_cast<const double4v*>(smem + (m * 8 + c) * 256 + lane * 4);
                const double4v fb = *reinterpret_cast<const double4v*>(smem + (np * 8 + c) * 256 + lane * 4);
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) acc[s2] = __builtin_amdgcn_mfma_f64_16x16x4f64(-fa[qq], fb[qq], acc[s2], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) gst<false>(&Cb[(size_t)(16 * m + g + 4 * r) * lda + 16 * np + n], acc[s2][r]);
        }
    }
    // D(b+1) needs nothing but this tile (it runs on this XCD and finds it in the L2): signal it now, the
    // panel rows below are off the critical chain
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        __hip_atomic_fetch_add(ver_diag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // The three other workgroups hold later tickets of the same queue; they are taken as soon as any
        // workgroup of this XCD is free (needs >= 4 resident workgroups per XCD, checked on the host).
        *s_ok = mega_wait(loaded, 4, abortf, spin_limit) ? 1 : 0;
    }
    __syncthreads();
    if (!*s_ok) return false;
    if ((w >> 1) == q) {
#pragma unroll
        for (int J = 0; J < 8; ++J) gst4<true>(rowp + 16 * J + 4 * g, W[J]);
    }
    PHASE_STAMP(2);
    return true;
}

// TU task (b, q): the critical hand-off D(b) -> D(b+1) in ONE kind of task instead of a panel solve, a flag, and a
// trailing update: block row b+1 of the panel solve AND the update of the next diagonal tile A[b+1, b+1] -= X X^T, fused.
// The 8 x 8 grid of 16 x 16 tiles of the diagonal tile is cut into 2 x 2-tile blocks (I, J), I >= J, q = I (I + 1) / 2 + J:
// MEGA_NTU = 10 workgroups.  Block (I, J) needs the tile rows {2I, 2I+1} and {2J, 2J+1} of X only, so it solves 64 rows
// (32 on the diagonal) of X = A[b+1, b] L_bb^-T -- one wave per tile row, a SIMD each --, exchanges them between its waves
// through LDS (a wave's accumulators are already MFMA operand fragments) and updates its four (three) tiles.  The diagonal
// blocks (I, I) also write their 32 rows of X.  (Until round 3 FOUR workgroups each solved the WHOLE block row redundantly
// -- 200 KB to wait for instead of 136 / 104, two waves per SIMD -- and updated 32 rows of the tile: 17 us from the
// diagonal block's flag to the next one's, now ~12.)
constexpr int MEGA_NTU = 10;
__device__ __forceinline__ bool tu_task512(double* __restrict__ A, int lda, int k0, int rb, int q,
                                           const double* __restrict__ dinv, double* smem, int t, long long* ph,
                                           int* loaded, int* done_cnt, int* ver_diag, const int* dflag, int* abortf, int* s_ok, long long spin_limit,
                                           bool flag_known = false) {
    const int lane = t & 63, w = t >> 6;
    const int n = lane & 15, g = lane >> 4;
    int I = 0;
    while ((I + 1) * (I + 2) / 2 <= q) ++I;
    const int J = q - I * (I + 1) / 2;
    const bool diag = (I == J);
    const int nrow = diag ? 2 : 4;                 // tile rows of X this block solves: local 0, 1 = 2I, 2I+1; local 2, 3 = 2J, 2J+1
    const bool active = w < nrow;
    const int trow = (w < 2) ? 2 * I + (w & 1) : 2 * J + (w & 1);
    double* rowp = A + (size_t)(rb * NB + 16 * trow + n) * lda + k0;
    // output tiles, one per wave e = w: diagonal block (2I,2I) (2I+1,2I) (2I+1,2I+1); off-diagonal (2I,2J) (2I,2J+1) (2I+1,2J) (2I+1,2J+1).
    // lm / ln: the LOCAL tile rows of X the tile multiplies.  Their current values are requested first: they do not depend on D(b).
    const int ntile = diag ? 3 : 4;
    const int lm = diag ? (w >= 1 ? 1 : 0) : (w >> 1), ln = diag ? (w == 2 ? 1 : 0) : 2 + (w & 1);
    const int tm = 2 * I + lm, tn = diag ? 2 * I + ln : 2 * J + (w & 1);
    double* Cb = A + (size_t)(rb * NB) * lda + rb * NB;
    double4v acc = {0.0, 0.0, 0.0, 0.0};
    if (w < ntile) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = Cb[(size_t)(16 * tm + g + 4 * r) * lda + 16 * tn + n];
    }
    double4v W[8];
    if (!trsm_compute512(W, rowp, false, 0, NB, A + (size_t)k0 * lda + k0, lda, dinv, smem, t, ph, dflag, abortf, s_ok, spin_limit, flag_known, active)) return false;
    __syncthreads();                          // every wave is done with the L11 tiles in smem and has consumed its rows
    // the workgroups of this step all READ rows of the block row and the diagonal blocks WRITE 32 rows of it each in
    // place: count the readers, and store only once all of them have their copy (see below)
    if (t == 0) __hip_atomic_fetch_add(loaded, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (active) {
#pragma unroll
        for (int J2 = 0; J2 < 8; ++J2) *reinterpret_cast<double4v*>(smem + (w * 8 + J2) * 256 + lane * 4) = W[J2];
    }
    __syncthreads();
    if (w < ntile) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const double4v fa = *reinterpret_cast<const double4v*>(smem + (lm * 8 + c) * 256 + lane * 4);
            const double4v fb = *reinterpret_cast<const double4v*>(smem + (ln * 8 + c) * 256 + lane * 4);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-fa[qq], fb[qq], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) gst<false>(&Cb[(size_t)(16 * tm + g + 4 * r) * lda + 16 * tn + n], acc[r]);
    }
    // D(b+1) needs nothing but this tile (it runs on this XCD and finds it in the L2): the LAST block to get here signals
    // it now, the panel rows below are off the critical chain
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        if (__hip_atomic_fetch_add(done_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == MEGA_NTU - 1)
            __hip_atomic_fetch_add(ver_diag, 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // Only the diagonal blocks write (their 32 rows of X, over rows the others read): they alone wait for every
        // sibling to have its copy -- the siblings hold tickets of the same queue, the off-diagonal ones EARLIER ones (see
        // mega_build_tasks), and are taken as soon as any workgroup of this XCD is free (needs >= MEGA_NTU resident
        // workgroups per XCD, checked on the host).
        *s_ok = (!diag || mega_wait(loaded, MEGA_NTU, abortf, spin_limit)) ? 1 : 0;
    }
    __syncthreads();
    if (!*s_ok) return false;
    if (diag && active) {
#pragma unroll
        for (int J2 = 0; J2 < 8; ++J2) gst4<true>(rowp + 16 * J2 + 4 * g, W[J2]);
    }
    PHASE_STAMP(2);
    return true;
}

// C(32 x 128) -= P_i P_j^T for the next panel's tile column: operands straight from global memory
// into MFMA operand registers (every load of the task is in flight at once: one memory latency
// instead of eight), wave w -> output columns 16w..16w+15, two 16x16 tiles.
__device__ __forceinline__ void syrk_q32(double* __restrict__ A, int lda, int k0, int row_i, int row_j, int t, long long* ph) {
    const int lane = t & 63, w = t >> 6;
    const int n = lane & 15, g = lane >> 4;
    const double* pa0 = A + (size_t)(row_i + n) * lda + k0 + 4 * g;
    const double* pa1 = pa0 + (size_t)16 * lda;
    const double* pb = A + (size_t)(row_j + 16 * w + n) * lda + k0 + 4 * g;
    double4v a0[8], a1[8], bb[8], acc[2];
#pragma unroll
    for (int c = 0; c < 8; ++c) { bb[c] = gld4(pb + 16 * c); a0[c] = gld4(pa0 + 16 * c); a1[c] = gld4(pa1 + 16 * c); }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][r] = A[(size_t)(row_i + 16 * m + g + 4 * r) * lda + row_j + 16 * w + n];
    PHASE_STAMP(0);
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0[c][q], bb[c][q], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1[c][q], bb[c][q], acc[1], 0, 0, 0);
        }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) gst<false>(&A[(size_t)(row_i + 16 * m + g + 4 * r) * lda + row_j + 16 * w + n], acc[m][r]);
    PHASE_STAMP(1);
}

// tickets, flags and tile versions of a factorisation, and the pivot flag, back to zero (one launch in front of the
// persistent kernel instead of two memsets)
__global__ __launch_bounds__(256) void chol_reset_kernel(int* __restrict__ sync, int n, int* __restrict__ flag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) sync[i] = 0;
    if (i == 0) flag[0] = 0;
}

// ---- the flag arrays behind the header of MegaArgs::sync, and the readiness test of a task ----
struct MegaView {
    int nblk, nrow;
    int* abortf; int* dflag; int* tuflag; int* tudone; int* tflag; int* ver;
};
__device__ __forceinline__ MegaView mega_view(const MegaArgs& a) {
    MegaView v;
    v.nblk = a.nblk;
    // Block rows >= nblk are VIRTUAL: row nblk + p is the identity block under panel p of a wide (4-panel) diagonal
    // block; carried through the panel solves and updates of the panels p .. 4q+3 of its wide block q = p / 4 it
    // becomes block row p - 4q of the inverse transpose of that 512 x 512 diagonal block, which the backward
    // substitution multiplies by (12 dependent steps instead of 47).  Its tiles live in a.vbuf, not in A.
    v.nrow = a.nblk + 4 * a.nwide;
    v.abortf = a.sync + MEGA_ABORT;
    v.dflag = a.sync + MEGA_SYNC_HDR;
    v.tuflag = v.dflag + a.nblk;
    v.tudone = v.tuflag + a.nblk;
    v.tflag = v.tudone + a.nblk;                // [panel b][row i], i < nrow
    v.ver = v.tflag + a.nblk * v.nrow;          // [row i][panel j]
    return v;
}
__device__ __forceinline__ int mega_ldf(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// (tu_dset: for a TU task, whether its diagonal block was there already -- then the task fetches the factor at once)
__device__ __forceinline__ bool mega_ready(const MegaView& v, const int4 d, bool& tu_dset) {
    const int type = d.x & 0xff, b = d.y, ti = d.z, tj = d.w;
    const int nblk = v.nblk, nrow = v.nrow;
    auto first_panel = [&](int i) { return i < nblk ? 0 : i - nblk; };      // the first panel that updates row i
    if (type == TASK_D) return mega_ldf(&v.ver[b * nblk + b]) >= 4 * b;
    if (type == TASK_T) {       // (a panel solve is only started once its diagonal block is there; starting it earlier, with its
                                // rows requested before the wait, measured the same: 2.465 against 2.463 ms)
        const int ve = mega_ldf(&v.ver[ti * nblk + b]), df = mega_ldf(&v.dflag[b]);
        return (ve >= 4 * (b - first_panel(ti))) & (df >= 1);
    }
    if (type == TASK_TI) return mega_ldf(&v.dflag[b]) >= 1;
    if (type == TASK_TU) {
        const int v0 = mega_ldf(&v.ver[(b + 1) * nblk + b]), v1 = mega_ldf(&v.ver[(b + 1) * nblk + b + 1]), df = mega_ldf(&v.dflag[b]);
        tu_dset = df >= 1;
        return (v0 >= 4 * b) & (v1 >= 4 * b);
    }
    const int i = (type == TASK_UQ) ? (ti >> 2) : ti;
    const int nbp = max(1, (d.x >> 16) & 0xff);            // panels in this task (a batched trailing update: b is its last one)
    const int f0 = mega_ldf(&v.tflag[b * nrow + i]), f1 = mega_ldf(&v.tflag[b * nrow + tj]), ve = mega_ldf(&v.ver[i * nblk + tj]);
    return (f0 >= 4) & (f1 >= 4) & (ve >= 4 * (b - nbp + 1 - first_panel(i)));
}

// Wave 0 of a workgroup: the next task of ticket list [lbeg, lend).  One in-order list per XCD: a workgroup
// draws the next ticket while its current task starts (the atomic's latency hides behind the task), reads the
// descriptor afterwards and PARKS on it, polling the task's input flags; it starts the instant the last one flips.
// (TU tasks go on waiting for the diagonal block inside, with their operands loaded.)  Returns the task index, -2 when
// the list is exhausted, -1 on a time-out / abort (ok = false).
struct MegaTicket {
    int mine = -1;          // the ticket held (-1: none), wave-uniform
    int4 md;
    bool done = false;      // the list is exhausted
    int m_raw = 0;          // lane 0: the ticket drawn when the previous task started
    bool m_pending = false;
};
__device__ __forceinline__ int mega_next_task(const MegaArgs& a, const MegaView& v, MegaTicket& tk, int* ticket, int lbeg, int lend, int lane,
                                              bool& ok, bool& tu_dset, long long& t_poll) {
    t_poll = a.trace ? wall_clock64() : 0;
    ok = true;
    if (tk.mine < 0 && !tk.done) {
        if (!tk.m_pending && lane == 0) tk.m_raw = lbeg + atomicAdd(ticket, 1);
        tk.m_pending = false;
        tk.mine = __builtin_amdgcn_readfirstlane(tk.m_raw);
        if (tk.mine >= lend) { tk.mine = -1; tk.done = true; }
        else tk.md = a.tasks[tk.mine];
    }
    int pick = -2;                          // nothing left for this workgroup
    if (tk.mine >= 0) {
        const long long t0 = wall_clock64();
        for (;;) {
            if (mega_ready(v, tk.md, tu_dset)) { pick = tk.mine; tk.mine = -1; break; }
            if (mega_ldf(v.abortf) != 0) { ok = false; pick = -1; break; }
            if (wall_clock64() - t0 > a.spin_limit) {            // 100 MHz counter
                __hip_atomic_store(v.abortf, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false; pick = -1;
                break;
            }
            __builtin_amdgcn_s_sleep(MEGA_POLL_SLEEP);
        }
    }
    // this CU's L1 may hold lines of tiles that other CUs have rewritten since
    // (buffer_inv sc0 does NOT do it outside threadgroup-split mode: measured, stale L1 hits)
    asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
    return pick;
}

__global__ __launch_bounds__(512) void chol_mega_kernel(MegaArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];   // MEGA_SMEM_BYTES: SYRK staging | diagonal block | L11 tiles | X
    __shared__ int s_task, s_ok, s_dset;
    const int t = threadIdx.x;
    const int nblk = a.nblk;
    const MegaView v = mega_view(a);
    int* abortf = v.abortf;
    int* dflag = v.dflag;
    int* tuflag = v.tuflag;
    const int nrow = v.nrow;
    int* tflag = v.tflag;
    int* ver = v.ver;
    // row i, column panel j: address of the tile's first element and its leading dimension
    auto tile_ptr = [&](int i, int j, int& ld) -> double* {
        if (i < nblk) { ld = a.lda; return a.A + (size_t)i * NB * a.lda + (size_t)j * NB; }
        const int p = i - nblk, qw = p >> 2;
        ld = 4 * NB;
        return a.vbuf + (size_t)qw * (4 * NB) * (4 * NB) + (size_t)(p & 3) * NB * (4 * NB) + (size_t)(j - 4 * qw) * NB;
    };
    auto first_panel = [&](int i) { return i < nblk ? 0 : i - nblk; };      // the first panel that updates row i
    const int q = a.xcc_queue[xcc_id()];
    if (q < 0) return;                        // an XCD the probe did not see: no list, nothing to do
    int* ticket = a.sync + q;
    const int lbeg = a.lstart[q], lend = a.lstart[q + 1];
    MegaTicket tk;
    for (;;) {
        if (t < 64) {
            bool ok, tu_dset = false;
            long long t_poll;
            const int pick = mega_next_task(a, v, tk, ticket, lbeg, lend, t, ok, tu_dset, t_poll);
            if (t == 0) {
                if (a.trace && pick >= 0) {
                    a.trace[8 * (size_t)pick] = blockIdx.x; a.trace[8 * (size_t)pick + 1] = t_poll;
                    a.trace[8 * (size_t)pick + 2] = wall_clock64();
                }
                s_dset = tu_dset ? 1 : 0;
                s_task = pick;
                s_ok = ok ? 1 : 0;
            }
            // draw the next ticket now, look at it after the task.  Not before a TU task: it waits for its three
            // siblings, which hold LATER tickets -- this workgroup must not sit on one of them.
            if (!tk.done && pick >= 0 && (tk.md.x & 0xff) != TASK_TU && ((tk.md.x >> 16) & 0xff) <= MEGA_PREDRAW_NB) {
                if (t == 0) tk.m_raw = lbeg + atomicAdd(ticket, 1);
                tk.m_pending = true;
            }
        }
        __syncthreads();
        if (!s_ok) {                            // dependency time-out: make the host see it (it reruns the stage kernels)
            if (t == 0) atomicExch(a.flag, CHOL_FLAG_TIMEOUT);
            break;
        }
        const int task = s_task;
        if (task < 0) break;
        const int4 d = a.tasks[task];
        const int type = d.x & 0xff, b = d.y, ti = d.z, tj = d.w;
        const int k0 = b * NB;
        // a fresh copy of the thread index per task: keeps the compiler from hoisting every task's
        // lane-dependent address arithmetic out of the ticket loop (that cost 100+ spilled registers)
        int tt = t;
        asm volatile("" : "+v"(tt));
        double* li = a.linv + (size_t)b * a.linv_stride;
        long long* ph = a.trace ? a.trace + 8 * (size_t)task + 4 : nullptr;
        if (type == TASK_D) {
            diag_block2<true>(a.A, a.lda, k0, a.n, a.flag, li + NB * NB, smem, tt, ph);
        } else if (type == TASK_T) {
            const int lane = tt & 63, w = tt >> 6;
            int ldr;
            // tj = 0: the whole tile row (eight waves, 16 rows each); tj = 1 / 2: its upper / lower 64 rows on waves 0..3 -- half
            // the bytes to wait for and a SIMD per wave: the panel solve of a row is one link of the row sweeps T -> U -> T
            const bool active = (tj == 0) || (w < 4);
            const int row_in_tile = (tj == 0 ? 16 * w : 64 * (tj - 1) + 16 * (w & 3)) + (lane & 15);
            double* rowp = tile_ptr(ti, b, ldr) + (size_t)row_in_tile * ldr;
            // (a panel solve is only started once its diagonal block is there: see mega_ready)
            if (!trsm_task512(rowp, false, 0, NB, a.A + (size_t)k0 * a.lda + k0, a.lda, li + NB * NB, smem, tt, ph, &dflag[b], abortf, &s_ok,
                              a.spin_limit, true, active)) {
                if (t == 0) atomicExch(a.flag, CHOL_FLAG_TIMEOUT);
                break;
            }
        } else if (type == TASK_TI) {
            const int lane = tt & 63, w = tt >> 6;
            const int nv = min(NB, (a.lda - 1) - k0);
            // inverse transpose of block b: into its wide block's diagonal tile, or (tail blocks) into linv
            int ldr = NB;
            double* base = li;
            if (b < 4 * a.nwide) base = tile_ptr(nblk + b, b, ldr);
            double* rowp = base + (size_t)(16 * w + (lane & 15)) * ldr;
            if (!trsm_task512(rowp, true, 16 * w, nv, a.A + (size_t)k0 * a.lda + k0, a.lda, li + NB * NB, smem, tt, ph, &dflag[b], abortf, &s_ok,
                              a.spin_limit, true)) {
                if (t == 0) atomicExch(a.flag, CHOL_FLAG_TIMEOUT);
                break;
            }
        } else if (type == TASK_U) {
            if (ti < nblk) {
                const int nbp = max(1, (d.x >> 16) & 0xff);
                syrk_tile512<128>(a.A, a.lda, k0 - (nbp - 1) * NB, ti * NB, tj * NB, smem, tt, nbp * (NB / 16));
            } else {
                int ldc, ldi, ldj;
                double* Cb = tile_ptr(ti, tj, ldc);
                const double* Pi = tile_ptr(ti, b, ldi);
                const double* Pj = tile_ptr(tj, b, ldj);
                syrk_tile512_gen<128>(Cb, ldc, Pi, ldi, Pj, ldj, b == first_panel(ti), smem, tt);
            }
        } else if (type == TASK_TU) {
            const bool tu_ok = (tj == 2)
                ? tu4_task512(a.A, a.lda, k0, b + 1, ti, li + NB * NB, smem, tt, ph, &tuflag[b], &ver[(b + 1) * nblk + b + 1], &dflag[b], abortf, &s_ok,
                              a.spin_limit, s_dset != 0)
                : tu_task512(a.A, a.lda, k0, b + 1, ti, li + NB * NB, smem, tt, ph, &tuflag[b], &v.tudone[b], &ver[(b + 1) * nblk + b + 1], &dflag[b], abortf, &s_ok,
                             a.spin_limit, s_dset != 0);
            if (!tu_ok) {
                if (t == 0) atomicExch(a.flag, CHOL_FLAG_TIMEOUT);
                break;
            }
        } else {
            syrk_q32(a.A, a.lda, k0, (ti >> 2) * NB + (ti & 3) * 32, tj * NB, tt, ph);
        }
        // release: every wave waits until its stores are acknowledged (by the L2; by memory for the
        // write-through ones), then the flag is raised with an agent-scope atomic
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            if (type == TASK_D) __hip_atomic_fetch_add(&dflag[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (type == TASK_T) __hip_atomic_fetch_add(&tflag[b * nrow + ti], tj == 0 ? 4 : 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (type == TASK_TI) { if (b < 4 * a.nwide) __hip_atomic_fetch_add(&tflag[b * nrow + nblk + b], 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            else if (type == TASK_TU) { if (tj != 0) __hip_atomic_fetch_add(&tflag[b * nrow + b + 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // (the four diagonal blocks: they wrote X; ver: inside the task)
            else if (type == TASK_U) __hip_atomic_fetch_add(&ver[ti * nblk + tj], 4 * max(1, (d.x >> 16) & 0xff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (type == TASK_UQ) __hip_atomic_fetch_add(&ver[(ti >> 2) * nblk + tj], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.trace) a.trace[8 * (size_t)task + 3] = wall_clock64();
        }
    }
}

// host side: the task list for nblk block columns (static per size), in one global dataflow order,
// then split stably into one queue per XCD by the owner of the tile each task writes
static int mega_task_row(const int4& tk) {
    // every task writes tiles of one tile row only (TU writes (b+1, b) and (b+1, b+1))
    switch (tk.x & 0xff) {
        case TASK_D: case TASK_TI: return tk.y;
        case TASK_TU: return tk.y + 1;
        case TASK_T: case TASK_U: return tk.z;
        default: return tk.z >> 2;
    }
}
// A tile belongs to the XCD of its tile ROW.  Row i carries ~ i^2 / 2 trailing updates, so the rows are
// dealt out heaviest first, each to the XCD with the least work so far (LPT): the totals differ by < 1 %
// (a boustrophedon deal leaves 8 % between the heaviest XCD and the mean when nblk is not a multiple
// of 16, and the heaviest XCD sets the pace of the throughput-bound first half).
static std::vector<int> mega_row_owner(int nblk, int nq, int nvirt = 0) {
    std::vector<int> owner((size_t)(nblk + nvirt), 0);
    for (int v = 0; v < nvirt; ++v) owner[(size_t)(nblk + v)] = v % nq;      // virtual rows (inverse blocks): a few light tasks each
    // (dealt boustrophedon or cyclically instead: measured 8 % worse)
    std::vector<double> load((size_t)nq, 0.0);
    for (int row = nblk - 1; row >= 0; --row) {
        int best = 0;
        for (int q = 1; q < nq; ++q)
            if (load[(size_t)q] < load[(size_t)best]) best = q;
        owner[(size_t)row] = best;
        load[(size_t)best] += 24.0 * row * (row - 1) / 2.0 + 23.0 * row + 110.0;
    }
    return owner;
}
// The ticket order is produced by LIST SCHEDULING a model of the machine on the host: `wg` workers per XCD queue, measured
// task durations, and bottom-level priorities (longest path to the end of the graph, HLFET).  Sorting the tasks by the
// time their model worker became free gives (i) one global topological order, which the deadlock argument needs, and
// (ii) per-queue orders in which a workgroup rarely takes a ticket whose inputs are far from ready (an in-order ticket
// list has no other notion of priority).
struct MegaShardModel {          // what-if: the queues are spread over several GPUs (design study, DESIGN.md "sharded reduced solve")
    int n_gpus = 1;              // queue q belongs to GPU q / (nq / n_gpus)
    int rows_per_group = 0;      // tile rows are dealt to the GPUs in groups of this many consecutive rows (0: one XCD-round, nq / n_gpus)
    double hop_us = 0.0;         // flag latency of a dependency that crosses GPUs
    double tile_us = 0.0;        // + transfer time of the 128 x 128 tile it carries (128 KiB over one xGMI link)
    double cross_edges = 0.0;    // out: dependencies that crossed GPUs
    double tiles_in_max = 0.0;   // out: distinct remote tiles fetched by the busiest GPU
};
// Far trailing updates are applied `batch` panels per pass over a tile (see mega_build_tasks).  The batch trades traffic for
// task length: a tile is read and written once per batch (round 4, PMC passes at n = 24 000 with two-panel batches: 383 GB per
// factorisation = 4 TB/s averaged over the kernel -- HBM-bound, 256 KB of operands + 128 KB of tile per panel and tile, no
// reuse of the operand panels in the L2s), but a batch of B panels is a task of ~6 + 18 B us that nothing can pre-empt.  Small
// systems are chain-bound and want short tasks (n = 6000: 2.294 / 2.295 / 2.344 ms with B = 2 / 3 / 4); large ones are
// throughput-bound (n = 12 000: 13.09 / 12.01 / 11.55 ms with B = 2 / 4 / 8; n = 24 000: 96.9 / 88.6 / 84.7 / 82.6 ms with
// B = 2 / 4 / 8 / 16, 0.61 -> 0.71 of the FP64 matrix-core peak).  Rule: double the batch while a batch round still
// leaves every workgroup a few tasks (nblk^2 / 2 tiles on 256 workgroups).
static int mega_default_batch(int nblk) {
    const double tiles_per_wg = 0.5 * nblk * nblk / 256.0;
    int b = 2;
    while (b < 16 && tiles_per_wg / b >= 2.2) b *= 2;
    return b;
}
struct MegaMachine {
    int nq = 8;                  // XCD queues
    int wg = 32;                 // model workers (= resident workgroups) per queue
};
static void mega_build_tasks(int nblk, const MegaMachine& mach, std::vector<int4>& out, int* lstart, std::vector<float>* sim_start = nullptr,
                             double* makespan_out = nullptr, int nwide = 0, MegaShardModel* shard = nullptr) {
    struct Node { int4 tk; std::vector<int> succ; int indeg = 0; double dur = 0, prio = 0, start = 0, ready = 0, avail = 0; int q = 0; int last_pred = -1; };
    std::vector<Node> nodes;
    const int NBK = nblk;
    const int nq = mach.nq;
    const int nl = nq;                             // one ticket list per XCD queue
    std::vector<int> rowq = mega_row_owner(nblk, nq, 4 * nwide);
    const int n_gpus = shard ? shard->n_gpus : 1, q_per_gpu = nq / std::max(1, n_gpus);
    if (shard && n_gpus > 1) {
        // block-cyclic over the GPUs, cyclic over a GPU's XCDs: rows [g R, (g+1) R) of every round of n_gpus R rows go to GPU g
        const int R = shard->rows_per_group > 0 ? shard->rows_per_group : q_per_gpu;
        for (int row = 0; row < nblk; ++row) {
            const int local = (row / (R * n_gpus)) * R + row % R;        // index among the rows of its GPU
            rowq[(size_t)row] = ((row / R) % n_gpus) * q_per_gpu + local % q_per_gpu;
        }
    }
    auto gpu_of = [&](int l) { return n_gpus > 1 ? l / q_per_gpu : 0; };
    static const int QROWS = knob_int("STBA_MEGA_QROWS", 2);
    // (The chains pace the last ~30 panels whatever the size: the trailing updates of panel nblk - r are r^2 / 2 tiles of ~24 us
    // on 256 workgroups, < 45 us for r < 31.  The three thresholds below count from the END: measured at n = 6000 -- nblk = 47,
    // panels 17 / 32 / 30 -- and at n = 12 000, where fractions of nblk cost 2-3 %.)
    const int THALF_FROM = knob_int("STBA_MEGA_THALF_FROM", std::max(0, nblk - 30));
    static const double THALF_DUR = knob_double("STBA_MEGA_THALF_DUR", 0.6);
    // the fused panel-solve + diagonal-tile update of a step: ten 2 x 2-tile block tasks from panel TU10_FROM on, four
    // quarter tasks (each solving the whole block row) before
    // (n = 6000, medians of interleaved runs on one box, tools/chol_ab.py: never 2.388 ms; from panel 18 / 26 / 28 / 32 / 36 / 40:
    // 2.373 / 2.357 / 2.364 / 2.351 / 2.365 / 2.382 -- while workgroups are scarce ten siblings wait for each other)
    const int TU10_FROM = knob_int("STBA_MEGA_TU10_FROM", std::max(0, nblk - 15));
    static const double TU10_DUR = knob_double("STBA_MEGA_TU10_DUR", 0.65);
    auto ntu = [&](int b) { return b >= TU10_FROM ? MEGA_NTU : 4; };
    // from panel QFROM on (the chain-bound part of the factorisation, where workgroups are idle) every row's tile in the next
    // panel column is updated by four quarter tasks: the row sweeps T -> U -> T get shorter
    const int DQ_FROM = knob_int("STBA_MEGA_DQ_FROM", std::max(0, nblk - 17));      // quarter updates of the diagonal tile after next, see below
    const int QFROM = knob_int("STBA_MEGA_QFROM", std::max(0, nblk - 17));      // (panel 30 of 47; with the short TU tasks behind it: 2.351 -> 2.343 ms)
    std::vector<int> idD((size_t)NBK, -1), idTI((size_t)NBK, -1), idTU((size_t)NBK * MEGA_NTU, -1), idT((size_t)NBK * NBK, -1), idT2((size_t)NBK * NBK, -1),
        idUq((size_t)NBK * NBK * 4, -1), idU((size_t)NBK * NBK * NBK, -1), idUd((size_t)NBK * 4, -1);
    // measured on MI355X (tools/mega_trace.py), microseconds, plus ~2 us of flag latency per hop
    // (debug builds: STBA_MEGA_DUR=d,t,ti,u,uq,tu overrides them for experiments)
    double DUR[6] = {23.0, 23.0, 19.0, 25.0, 16.5, 20.0};      // (D: 21.3 us since round 2)
    double DUR_K = 18.0;           // one more panel (K += 128) inside a batched trailing update (measured: 24 us for one panel, 42 for two)
    static const int BATCH_KNOB = knob_int("STBA_MEGA_BATCH", 0);
    const int BATCH = BATCH_KNOB > 0 ? std::min(64, BATCH_KNOB) : mega_default_batch(nblk);
    static const int FAR = std::max(0, knob_int("STBA_MEGA_FAR", 0));
    static const int BLAG = std::max(0, knob_int("STBA_MEGA_BLAG", 2));      // (round 3, with the shorter chain tasks: 3 -> 2: -0.02 ms; 1: the same; 0: +0.25 ms)
    if (const char* e = knob_str("STBA_MEGA_DUR")) sscanf(e, "%lf,%lf,%lf,%lf,%lf,%lf", &DUR[0], &DUR[1], &DUR[2], &DUR[3], &DUR[4], &DUR[5]);
    auto add = [&](int type, int b, int i, int j, double prio) {
        Node nd; nd.tk = make_int4(type, b, i, j); nd.dur = DUR[type]; nd.prio = prio;
        nd.q = rowq[(size_t)mega_task_row(nd.tk)];
        nodes.push_back(nd);
        return (int)nodes.size() - 1;
    };
    // priorities: smaller = sooner.  The key is the block column a task works towards (10 per column)
    // plus a class offset, so that everything the NEXT panels need goes before trailing updates of far
    // columns, whatever step they belong to (an in-step order would bury the update of tile (b+1, b+1)
    // by panel b-1 behind the whole backlog of panel b-2).  (Replaced by the bottom level below; kept as the tie-break.)
    for (int b = 0; b < NBK; ++b) {
        idD[(size_t)b] = add(TASK_D, b, 0, 0, 10.0 * b);
        idTI[(size_t)b] = add(TASK_TI, b, 0, 0, 10.0 * NBK + b);
        if (b + 1 < NBK)
            for (int q = 0; q < ntu(b); ++q) {        // (tk.w = 1: a diagonal block (I, I), which writes 32 rows of X; 2: the four-workgroup form)
                int I = 0;
                while ((I + 1) * (I + 2) / 2 <= q) ++I;
                const bool dg = (q == I * (I + 1) / 2 + I);
                idTU[(size_t)b * MEGA_NTU + q] = add(TASK_TU, b, q, ntu(b) == 4 ? 2 : dg ? 1 : 0, 10.0 * b + 5 + (ntu(b) != 4 && dg ? 1e-4 : 0.0));
                if (ntu(b) != 4) nodes[(size_t)idTU[(size_t)b * MEGA_NTU + q]].dur = TU10_DUR * DUR[TASK_TU];
            }
        for (int i = b + 2; i < NBK; ++i) {
            if (b >= THALF_FROM) {
                // two HALF tasks (64 rows each on four waves, see the kernel's TASK_T) once the chains pace the factorisation:
                // a panel solve is one of the two links of every row sweep T -> U -> T, and as a half it waits for 136 KB
                // instead of 200 and has a SIMD per wave -- 11 us instead of 18.4.  (Measured at n = 6000, same box: whole
                // tasks 2.509 ms; halves from panel 0 / 10 / 14 / 17 / 20 / 24: 2.493 / 2.445 / 2.425 / 2.425 / 2.431 / 2.427;
                // QUARTER tasks from 17 / 24: 2.517 / 2.472 -- the 72 KB of the diagonal factor every part stages do not shrink.)
                idT[(size_t)b * NBK + i] = add(TASK_T, b, i, 1, 10.0 * b + 6 + 1e-3 * i);
                idT2[(size_t)b * NBK + i] = add(TASK_T, b, i, 2, 10.0 * b + 6 + 1e-3 * i + 5e-4);
                nodes[(size_t)idT[(size_t)b * NBK + i]].dur = nodes[(size_t)idT2[(size_t)b * NBK + i]].dur = THALF_DUR * DUR[TASK_T];
            } else idT[(size_t)b * NBK + i] = add(TASK_T, b, i, 0, 10.0 * b + 6 + 1e-3 * i);
            // the next panel's column: 32-row tasks (short latency) only for the rows the second critical chain
            // needs soon; a quarter task costs 14.5 us of a workgroup against 23.4 us for a whole tile, so the
            // rows further down take the whole-tile task (their panel solve comes a diagonal block later)
            if (i <= b + 1 + QROWS || b >= QFROM) {
                for (int q = 0; q < 4; ++q)
                    idUq[((size_t)b * NBK + i) * 4 + q] = add(TASK_UQ, b, i * 4 + q, b + 1, 10.0 * (b + 1) + 2 + 1e-3 * i);
            } else {
                idUq[((size_t)b * NBK + i) * 4] = add(TASK_U, b, i, b + 1, 10.0 * (b + 1) + 2 + 1e-3 * i);
            }
        }
        // Trailing updates of the tiles beyond the next panel column.  Tile (i, j) needs the panels 0 .. j-2 at some point
        // before its last update (panel j-1, above: the latency-critical one); they are applied BATCH panels at a time in
        // one pass over the tile, K = 128 BATCH: the tile is read and written once per batch instead of once per panel and
        // the task's fixed costs (waiting for the first operands and the tile, the stores, the hand-over: ~10 of 24 us) are
        // paid once per batch.  The task of a batch carries its LAST panel in .y and the panel count in bits 16..23 of .x;
        // the MFMAs run on the same accumulators in the same order as panel-by-panel, so the result is bit-identical.
        for (int j = b + 2; j < NBK; ++j) {
            // panels 0 .. j-2 of column j in batches [0, BATCH), [BATCH, 2 BATCH), ...: this panel closes a batch if it
            // is the last of its group or the last one of the column
            // (the last BLAG panels before the final one stay single tasks: a long batch there would sit on the path to
            // the tile's panel solve)
            const int lim = j - 2 - BLAG;
            // (FAR batches, debug knob STBA_MEGA_FAR = f > 0: a group of 2 BATCH panels whose last one is still more than f panels in front of
            // column j goes in ONE pass -- longer tasks only where the chains are far away)
            int BJ = BATCH;
            if (FAR > 0 && BATCH > 1 && j - ((b / (2 * BATCH)) * 2 * BATCH + 2 * BATCH - 1) > FAR) BJ = 2 * BATCH;
            const int b0 = (BJ <= 1 || b > lim) ? b : (b / BJ) * BJ, last = (BJ <= 1 || b > lim) ? b : std::min(b0 + BJ - 1, lim);
            if (b != last) continue;
            const int nb = b - b0 + 1;
            for (int i = j; i < NBK; ++i) {
                if (i == j && j == b + 2 && b >= DQ_FROM) {
                    // The diagonal tile (b+2, b+2): its update by panel b is the last input of TU(b+1) to arrive in the
                    // chain-bound part -- D(b) -> half panel solve of row b+2 (12 us) -> this update (24 us as a whole-tile
                    // task) = 38 us against a step of 36 (trace: the last TU(b) block became ready 6 us AFTER D(b) every
                    // other panel).  Four quarter tasks (32 rows each, operands straight from memory: 14.6 us) instead.
                    for (int q = 0; q < 4; ++q) {
                        const int id = add(TASK_UQ, b, i * 4 + q, j, 10.0 * j + 1 + 1e-3 * q);
                        idUd[(size_t)b * 4 + q] = id;
                        if (q == 0) idU[((size_t)b * NBK + i) * NBK + j] = id;
                    }
                    continue;
                }
                const int id = add(TASK_U, b, i, j, 10.0 * j + 3 + 1e-3 * i + 1e-6 * b);
                nodes[(size_t)id].tk.x |= nb << 16;
                nodes[(size_t)id].dur += (nb - 1) * DUR_K;
                for (int bb = b0; bb <= b; ++bb) idU[((size_t)bb * NBK + i) * NBK + j] = id;
            }
        }
    }
    auto dep = [&](int from, int to) {     // `to` needs `from`
        if (from < 0 || to < 0) return;
        nodes[(size_t)from].succ.push_back(to);
        nodes[(size_t)to].indeg++;
    };
    for (int b = 0; b < NBK; ++b) {
        const int d = idD[(size_t)b];
        if (b > 0)
            for (int q = 0; q < ntu(b - 1); ++q) dep(idTU[(size_t)(b - 1) * MEGA_NTU + q], d);
        dep(d, idTI[(size_t)b]);
        if (b + 1 < NBK)
            for (int q = 0; q < ntu(b); ++q) {
                const int tu = idTU[(size_t)b * MEGA_NTU + q];
                dep(d, tu);
                if (b > 0) {
                    for (int q2 = 0; q2 < 4; ++q2) dep(idUq[((size_t)(b - 1) * NBK + (b + 1)) * 4 + q2], tu);
                    dep(idU[((size_t)(b - 1) * NBK + (b + 1)) * NBK + (b + 1)], tu);
                    for (int q2 = 1; q2 < 4; ++q2) dep(idUd[(size_t)(b - 1) * 4 + q2], tu);     // (its quarters, if it was split)
                }
            }
        for (int i = b + 2; i < NBK; ++i) {
            const int t = idT[(size_t)b * NBK + i], t2 = idT2[(size_t)b * NBK + i];
            dep(d, t); dep(d, t2);
            if (b > 0)
                for (int q2 = 0; q2 < 4; ++q2) { const int uu = idUq[((size_t)(b - 1) * NBK + i) * 4 + q2]; dep(uu, t); dep(uu, t2); }
            for (int q = 0; q < 4; ++q) {
                const int uq = idUq[((size_t)b * NBK + i) * 4 + q];
                if (uq < 0) continue;            // (rows with ONE whole-tile task use slot 0 only)
                dep(t, uq); dep(t2, uq);
                for (int q2 = 0; q2 < ntu(b); ++q2) dep(idTU[(size_t)b * MEGA_NTU + q2], uq);
                if (b > 0) dep(idU[((size_t)(b - 1) * NBK + i) * NBK + (b + 1)], uq);
            }
        }
        for (int j = b + 2; j < NBK; ++j)
            for (int i = j; i < NBK; ++i) {
                const int u0 = idU[((size_t)b * NBK + i) * NBK + j];
                if (nodes[(size_t)u0].tk.y != b) continue;          // (a batch: its dependencies hang on its last panel)
                const bool split = (nodes[(size_t)u0].tk.x & 0xff) == TASK_UQ;       // (the diagonal tile after next, in quarters)
                for (int q2 = 0; q2 < (split ? 4 : 1); ++q2) {
                    const int u = split ? idUd[(size_t)b * 4 + q2] : u0;
                    for (auto* v : {&idT, &idT2}) { dep((*v)[(size_t)b * NBK + i], u); if (j != i) dep((*v)[(size_t)b * NBK + j], u); }
                    const int nbp = split ? 1 : std::max(1, (nodes[(size_t)u].tk.x >> 16) & 0xff);
                    if (b - nbp >= 0) dep(idU[((size_t)(b - nbp) * NBK + i) * NBK + j], u);
                }
            }
    }
    // wide inverse blocks: the identity block under panel p = 4q + v of wide block q (virtual row nblk + p) is carried
    // through panels p .. 4q+3: TI(p) makes its diagonal tile, then per later panel bj of the block the updates
    // U(k; row, bj), k = p .. bj-1, and the panel solve T(bj; row)
    for (int qw = 0; qw < nwide; ++qw)
        for (int v = 0; v < 4; ++v) {
            const int pnl = 4 * qw + v, iv = NBK + pnl;
            std::vector<int> xdone(4, -1);            // task that finishes tile (iv, 4qw + j)
            xdone[(size_t)v] = idTI[(size_t)pnl];
            for (int j = v + 1; j < 4; ++j) {
                const int bj = 4 * qw + j;
                int prev = -1;
                for (int k = pnl; k < bj; ++k) {
                    const int u = add(TASK_U, k, iv, bj, 10.0 * NBK + bj);
                    dep(xdone[(size_t)(k - 4 * qw)], u);
                    if (bj == k + 1) { for (int q2 = 0; q2 < ntu(k); ++q2) dep(idTU[(size_t)k * MEGA_NTU + q2], u); }
                    else for (auto* v : {&idT, &idT2}) dep((*v)[(size_t)k * NBK + bj], u);
                    dep(prev, u);
                    prev = u;
                }
                const int tv = add(TASK_T, bj, iv, 0, 10.0 * NBK + bj);
                dep(idD[(size_t)bj], tv);
                dep(prev, tv);
                xdone[(size_t)j] = tv;
            }
        }
    static const double BLW = knob_double("STBA_MEGA_BLEVEL", 1.0);
    if (BLW > 0.0) {
        // bottom level (longest path to the end of the graph) as the priority: HLFET list scheduling
        std::vector<int> indeg2(nodes.size()), topo;
        topo.reserve(nodes.size());
        for (size_t k = 0; k < nodes.size(); ++k) { indeg2[k] = nodes[k].indeg; if (indeg2[k] == 0) topo.push_back((int)k); }
        for (size_t h = 0; h < topo.size(); ++h)
            for (int sidx : nodes[(size_t)topo[h]].succ)
                if (--indeg2[(size_t)sidx] == 0) topo.push_back(sidx);
        std::vector<double> bl(nodes.size(), 0.0);
        for (size_t h = topo.size(); h-- > 0;) {
            const int k = topo[h];
            double m = 0.0;
            for (int sidx : nodes[(size_t)k].succ) m = std::max(m, bl[(size_t)sidx]);
            bl[(size_t)k] = nodes[(size_t)k].dur + m;
        }
        for (size_t k = 0; k < nodes.size(); ++k) nodes[k].prio = nodes[k].prio * (1.0 - BLW) * 5.0 - BLW * bl[k];
    }
    // Event-driven list scheduling.  A ticket list replays the model if the tickets are sorted by the time
    // their worker became FREE (= the moment a workgroup picks its next ticket), not by the task's start:
    // a workgroup that picks a ticket early parks on it until its inputs arrive.
    // Liveness: a ticket occupies its model worker from pick to end, so at most W - 1 tickets that come later in the
    // topological (start time) order can precede any ticket in its list, W = the list's model workers; with at least W
    // real workgroups per list one of them always reaches the earliest unfinished task.
    // The model is faithful (C5, one class: 2.55 ms predicted, 2.53-2.60 ms measured; tools/mega_trace.py prints the
    // drift), which makes it the place to try policies -- tools/sim_sweep.py, no GPU.  Tried and rejected there AND on the
    // machine in rounds 1-2 (and removed from the code since): workers reserved for the tasks of the critical chains,
    // priority boosts for those rows, an urgent list per XCD claimed by whichever workgroup is free (a parked workgroup
    // starts a critical task the moment its last flag flips; a claimed task pays 2-5 us per hand-over), the last update of
    // a tile fused into the panel solve that consumes it.  DESIGN.md 4 has the numbers.
    static const double ADV_D = knob_double("STBA_MEGA_ADV_D", 0.0), ADV_TU = knob_double("STBA_MEGA_ADV_TU", 0.0);
    typedef std::pair<double, int> PI;
    typedef std::priority_queue<PI, std::vector<PI>, std::greater<PI>> Heap;
    std::vector<Heap> ready_h((size_t)nl);
    Heap events;                                // (time, node) a task ends | (time, -(node + 1)) the last remote input of a task arrives
    std::vector<std::vector<double>> idle_since((size_t)nl, std::vector<double>((size_t)mach.wg, 0.0));   // LIFO
    double now = 0.0;
    struct Pick { double key, start; int node; };
    std::vector<std::vector<Pick>> order((size_t)nl);
    auto push_ready = [&](int k) { ready_h[(size_t)nodes[(size_t)k].q].push(PI(nodes[(size_t)k].prio, k)); };
    for (int k = 0; k < (int)nodes.size(); ++k)
        if (nodes[(size_t)k].indeg == 0) push_ready(k);
    double makespan = 0.0;
    std::vector<double> tiles_in((size_t)std::max(1, n_gpus), 0.0);
    for (;;) {
        for (int l = 0; l < nl; ++l) {
            std::vector<double>& idl = idle_since[(size_t)l];
            Heap& hb = ready_h[(size_t)l];
            while (!idl.empty() && !hb.empty()) {
                const int k = hb.top().second;
                hb.pop();
                Node& nd = nodes[(size_t)k];
                // (ADVANCE: the tickets of the critical hand-off -- D and TU -- are moved up in their list by so many microseconds of model
                // time, so that a workgroup is already parked on them when their last input arrives instead of reaching them a 42 us
                // update task later; a parked workgroup idles, which is cheap: at most five tickets per panel and list)
                const int ty = nd.tk.x & 0xff;
                const double adv = ty == TASK_D ? ADV_D : ty == TASK_TU ? ADV_TU : 0.0;
                order[(size_t)l].push_back({idl.back() - adv, now, k});
                idl.pop_back();
                nd.start = now;
                events.push(PI(now + nd.dur, k));
            }
        }
        if (events.empty()) break;
        const PI ev = events.top();
        events.pop();
        now = ev.first;
        if (ev.second < 0) {                    // (sharded model) the last remote input of a task has arrived
            const int k = -(ev.second + 1);
            nodes[(size_t)k].ready = now;
            push_ready(k);
            continue;
        }
        makespan = now;
        const Node& nd = nodes[(size_t)ev.second];
        idle_since[(size_t)nd.q].push_back(now);
        const int g0 = gpu_of(nd.q);
        unsigned sent_to = 0;                   // GPUs this task's output tile has been shipped to
        for (int sidx : nd.succ) {
            Node& sn = nodes[(size_t)sidx];
            double at = now;
            const int g1 = gpu_of(sn.q);
            if (g1 != g0) {
                at += shard->hop_us + shard->tile_us;
                shard->cross_edges += 1.0;
                if (!(sent_to >> g1 & 1u)) { sent_to |= 1u << g1; tiles_in[(size_t)g1] += 1.0; }
            }
            if (at > sn.avail) { sn.avail = at; sn.last_pred = ev.second; }
            if (--sn.indeg == 0) {
                if (sn.avail <= now) { sn.ready = now; push_ready(sidx); }
                else events.push(PI(sn.avail, -(sidx + 1)));
            }
        }
    }
    if (makespan_out) *makespan_out = makespan;
    if (shard) shard->tiles_in_max = *std::max_element(tiles_in.begin(), tiles_in.end());
    if (knob_str("STBA_MEGA_SIMDBG")) {
        static const char* NM[6] = {"D", "T", "TI", "U", "Uq", "TU"};
        for (int b = 0; b + 1 < NBK; ++b) {
            const Node& d0 = nodes[(size_t)idD[(size_t)b]];
            const Node& d1 = nodes[(size_t)idD[(size_t)b + 1]];
            const Node& tu = nodes[(size_t)idTU[(size_t)b * MEGA_NTU + ntu(b) - 1]];
            const double e0 = d0.start + d0.dur;
            fprintf(stderr, "b=%2d step %6.1f | TU ready %+6.1f start %+6.1f | D+1 ready %+6.1f start %+6.1f", b, d1.start - d0.start,
                    tu.ready - e0, tu.start - e0, d1.ready - e0, d1.start - e0);
            int k = idTU[(size_t)b * MEGA_NTU + ntu(b) - 1];
            for (int hop = 0; hop < 4 && k >= 0; ++hop) {       // walk back along the last-arriving inputs
                const Node& nd = nodes[(size_t)k];
                fprintf(stderr, " <- %s(%d;%d,%d) rdy %+.1f st %+.1f", NM[nd.tk.x & 0xff], nd.tk.y, nd.tk.z, nd.tk.w, nd.ready - e0, nd.start - e0);
                k = nd.last_pred;
            }
            fprintf(stderr, "\n");
        }
    }
    out.clear();
    for (int l = 0; l < nl; ++l) {
        std::stable_sort(order[(size_t)l].begin(), order[(size_t)l].end(), [](const Pick& x, const Pick& y) {
            return x.key != y.key ? x.key < y.key : x.start < y.start;
        });
        lstart[l] = (int)out.size();
        for (const Pick& pk : order[(size_t)l]) {
            out.push_back(nodes[(size_t)pk.node].tk);
            if (sim_start) sim_start->push_back((float)pk.start);
        }
    }
    lstart[nl] = (int)out.size();
}

// diagnostics (tools/sim_sweep.py): the simulated makespan of the task graph, no GPU involved
double chol_schedule_makespan(int nblk, int nq, int wg_per_q) {
    std::vector<int4> tasks;
    std::vector<int> lstart((size_t)nq + 1);
    double ms = 0.0;
    MegaMachine m;
    m.nq = nq; m.wg = wg_per_q;
    mega_build_tasks(nblk, m, tasks, lstart.data(), nullptr, &ms);
    return ms;
}

// the same task graph on n_gpus x n_xcd queues: tile rows block-cyclic over the GPUs, a dependency that crosses GPUs
// costs hop_us + tile_us (see MegaShardModel); out3 = {makespan us, cross-GPU dependencies, remote tiles fetched by the
// busiest GPU}
void chol_shard_model(int nblk, int n_gpus, int n_xcd, int wg_per_q, int rows_per_group, double hop_us, double tile_us, double* out3) {
    std::vector<int4> tasks;
    std::vector<int> lstart((size_t)n_gpus * n_xcd + 1);
    MegaShardModel sh;
    sh.n_gpus = n_gpus; sh.rows_per_group = rows_per_group; sh.hop_us = hop_us; sh.tile_us = tile_us;
    double ms = 0.0;
    MegaMachine m;
    m.nq = n_gpus * n_xcd; m.wg = wg_per_q;
    mega_build_tasks(nblk, m, tasks, lstart.data(), nullptr, &ms, 0, &sh);
    out3[0] = ms; out3[1] = sh.cross_edges; out3[2] = sh.tiles_in_max;
}

// tile row -> GPU of the block-cyclic distribution the model uses
int chol_shard_row_owner(int row, int n_gpus, int rows_per_group) { return (row / std::max(1, rows_per_group)) % std::max(1, n_gpus); }

// ------------------------------------------------------------------------------------------
// backward substitution, one launch per 128-unknown block b (from the last block up):
//   workgroup 0      applies the previous block's solution x_{b+1} to ITS OWN 128 right-hand-side
//                    entries, then x_b = (L_bb^-T) y_b as a 128x128 GEMV with the inverse transpose
//                    the panel-solve kernel produced during the factorisation;
//   workgroups 1..   apply x_{b+1} to the remaining entries y[0 : k0), 128 entries each (GEMV with the row panel).
// y lives in row lda-1 (it was forward-substituted for free by the factorisation).
constexpr int BWD_ROW_CHUNKS = 8;   // the 128 panel rows are split 8 ways over the threads of a workgroup
// Every launch is one dependent step of 47, so what counts is its latency: the panel block of L and (workgroup 0) the
// inverse transpose do NOT depend on the previous step's solution and are requested first, together; only the 1 KB
// of x_{b+1} is read behind the kernel boundary.  One memory round trip per step instead of three (7.9 -> ~4 us).
__global__ __launch_bounds__(1024) void chol_bwd_step_kernel(double* __restrict__ A, int lda, int k0, int has_next,
                                                             const double* __restrict__ Xinv, double* __restrict__ x) {
    __shared__ double xs[NB];                       // previous block's solution
    __shared__ double part[BWD_ROW_CHUNKS][NB];     // partial sums of this block's own update
    __shared__ double ybuf[NB];                     // this block's right-hand side
    const int t = threadIdx.x;
    const int kn = k0 + NB;                                  // first row of the previous (next-lower) block
    const int nvn = has_next ? min(NB, (lda - 1) - kn) : 0;   // its rows that belong to the system
    constexpr int RPC = NB / BWD_ROW_CHUNKS;        // rows per chunk (16)
    const int rchunk = t >> 7, c = t & 127;
    // ---- requests that do not depend on x: 16 panel values per thread (+ 16 of the inverse transpose and y)
    const int c0 = (blockIdx.x > 0) ? (blockIdx.x - 1) * NB : k0;
    double lv[RPC];
#pragma unroll
    for (int j = 0; j < RPC; ++j) lv[j] = 0.0;
    if (has_next) {
        const double* col = A + (size_t)(kn + rchunk * RPC) * lda + c0 + c;
#pragma unroll
        for (int j = 0; j < RPC; ++j) lv[j] = col[(size_t)j * lda];
    }
    const int i = t >> 3, p8 = t & 7;
    double xv[16];
    double y_own = 0.0;
    const int nv = min(NB, (lda - 1) - k0);         // rows of this block that belong to the system
    if (blockIdx.x == 0) {
        const double* xr = Xinv + (size_t)i * NB + p8 * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) xv[j] = xr[j];
        if (t < NB) y_own = (t < nv) ? A[(size_t)(lda - 1) * lda + k0 + t] : 0.0;
    }
    double y_upd = 0.0;
    if (blockIdx.x > 0 && t < NB) y_upd = A[(size_t)(lda - 1) * lda + c0 + t];
    // ---- the previous block's solution
    if (t < NB) xs[t] = (t < nvn) ? x[kn + t] : 0.0;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < RPC; ++j) s = fma(lv[j], xs[rchunk * RPC + j], s);
    part[rchunk][c] = s;
    __syncthreads();
    if (blockIdx.x > 0) {
        // update role: this workgroup OWNS the 128 entries y[c0 .. c0+128) of block column blockIdx.x - 1.
        // Thread (rchunk, column) summed 16 panel rows, the 8 partial sums of a column meet in LDS and are added
        // in a fixed order: no atomics, so the solution is bitwise reproducible (two ranks that factor the same
        // system get the same step).
        if (t < NB) {
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < BWD_ROW_CHUNKS; ++q) acc += part[q][t];
            A[(size_t)(lda - 1) * lda + c0 + t] = y_upd - acc;
        }
        return;
    }
    if (t < NB) {
        double yv = y_own;
#pragma unroll
        for (int q = 0; q < BWD_ROW_CHUNKS; ++q) yv -= part[q][t];
        ybuf[t] = (t < nv) ? yv : 0.0;
    }
    __syncthreads();
    // x_i = sum_{j >= i} X[i][j] y_j : 8 threads per row, 16 columns each, shuffle tree
    double r = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) r = fma(xv[j], ybuf[p8 * 16 + j], r);
    r += __shfl_xor(r, 1, 64);
    r += __shfl_xor(r, 2, 64);
    r += __shfl_xor(r, 4, 64);
    if (p8 == 0) x[k0 + i] = (i < nv) ? r : 0.0;
}

// Wide backward steps.  With the inverse transposes V_q of the 512 x 512 diagonal blocks (built by extra tasks of the
// persistent kernel, see chol_mega_kernel) a step covers four panels:  x_B = V_q y_B  (solve), then
// y[0 : row0) -= L[B, 0 : row0)^T x_B  (apply).  Both are GEMVs spread over many workgroups, partial sums are added in a
// fixed order (no atomics: bitwise reproducible).
// apply: workgroup g owns the 32 entries y[32 g .. 32 g + 32); thread (rc, c) = (t >> 5, t & 31) sums RPC source rows
template <int RPC>
__global__ __launch_bounds__(1024) void chol_bwd_apply_kernel(double* __restrict__ A, int lda, int src_row0, const double* __restrict__ x) {
    __shared__ double xs[32 * RPC];
    __shared__ double part[32][33];
    const int t = threadIdx.x, rc = t >> 5, c = t & 31;
    const int c0 = blockIdx.x * 32;
    const double* col = A + (size_t)(src_row0 + rc * RPC) * lda + c0 + c;
    double lv[RPC];
#pragma unroll
    for (int j = 0; j < RPC; ++j) lv[j] = col[(size_t)j * lda];
    for (int e = t; e < 32 * RPC; e += 1024) xs[e] = x[src_row0 + e];
    double yv = 0.0;
    if (t < 32) yv = A[(size_t)(lda - 1) * lda + c0 + t];
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < RPC; ++j) s = fma(lv[j], xs[rc * RPC + j], s);
    part[rc][c] = s;
    __syncthreads();
    if (t < 32) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 32; ++q) acc += part[q][t];
        A[(size_t)(lda - 1) * lda + c0 + t] = yv - acc;
    }
}
// solve: workgroup g owns rows BWD_SOLVE_ROWS g .. of the block; thread (r, cc) = (t >> 5, t & 31) takes 16 columns.
// (A GEMV of this size is paced by what ONE workgroup can pull from memory, ~60 GB/s: 64 workgroups of 8 rows instead
// of 16 of 32 took the substitution from 0.139 to 0.106 ms.  The apply kernel with 16 columns per workgroup instead of
// 32: no change -- it is bound by the panel rows' total, not per workgroup.)
constexpr int BWD_SOLVE_ROWS = 8;
__global__ __launch_bounds__(32 * BWD_SOLVE_ROWS) void chol_bwd_wide_solve_kernel(const double* __restrict__ A, int lda, const double* __restrict__ V, int row0,
                                                                                  double* __restrict__ x) {
    __shared__ double ys[4 * NB];
    const int t = threadIdx.x, r = blockIdx.x * BWD_SOLVE_ROWS + (t >> 5), cc = t & 31;
    const bool live = (cc >> 3) >= (r >> 7);            // tiles left of the block diagonal are zero (and never written)
    double v[16];
    if (live) {
        const double* vr = V + (size_t)r * (4 * NB) + 16 * cc;
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = vr[j];
    }
    for (int e = t; e < 4 * NB; e += 32 * BWD_SOLVE_ROWS) ys[e] = A[(size_t)(lda - 1) * lda + row0 + e];
    __syncthreads();
    double s = 0.0;
    if (live) {
#pragma unroll
        for (int j = 0; j < 16; ++j) s = fma(v[j], ys[16 * cc + j], s);
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);
    s += __shfl_xor(s, 16, 64);
    if (cc == 0) x[row0 + r] = s;
}

// ------------------------------------------------------------------------------------------
// Per-device state shared by every engine of the process.  Two persistent kernels must not share the device: each needs
// its workgroups resident at the same time (four of one XCD for the TU quarters), and with the CUs split unevenly between
// two of them each can starve the other on a different XCD.  Factorisations of one process are therefore chained on the
// device, stream to stream, through an event; the host does not wait.  (Another PROCESS on the same device is what the
// time-out and the stage-kernel fallback are for.)
struct MegaDevice {
    std::mutex m;
    bool probed = false;
    int ncu = 0, nq = 0;                // CUs of the device, XCDs seen by the probe
    signed char xcc_queue[16];
    hipEvent_t last = nullptr;          // recorded behind the most recent persistent kernel of this device when another stream needs it ...
    hipStream_t last_stream = nullptr;  // ... which ran on this stream (null: none, or the stream is gone)
    int cooldown = 0;                   // factorisations that take the stage kernels after a time-out of the persistent program
};
static MegaDevice& mega_device(int dev) {
    static std::mutex mm;
    static std::map<int, MegaDevice> devs;
    std::lock_guard<std::mutex> g(mm);
    return devs[dev];
}

// once per device: which XCDs do the workgroups of a ncu-wide launch land on?
static int mega_device_init(MegaDevice& D, hipStream_t st) {
    if (D.probed) return STBA_OK;
    int dev = 0;
    hipDeviceProp_t prop;
    STBA_HIP(hipGetDevice(&dev));
    STBA_HIP(hipGetDeviceProperties(&prop, dev));
    D.ncu = prop.multiProcessorCount;
    int* probe = nullptr;
    std::vector<int> h((size_t)D.ncu);
    STBA_HIP(hipMalloc(reinterpret_cast<void**>(&probe), h.size() * sizeof(int)));
    hipLaunchKernelGGL(xcc_probe_kernel, dim3(D.ncu), dim3(512), 0, st, probe);
    hipError_t e = hipMemcpyAsync(h.data(), probe, h.size() * sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(probe);
    STBA_HIP(e);
    for (int i = 0; i < 16; ++i) D.xcc_queue[i] = -1;
    int per_xcc[16] = {0};
    D.nq = 0;
    for (int x : h) {
        if (D.xcc_queue[x & 15] < 0) D.xcc_queue[x & 15] = (signed char)D.nq++;
        per_xcc[x & 15]++;
    }
    for (int i = 0; i < 16; ++i)
        if (D.xcc_queue[i] >= 0 && per_xcc[i] < MEGA_NTU)
            return fail(STBA_ERR_HIP, "chol: fewer than 10 workgroups per XCD (the TU tasks of a step need one each)");
    if (!D.last) STBA_HIP(hipEventCreateWithFlags(&D.last, hipEventDisableTiming));
    D.probed = true;
    return STBA_OK;
}

// a stream is about to be destroyed (its owner has synchronised it): nobody must record an event on it any more
void chol_forget_stream(hipStream_t st) {
    for (int dev = 0; dev < 64; ++dev) {
        MegaDevice& D = mega_device(dev);
        std::lock_guard<std::mutex> g(D.m);
        if (D.last_stream == st) D.last_stream = nullptr;
        if (dev > 0 && !D.probed) break;              // (devices are numbered from 0; an unprobed one has never run a factorisation)
    }
}

static std::atomic<int> g_timeouts{0};
static std::atomic<long long> g_spin_override{0};     // ticks of the 100 MHz clock; 0: automatic
void chol_set_spin_limit_us(double us) { g_spin_override.store(us > 0.0 ? std::max(1LL, (long long)(us * 100.0)) : 0LL); }
void chol_note_timeout() {
    g_timeouts.fetch_add(1);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    MegaDevice& D = mega_device(dev);
    std::lock_guard<std::mutex> g(D.m);
    D.cooldown = 64;
}
void chol_count_timeout() { g_timeouts.fetch_add(1); }     // (counted, the device not marked: several ranks keep their own cool-down)
// several ranks: ANOTHER rank's factorisation gave up.  This rank goes through the stage kernels for the same 64 factorisations, so that
// all ranks keep factoring the identical reduced system with the identical schedule -- the two schedules differ in the last bits, and
// every rank must hold the same camera blocks (include/stba.h, stba_ba_set_allreduce).  Not counted as a time-out of this rank.
void chol_note_peer_timeout() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    MegaDevice& D = mega_device(dev);
    std::lock_guard<std::mutex> g(D.m);
    D.cooldown = 64;
}
int chol_timeout_count() { return g_timeouts.load(); }

// which schedule a factorisation runs through
enum { CHOL_PERSISTENT = 0, CHOL_STAGES = 1 };

static int chol_run(double* A, int lda, int n, double* x_dev, int* flag_dev, hipStream_t st, CholProfile* prof,
                    hipEvent_t mid_event = nullptr, int schedule = CHOL_PERSISTENT, hipEvent_t pre_event = nullptr) {
    if (lda % NB != 0 || lda < n + 1) return fail(STBA_ERR_INVALID_ARGUMENT, "chol: bad padded dimension");
    const int nblk = lda / NB;
    int cur_dev = 0;
    STBA_HIP(hipGetDevice(&cur_dev));
    if (!prof && schedule == CHOL_PERSISTENT) {
        // after a time-out (chol_note_timeout) the device is taken to be shared: the next factorisations do not try again
        MegaDevice& D = mega_device(cur_dev);
        std::lock_guard<std::mutex> g(D.m);
        if (D.cooldown > 0) { --D.cooldown; schedule = CHOL_STAGES; }
    }
    const bool stages = prof != nullptr || schedule == CHOL_STAGES;
    // per diagonal block: its inverse transpose (written by the panel solve, read by the backward pass)
    // followed by the inverses of its eight 16x16 diagonal tiles (written by the diagonal kernel, read
    // by the panel solve)
    constexpr size_t LINV_STRIDE = (size_t)NB * NB + 8 * 256;
    // workspace and task plan are per (host thread, device): a thread that switches devices, or a second GPU in
    // the same process, must not reuse another device's pointers or XCD probe
    struct MegaPlan {
        int nblk = 0, nwide = -1, ntasks = 0;
        int lstart[MEGA_MAX_Q + 1] = {0};
        int4* tasks = nullptr; int* sync = nullptr; size_t sync_ints = 0;
        double makespan_us = 0.0;
        std::vector<float> sim_start;     // simulated start time of every task (written to the trace file)
    };
    struct DevWs { double* linv = nullptr; int linv_blocks = 0; double* vbuf = nullptr; int vbuf_blocks = 0; MegaPlan plan; };
    static thread_local std::map<int, DevWs> ws_by_dev;
    DevWs& ws = ws_by_dev[cur_dev];
    double*& linv = ws.linv;
    int& linv_blocks = ws.linv_blocks;
    if (linv_blocks < nblk) {
        if (linv) (void)hipFree(linv);
        linv = nullptr; linv_blocks = 0;
        STBA_HIP(hipMalloc(reinterpret_cast<void**>(&linv), (size_t)nblk * LINV_STRIDE * sizeof(double)));
        linv_blocks = nblk;
    }
    // wide (4-panel) inverse blocks for the backward substitution: every panel but the last 1..4 (the tail keeps
    // single-panel steps, so that no inverse-block task runs behind the last diagonal block)
    const int nwide = stages ? 0 : (nblk - 1) / 4;
    if (ws.vbuf_blocks < nwide) {
        if (ws.vbuf) (void)hipFree(ws.vbuf);
        ws.vbuf = nullptr; ws.vbuf_blocks = 0;
        STBA_HIP(hipMalloc(reinterpret_cast<void**>(&ws.vbuf), (size_t)nwide * 16 * NB * NB * sizeof(double)));
        ws.vbuf_blocks = nwide;
    }
    // panel of block b: diagonal kernel + panel solve of the (nblk - b - 1) * 8 row groups below it
    static DeviceOnce diag_attr;
    STBA_TRY(diag_attr.run([]() -> int {
        STBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_diag_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Diag2Smem)));
        return STBA_OK;
    }));
    auto launch_panel_diag = [&](int b, hipStream_t s_) {
        double* li = linv + (size_t)b * LINV_STRIDE;
        hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(512), sizeof(Diag2Smem), s_, A, lda, b * NB, n, flag_dev, li + NB * NB);
    };
    auto launch_panel_trsm = [&](int b, hipStream_t s_) {
        double* li = linv + (size_t)b * LINV_STRIDE;
        const int groups = (nblk - b - 1) * (NB / 16);
        hipLaunchKernelGGL(chol_trsm_kernel, dim3(groups + NB / 16), dim3(64), 0, s_, A, lda, b * NB, groups, li, li + NB * NB);
    };
    std::vector<hipEvent_t> ev;
    if (prof) {
        ev.resize((size_t)nblk * 4 + 2);
        for (auto& e : ev) STBA_HIP(hipEventCreate(&e));
        memset(prof, 0, sizeof *prof);
    }
    auto mark = [&](size_t k) -> int { if (prof) STBA_HIP(hipEventRecord(ev[k], st)); return STBA_OK; };
    if (stages) {
        {   // the per-thread workspace (linv) is shared by every stream this host thread factors on: order this factorisation
            // behind the previous one's backward substitution if that ran on another stream (as the persistent branch does)
            MegaDevice& D = mega_device(cur_dev);
            std::lock_guard<std::mutex> dev_lock(D.m);
            if (!D.last) STBA_HIP(hipEventCreateWithFlags(&D.last, hipEventDisableTiming));
            if (D.last_stream != nullptr && D.last_stream != st) {
                STBA_HIP(hipEventRecord(D.last, D.last_stream));
                STBA_HIP(hipStreamWaitEvent(st, D.last, 0));
            }
            D.last_stream = st;
        }
        STBA_HIP(hipMemsetAsync(flag_dev, 0, sizeof(int), st));
        // STAGE KERNELS: one kernel per stage and panel, in order on the caller's stream.  The diagnostic schedule of
        // stba_cholesky_profile (per-class event times) and the FALLBACK of the persistent program: it needs nothing
        // resident, so it always finishes, whoever else uses the device (see chol_factor_solve_robust).
        for (int b = 0; b < nblk; ++b) {
            const int k0 = b * NB;
            const int mt = nblk - b - 1;
            STBA_TRY(mark(4 * (size_t)b + 0));
            launch_panel_diag(b, st);
            STBA_TRY(mark(4 * (size_t)b + 1));
            launch_panel_trsm(b, st);
            STBA_TRY(mark(4 * (size_t)b + 2));
            if (mt > 0) launch_syrk(A, lda, k0, 0, mt * (mt + 1) / 2, st);
            STBA_TRY(mark(4 * (size_t)b + 3));
            if (mt > 0 && prof) {
                const double m = std::max(0, n - (k0 + NB));
                prof->syrk_flops += m * (m + 1.0) * NB;
                prof->syrk_flops_padded += (double)(mt * (mt + 1) / 2) * 2.0 * NB * NB * NB;
                prof->syrk_launches += 1;
            }
        }
    } else {
        // production: the persistent dataflow program (see chol_mega_kernel)
        MegaDevice& D = mega_device(cur_dev);
        std::lock_guard<std::mutex> dev_lock(D.m);
        STBA_TRY(mega_device_init(D, st));
        MegaPlan& plan = ws.plan;
        if (plan.nblk != nblk || plan.nwide != nwide) {
            if (plan.tasks) (void)hipFree(plan.tasks);
            if (plan.sync) (void)hipFree(plan.sync);
            plan = MegaPlan();
            MegaMachine mach;
            mach.nq = D.nq;
            mach.wg = std::max(MEGA_NTU, D.ncu / D.nq);
            std::vector<int4> tasks;
            mega_build_tasks(nblk, mach, tasks, plan.lstart, &plan.sim_start, &plan.makespan_us, nwide);
            STBA_HIP(hipMalloc(reinterpret_cast<void**>(&plan.tasks), tasks.size() * sizeof(int4)));
            STBA_HIP(hipMemcpy(plan.tasks, tasks.data(), tasks.size() * sizeof(int4), hipMemcpyHostToDevice));
            plan.sync_ints = MEGA_SYNC_HDR + 3 * (size_t)nblk + 2 * (size_t)nblk * (nblk + 4 * nwide);
            STBA_HIP(hipMalloc(reinterpret_cast<void**>(&plan.sync), plan.sync_ints * sizeof(int)));
            plan.ntasks = (int)tasks.size();
            plan.nblk = nblk; plan.nwide = nwide;
        }
        hipLaunchKernelGGL(chol_reset_kernel, dim3((unsigned)((plan.sync_ints + 255) / 256)), dim3(256), 0, st, plan.sync, (int)plan.sync_ints, flag_dev);
        MegaArgs ma;
        ma.A = A; ma.lda = lda; ma.n = n; ma.nblk = nblk;
        ma.tasks = plan.tasks; ma.nq = D.nq; ma.sync = plan.sync;
        memcpy(ma.lstart, plan.lstart, sizeof ma.lstart);
        memcpy(ma.xcc_queue, D.xcc_queue, sizeof ma.xcc_queue);
        ma.linv = linv; ma.linv_stride = LINV_STRIDE; ma.flag = flag_dev;
        ma.vbuf = ws.vbuf; ma.nwide = nwide;
        // a dependency that has not arrived after many times the predicted makespan never will: some workgroup of the
        // program is not resident (another process holds the CUs).  The host then takes the stage kernels.
        // (25 ms at least -- a starved factorisation used to cost 100 ms before the 3.7 ms fallback ran; 40 x the makespan covers a
        // profiler's PMC passes, which slow the kernel 2-2.5x; chol_set_spin_limit_us overrides it -- tests provoke the fallback)
        ma.spin_limit = (long long)std::max(25e3, 40.0 * plan.makespan_us) * 100LL;
        if (g_spin_override.load() > 0) ma.spin_limit = g_spin_override.load();
        static const char* TRACE = knob_str("STBA_MEGA_TRACE");
        ma.trace = nullptr;
        if (TRACE) STBA_HIP(hipMalloc(reinterpret_cast<void**>(&ma.trace), (size_t)plan.ntasks * 8 * sizeof(long long)));
        struct TraceGuard { long long*& p; ~TraceGuard() { if (p) { (void)hipFree(p); p = nullptr; } } } trace_guard{ma.trace};
        static DeviceOnce attr_set;
        STBA_TRY(attr_set.run([]() -> int {
            STBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_mega_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, MEGA_SMEM_BYTES));
            return STBA_OK;
        }));
        // (the previous factorisation of this device, if it went to ANOTHER stream, must have finished; on the same stream the
        // order is there already -- and a wait on an event costs ~10 us of idle GPU even when the event has long fired)
        // The event is recorded on the PREVIOUS stream only now, when a factorisation arrives on another one (it then also
        // covers what that stream was given since, which is harmless): an engine that keeps to its stream never pays for an
        // event record behind its persistent kernel (~6 us of idle GPU in front of the backward substitution).
        if (D.last_stream != nullptr && D.last_stream != st) {
            STBA_HIP(hipEventRecord(D.last, D.last_stream));
            STBA_HIP(hipStreamWaitEvent(st, D.last, 0));
        }
        if (pre_event) STBA_HIP(hipEventRecord(pre_event, st));    // (timing only: the persistent kernel alone, without the flag reset in front)
        hipLaunchKernelGGL(chol_mega_kernel, dim3(D.ncu), dim3(512), MEGA_SMEM_BYTES, st, ma);
        D.last_stream = st;
        if (TRACE) {    // debugging aid: dump the task timeline of this factorisation (tools/mega_trace.py)
            std::vector<long long> h((size_t)plan.ntasks * 8);
            std::vector<int4> ht((size_t)plan.ntasks);
            STBA_HIP(hipStreamSynchronize(st));
            STBA_HIP(hipMemcpy(h.data(), ma.trace, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
            STBA_HIP(hipMemcpy(ht.data(), plan.tasks, ht.size() * sizeof(int4), hipMemcpyDeviceToHost));
            if (FILE* f = fopen(TRACE, "wb")) {
                fwrite(&plan.ntasks, sizeof(int), 1, f);
                fwrite(ht.data(), sizeof(int4), ht.size(), f);
                fwrite(h.data(), sizeof(long long), h.size(), f);
                fwrite(plan.sim_start.data(), sizeof(float), plan.sim_start.size(), f);
                fclose(f);
            }
        }
    }
    STBA_TRY(mark((size_t)nblk * 4));
    if (mid_event) STBA_HIP(hipEventRecord(mid_event, st));     // factorisation | backward substitution
    // backward substitution: one small launch per block.  (A persistent variant -- one chain workgroup plus
    // GEMV workers synchronised with flags -- was measured at 13.7 us per block against 7.7 us for these
    // launches: every hand-off through memory costs ~2 us on this part, a dependent launch ~3 us.)
    for (int b = nblk - 1; b >= 4 * nwide; --b) {
        const int k0 = b * NB;
        const int has_next = (b < nblk - 1) ? 1 : 0;
        const int grid = 1 + (has_next ? b : 0);       // workgroup 0: this block; workgroup 1 + j: block column j < b
        hipLaunchKernelGGL(chol_bwd_step_kernel, dim3(grid), dim3(1024), 0, st, A, lda, k0, has_next, linv + (size_t)b * LINV_STRIDE, x_dev);
    }
    // wide steps: apply the solution found last to everything above it, then solve four panels at once
    for (int qw = nwide - 1; qw >= 0; --qw) {
        const int row0 = qw * 4 * NB;
        if (qw == nwide - 1) hipLaunchKernelGGL(chol_bwd_apply_kernel<4>, dim3((row0 + 4 * NB) / 32), dim3(1024), 0, st, A, lda, row0 + 4 * NB, x_dev);
        else hipLaunchKernelGGL(chol_bwd_apply_kernel<16>, dim3((row0 + 4 * NB) / 32), dim3(1024), 0, st, A, lda, row0 + 4 * NB, x_dev);
        hipLaunchKernelGGL(chol_bwd_wide_solve_kernel, dim3(4 * NB / BWD_SOLVE_ROWS), dim3(32 * BWD_SOLVE_ROWS), 0, st, A, lda, ws.vbuf + (size_t)qw * 16 * NB * NB, row0, x_dev);
    }
    STBA_TRY(mark((size_t)nblk * 4 + 1));
    STBA_HIP(hipGetLastError());
    if (prof) {
        STBA_HIP(hipStreamSynchronize(st));
        float ms = 0.f;
        for (int b = 0; b < nblk; ++b) {
            (void)hipEventElapsedTime(&ms, ev[4 * (size_t)b + 0], ev[4 * (size_t)b + 1]); prof->ms_diag += ms;
            (void)hipEventElapsedTime(&ms, ev[4 * (size_t)b + 1], ev[4 * (size_t)b + 2]); prof->ms_trsm += ms;
            (void)hipEventElapsedTime(&ms, ev[4 * (size_t)b + 2], ev[4 * (size_t)b + 3]); prof->ms_syrk += ms;
        }
        (void)hipEventElapsedTime(&ms, ev[(size_t)nblk * 4], ev[(size_t)nblk * 4 + 1]); prof->ms_bwd = ms;
        for (auto& e : ev) (void)hipEventDestroy(e);
    }
    return STBA_OK;
}

// Explicit inverse of a small SPD matrix through a PARTIAL factorisation with the stage kernels (the pose graph's coarse
// operator, pg_engine.hip: 6 unknowns per group of nodes, a few hundred to ~1500 in all; applied once per PCG iteration as
// a dense product, so it is wanted explicitly).  W is ldw x ldw, ldw = 2 np, np a multiple of 128:
//        [ A   . ]     panels 0 .. np/128 - 1 of the blocked right-looking      [ L       .     ]
//        [ I   0 ]     factorisation, applied to ALL 2 np rows            ->    [ L^-T   -A^-1  ]
// the panel solves turn the identity rows into X = I L^-T, and the trailing updates subtract X X^T = A^-1 from the lower
// right block (lower triangle).  No factorisation of the lower right block follows, so it survives.
// workspace: (NB NB + 8 256) doubles per panel (the diagonal tiles' inverses the panel solve multiplies by).
size_t chol_spd_inverse_workspace_doubles(int np) { return (size_t)(np / NB) * ((size_t)NB * NB + 8 * 256); }
int chol_spd_inverse_dev(double* W, int ldw, int np, int n_real, int* flag_dev, double* work, hipStream_t st) {
    if (np % NB != 0 || ldw != 2 * np || np <= 0) return fail(STBA_ERR_INVALID_ARGUMENT, "chol_spd_inverse_dev: bad dimensions");
    constexpr size_t LINV_STRIDE = (size_t)NB * NB + 8 * 256;
    static DeviceOnce diag_attr;
    STBA_TRY(diag_attr.run([]() -> int {
        STBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_diag_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Diag2Smem)));
        return STBA_OK;
    }));
    STBA_HIP(hipMemsetAsync(flag_dev, 0, sizeof(int), st));
    const int nblk = ldw / NB;
    for (int b = 0; b < np / NB; ++b) {
        double* li = work + (size_t)b * LINV_STRIDE;
        const int mt = nblk - b - 1;
        hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(512), sizeof(Diag2Smem), st, W, ldw, b * NB, n_real, flag_dev, li + NB * NB);
        const int groups = mt * (NB / 16);
        hipLaunchKernelGGL(chol_trsm_kernel, dim3(groups + NB / 16), dim3(64), 0, st, W, ldw, b * NB, groups, li, li + NB * NB);
        launch_syrk(W, ldw, b * NB, 0, mt * (mt + 1) / 2, st);
    }
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

// ------------------------------------------------------------------------------------------
// Lower triangle of S = -(Y Y^T), Y (lda x kcols, leading dimension ldy) the scaled camera-landmark blocks of a bundle adjustment
// with DENSE visibility (ba_kernels.hip: Y_i = J_c^T J_p chol(Hpp^-1) per observation, zero where a camera does not see a
// landmark): the Schur complement as a symmetric rank-k product on the matrix cores instead of one LDS-atomic 6 x 6 block per pair
// of observations (whose plan costs 16 B per pair -- 1000 cameras that all see 100 000 landmarks are 5e10 pairs).  One 512-thread
// workgroup per (tile, K slice): the update task's loop (syrk_tile512_gen) over the slice's K range on a zero accumulator; with
// more than one slice the partial tiles go to a workspace and are added in slice order (no atomics: bitwise reproducible).
__global__ __launch_bounds__(512) void yyt_tile_kernel(const double* __restrict__ Y, size_t ldy, double* __restrict__ S, int lda, int ntiles, int nsplit,
                                                       int chunks_per_split, int chunks_total, double* __restrict__ ws, const int2* __restrict__ tile_ij,
                                                       int per_xcd) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // XCD-aware order: workgroup b runs on XCD b % 8; every XCD walks ONE contiguous stretch of the work list (K slice major, tiles
    // in 8 x 8 blocks of the tile grid), so the ~64 workgroups resident on an XCD at a time are one block of tiles of one K slice:
    // they stream 16 row panels of Y between them instead of 128, started together, and find each other's chunks in the XCD's L2
    const int w = (int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8);
    if (blockIdx.x / 8 >= (unsigned)per_xcd || w >= ntiles * nsplit) return;
    const int split = w / ntiles, tile = w % ntiles;
    const int I = tile_ij[tile].x, J = tile_ij[tile].y;
    const int c0 = split * chunks_per_split, c1 = min(chunks_total, c0 + chunks_per_split);
    double* Cb = (nsplit == 1) ? S + (size_t)I * NB * lda + (size_t)J * NB : ws + ((size_t)split * ntiles + tile) * NB * NB;
    const int ldc = (nsplit == 1) ? lda : NB;
    if (c1 <= c0) {         // (an empty slice still defines its partial tile)
        for (int e = threadIdx.x; e < NB * NB; e += 512) Cb[(size_t)(e / NB) * ldc + e % NB] = 0.0;
        return;
    }
    syrk_tile512_gen<128>(Cb, ldc, Y + (size_t)I * NB * ldy + (size_t)c0 * 16, ldy, Y + (size_t)J * NB * ldy + (size_t)c0 * 16, ldy, true, smem,
                          threadIdx.x, c1 - c0);
}
// S tile = sum of the slices' partial tiles, in slice order; diagonal tiles keep their lower triangle only (the strictly upper part
// of the matrix is never read, but the diagonal 6 x 6 blocks are completed by later kernels that expect zeros there)
__global__ __launch_bounds__(256) void yyt_reduce_kernel(const double* __restrict__ ws, int ntiles, int nsplit, double* __restrict__ S, int lda,
                                                         const int2* __restrict__ tile_ij) {
    const int tile = blockIdx.x / (NB * NB / 256);
    const int e = (blockIdx.x % (NB * NB / 256)) * 256 + threadIdx.x;
    const int I = tile_ij[tile].x, J = tile_ij[tile].y;
    const int r = e / NB, c = e % NB;
    double s = 0.0;
    if (nsplit == 1) s = S[((size_t)I * NB + r) * lda + (size_t)J * NB + c];
    else for (int k = 0; k < nsplit; ++k) s += ws[((size_t)k * ntiles + tile) * NB * NB + e];
    if (I == J && c > r) s = 0.0;
    S[((size_t)I * NB + r) * lda + (size_t)J * NB + c] = s;
}
// K slices: enough workgroups to fill the part several times over when the matrix has few tiles
static int yyt_splits(int ntiles, int chunks_total) {
    // (two workgroups fit a CU: 512 at a time.  1128 tiles of a 6000 x 6000 system as ONE slice each are 2.2 rounds of 512 -- the third
    // round runs a fifth full; at least eight rounds' worth of workgroups keep that tail under a few per cent)
    int ns = (4096 + ntiles - 1) / ntiles;
    ns = std::min(ns, 64);
    ns = std::min(ns, std::max(1, chunks_total / 8));     // at least 128 columns per slice
    return std::max(1, ns);
}
size_t chol_yyt_workspace_doubles(int lda, size_t kcols) {
    const int nb = lda / NB, ntiles = nb * (nb + 1) / 2;
    const int ns = yyt_splits(ntiles, (int)(kcols / 16));
    return ns == 1 ? 0 : (size_t)ns * ntiles * NB * NB;
}
// the tiles of the lower triangle in 8 x 8 blocks of the tile grid (block rows top down, blocks left to right, row-major inside)
static int yyt_tile_list(int nb, int2** out_dev) {
    static std::mutex m;
    static std::map<std::pair<int, int>, int2*> cache;      // (device, nb) -> list
    int dev = 0;
    STBA_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(m);
    auto it = cache.find({dev, nb});
    if (it == cache.end()) {
        std::vector<int2> h;
        for (int bi = 0; bi * 8 < nb; ++bi)
            for (int bj = 0; bj <= bi; ++bj)
                for (int i = bi * 8; i < std::min(nb, bi * 8 + 8); ++i)
                    for (int j = bj * 8; j < std::min(nb, bj * 8 + 8); ++j)
                        if (j <= i) h.push_back(make_int2(i, j));
        int2* d = nullptr;
        STBA_HIP(hipMalloc(reinterpret_cast<void**>(&d), h.size() * sizeof(int2)));
        STBA_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(int2), hipMemcpyHostToDevice));
        it = cache.emplace(std::make_pair(dev, nb), d).first;
    }
    *out_dev = it->second;
    return STBA_OK;
}
int chol_yyt_lower_dev(const double* Y, size_t ldy, size_t kcols, double* S, int lda, double* ws, hipStream_t st) {
    if (lda % NB != 0 || kcols % 16 != 0 || ldy < kcols) return fail(STBA_ERR_INVALID_ARGUMENT, "chol_yyt_lower_dev: bad dimensions");
    const int nb = lda / NB, ntiles = nb * (nb + 1) / 2;
    const int chunks_total = (int)(kcols / 16);
    const int ns = yyt_splits(ntiles, chunks_total);
    if (ns > 1 && !ws) return fail(STBA_ERR_INVALID_ARGUMENT, "chol_yyt_lower_dev: workspace missing");
    const int cps = (chunks_total + ns - 1) / ns;
    constexpr int LDS = 2 * (128 + 128) * 16 * (int)sizeof(double);
    static DeviceOnce attr;
    STBA_TRY(attr.run([]() -> int {
        STBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(yyt_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        return STBA_OK;
    }));
    int2* tile_ij = nullptr;
    STBA_TRY(yyt_tile_list(nb, &tile_ij));
    const int per_xcd = (ntiles * ns + 7) / 8;
    hipLaunchKernelGGL(yyt_tile_kernel, dim3((unsigned)(per_xcd * 8)), dim3(512), LDS, st, Y, ldy, S, lda, ntiles, ns, cps, chunks_total, ws, tile_ij, per_xcd);
    hipLaunchKernelGGL(yyt_reduce_kernel, dim3((unsigned)(ntiles * (NB * NB / 256))), dim3(256), 0, st, ws, ntiles, ns, S, lda, tile_ij);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

int chol_factor_solve_dev(double* A, int lda, int n, double* x_dev, int* flag_dev, hipStream_t st) {
    return chol_run(A, lda, n, x_dev, flag_dev, st, nullptr);
}

int chol_factor_solve_stages(double* A, int lda, int n, double* x_dev, int* flag_dev, hipStream_t st) {
    return chol_run(A, lda, n, x_dev, flag_dev, st, nullptr, nullptr, CHOL_STAGES);
}

int chol_factor_solve_split(double* A, int lda, int n, double* x_dev, int* flag_dev, hipStream_t st, hipEvent_t mid_event, hipEvent_t pre_event) {
    return chol_run(A, lda, n, x_dev, flag_dev, st, nullptr, mid_event, CHOL_PERSISTENT, pre_event);
}

int chol_factor_solve_profiled(double* A, int lda, int n, double* x_dev, int* flag_dev, hipStream_t st,
                               CholProfile* prof) {
    return chol_run(A, lda, n, x_dev, flag_dev, st, prof);
}


}  // namespace stba

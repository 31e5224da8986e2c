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
// Per 128-column step:
//   chol_diag_kernel   1 workgroup x 1024 threads; the 128x128 diagonal block lives in registers
//                      (4x4 per thread, 32-cyclic distribution); one LDS column broadcast and one
//                      barrier per column; sub-blocks left of / above the active column are
//                      skipped with wave-uniform (compile-time) bounds.
//   chol_trsm_kernel   one workgroup per 32 rows of the panel (so that a 6000-row panel spreads
//                      over ~190 CUs); L11 staged in LDS (129-padded), panel rows in registers.
//   chol_syrk_kernel   128x128 output tiles of the trailing lower triangle, 4 waves x (4x4)
//                      16x16 MFMA tiles, K = 128 streamed through double-buffered LDS in
//                      fragment order (conflict-free ds_read_b64 / ds_write_b64).
//   backward           one wave per diagonal block (shuffle broadcast, no barriers) + a GEMV
//                      update of the remaining right-hand side.
#include <algorithm>
#include <vector>

#include "common.hpp"

namespace stba {

constexpr int NB = CHOL_NB;
constexpr int TRSM_ROWS = 32;
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
// fast reciprocal square root: v_rsq_f64 + two Newton steps (FMA only, ~1 ulp), and sqrt from it.
__device__ __forceinline__ double fast_rsqrt(double d) {
    double y = __builtin_amdgcn_rsq(d);
    double t = d * y;
    y = fma(0.5 * y, fma(-t, y, 1.0), y);
    t = d * y;
    y = fma(0.5 * y, fma(-t, y, 1.0), y);
    return y;
}
__device__ __forceinline__ double sqrt_from_rsqrt(double d, double y) {
    double g = d * y;
    return fma(fma(-g, g, d), 0.5 * y, g);
}

// ------------------------------------------------------------------------------------------
// diagonal block: 256 threads = 4 waves, the 128x128 block lives in MFMA accumulator layout
// (8x8 tiles of 16x16; wave w owns tile rows w and 7-w = 9 lower tiles, balanced), processed in
// 16 block steps of 8 columns:
//   a. lanes holding the 8 panel columns publish them to LDS                     -> barrier
//   c. 128 row threads factor the 8x8 diagonal mini-block redundantly (right-looking,
//      division-free v_rsq_f64 + Newton chain) and solve their panel row         -> barrier
//   e. rank-8 update of the trailing tiles on the matrix cores: two v_mfma_f64_16x16x4_f64 per
//      tile, operands straight from the scaled panel rows in LDS (2 x 8 B per lane per tile
//      instead of 32 LDS reads per thread for the same flops on the VALU)
//   f. the finished panel columns are written back into the accumulators.
template <int Jt>
__device__ __forceinline__ void chol_diag_tilecol(double4v (&acc)[2][8], double (*P)[9], double (*Lp)[9], int t, int lr,
                                                  int lc, const int (&Irow)[2], int k0, int n_real, int* flag) {
    // tile column Jt is a compile-time constant so that every accumulator index is static (a runtime
    // tile index makes the compiler spill the accumulators to scratch)
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        const int j0 = 16 * Jt + 8 * h;
        // a. publish the 8 panel columns (lanes with (lc >> 3) == h)
        if ((lc >> 3) == h) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
                if (Irow[s] >= Jt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) P[16 * Irow[s] + lr + 4 * r][lc & 7] = acc[s][Jt][r];
                }
        }
        __syncthreads();
        // c. one thread per row at or below the block
        if (t < NB && t >= j0) {
            const int i = t;
            double D[8][8], y[8], p[8], l[8];
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c <= r; ++c) D[r][c] = P[j0 + r][c];
#pragma unroll
            for (int c = 0; c < 8; ++c) p[c] = P[i][c];
            bool bad = false;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                double d = D[c][c];
                if (!(d > 0.0)) { bad = bad || ((k0 + j0 + c) < n_real); d = 1.0; }
                y[c] = fast_rsqrt(d);
                D[c][c] = sqrt_from_rsqrt(d, y[c]);
#pragma unroll
                for (int r = c + 1; r < 8; ++r) D[r][c] *= y[c];
#pragma unroll
                for (int r = c + 1; r < 8; ++r)
#pragma unroll
                    for (int q = c + 1; q <= r; ++q) D[r][q] = fma(-D[r][c], D[q][c], D[r][q]);
            }
            if (bad && i == j0) atomicCAS(flag, 0, k0 + j0 + 1);
            // l G^T = p, right-looking.  Rows INSIDE the 8x8 block take the same path: row r of
            // D = G G^T solves to row r of G in its first r+1 entries; the entries right of the
            // diagonal come out as garbage, but they only ever meet (i) panel-column elements that
            // step f overwrites and (ii) strictly-upper elements nobody reads.
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                l[c] = p[c] * y[c];
#pragma unroll
                for (int q = c + 1; q < 8; ++q) p[q] = fma(-l[c], D[q][c], p[q]);
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) Lp[i][c] = l[c];
        }
        __syncthreads();
        // e. rank-8 update of the trailing tiles on the matrix cores (tile columns > Jt, and Jt itself
        //    while its right half is still trailing, i.e. h == 0)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int I = Irow[s];
            if (I > Jt || (I == Jt && h == 0)) {
                const double a0 = -Lp[16 * I + lc][lr], a1 = -Lp[16 * I + lc][4 + lr];
#pragma unroll
                for (int J = Jt; J < 8; ++J)
                    if (J <= I && (J > Jt || h == 0)) {
                        const double b0 = Lp[16 * J + lc][lr], b1 = Lp[16 * J + lc][4 + lr];
                        acc[s][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[s][J], 0, 0, 0);
                        acc[s][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[s][J], 0, 0, 0);
                    }
            }
        }
        // f. finished columns of the panel -> accumulators (rows at or below the column)
        if ((lc >> 3) == h) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
                if (Irow[s] >= Jt) {
                    const int col = 16 * Jt + lc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * Irow[s] + lr + 4 * r;
                        const double v = Lp[row][lc & 7];
                        acc[s][Jt][r] = (row >= col) ? v : acc[s][Jt][r];
                    }
                }
        }
    }
}

__global__ __launch_bounds__(256) void chol_diag_kernel(double* __restrict__ A, int lda, int k0,
                                                         int n_real, int* __restrict__ flag) {
    __shared__ double P[NB][9];
    __shared__ double Lp[NB][9];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lr = lane >> 4, lc = lane & 15;
    const int Irow[2] = {w, 7 - w};
    double4v acc[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int J = 0; J < 8; ++J) {
            const bool in = (J <= Irow[s]);
            const double* src = A + (size_t)(k0 + 16 * Irow[s] + lr) * lda + k0 + 16 * (in ? J : 0) + lc;
            double4v tmp;
            tmp[0] = in ? src[0] : 0.0;
            tmp[1] = in ? src[(size_t)4 * lda] : 0.0;
            tmp[2] = in ? src[(size_t)8 * lda] : 0.0;
            tmp[3] = in ? src[(size_t)12 * lda] : 0.0;
            acc[s][J] = tmp;
        }
    chol_diag_tilecol<0>(acc, P, Lp, t, lr, lc, Irow, k0, n_real, flag);
    chol_diag_tilecol<1>(acc, P, Lp, t, lr, lc, Irow, k0, n_real, flag);
    chol_diag_tilecol<2>(acc, P, Lp, t, lr, lc, Irow, k0, n_real, flag);
    chol_diag_tilecol<3>(acc, P, Lp, t, lr, lc, Irow, k0, n_real, flag);
    chol_diag_tilecol<4>(acc, P, Lp, t, lr, lc, Irow, k0, n_real, flag);
    chol_diag_tilecol<5>(acc, P, Lp, t, lr, lc, Irow, k0, n_real, flag);
    chol_diag_tilecol<6>(acc, P, Lp, t, lr, lc, Irow, k0, n_real, flag);
    chol_diag_tilecol<7>(acc, P, Lp, t, lr, lc, Irow, k0, n_real, flag);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int J = 0; J < 8; ++J)
            if (J <= Irow[s]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * Irow[s] + lr + 4 * r, col = 16 * J + lc;
                    if (col <= row) A[(size_t)(k0 + row) * lda + k0 + col] = acc[s][J][r];
                }
            }
}

// ------------------------------------------------------------------------------------------
// shared by the panel solve and the backward substitution: stage the 128x128 lower block in LDS
// (row stride LDS_LD, 16 B aligned rows) and invert its sixteen 8x8 diagonal blocks:
// Gi[J][k][m] = (L_JJ^-1)[k][m].  Called by all threads of the workgroup; ends with a barrier.
constexpr int LDS_LD = 130;
template <int NT>
__device__ __forceinline__ void stage_block_and_inverses(const double* __restrict__ A, int lda, int k0, int nv,
                                                         double* __restrict__ Ld, double* __restrict__ invd,
                                                         double* __restrict__ Gi, int t) {
    // one 1 KiB row per wave instruction (16 B per lane), all loads of a thread issued back to back
    constexpr int ITER = NB * (NB / 2) / NT;
    double2 v[ITER];
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
        const int idx = t + k * NT;
        const int i = idx >> 6, c = (idx & 63) * 2;
        v[k] = make_double2(0.0, 0.0);
        if (c <= i && i < nv) v[k] = *reinterpret_cast<const double2*>(&A[(size_t)(k0 + i) * lda + k0 + c]);
    }
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
        const int idx = t + k * NT;
        const int i = idx >> 6, c = (idx & 63) * 2;
        double2 w = v[k];
        if (c + 1 > i) w.y = 0.0;                    // strictly upper element of the pair
        if (i >= nv) { w.x = (c == i) ? 1.0 : 0.0; w.y = (c + 1 == i) ? 1.0 : 0.0; }
        *reinterpret_cast<double2*>(&Ld[i * LDS_LD + c]) = w;
    }
    __syncthreads();
    if (t < NB) invd[t] = 1.0 / Ld[t * LDS_LD + t];
    __syncthreads();
    if (t < NB) {
        const int J = t >> 3, m = t & 7;
        double x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            double s = (k == m) ? 1.0 : 0.0;
#pragma unroll
            for (int q = 0; q < k; ++q) s = fma(-Ld[(8 * J + k) * LDS_LD + 8 * J + q], (q >= m) ? x[q] : 0.0, s);
            x[k] = (k >= m) ? s * invd[8 * J + k] : 0.0;
            Gi[(J * 8 + k) * 8 + m] = x[k];
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// X = A21 * L11^-T for TRSM_ROWS rows per workgroup, in place, in column blocks of width 8.
// 256 threads: thread (r, cg) owns row r = t >> 3 and columns cg + 8m (m < 16), i.e. exactly one
// column of every 8-wide block; the 8 threads of a row are 8 consecutive lanes, so a block step
// needs only lane shuffles (no LDS hand-off, no barrier):
//   x = T_J * Gi_J^T (8 shuffles), then columns to the right -= X_J * L11[c, J-block]^T.
template <int J>
__device__ __forceinline__ void chol_trsm_block(double (&acc)[16], const double* __restrict__ Ld,
                                                const double* __restrict__ Gi, int lane, int cg) {
    const int base = lane & ~7;
    const double tv = acc[J];
    double x = 0.0;
    const double* gi = Gi + (J * 8 + cg) * 8;      // row cg of the inverse block: Gi[J][cg][m]
#pragma unroll
    for (int m = 0; m < 8; ++m) x = fma(__shfl(tv, base + m, 64), gi[m], x);
    acc[J] = x;
    double xs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) xs[k] = __shfl(x, base + k, 64);
#pragma unroll
    for (int m = J + 1; m < 16; ++m) {
        const double2* lrow = reinterpret_cast<const double2*>(Ld + (cg + 8 * m) * LDS_LD + 8 * J);
        double v = acc[m];
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
            const double2 l2 = lrow[k2];
            v = fma(-xs[2 * k2], l2.x, v);
            v = fma(-xs[2 * k2 + 1], l2.y, v);
        }
        acc[m] = v;
    }
}

__global__ __launch_bounds__(256) void chol_trsm_kernel(double* __restrict__ A, int lda, int k0, int n_row_wgs,
                                                        double* __restrict__ Xinv) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Ld = smem;                          // [128][130]
    double* Gi = smem + NB * LDS_LD;            // [16][8][8]
    double* invd = Gi + 16 * 64;                // [128]
    const int t = threadIdx.x, r = t >> 3, cg = t & 7, lane = t & 63;
    // the last 4 workgroups run the same solve on the rows of the identity: X = I * L11^-T is the
    // (upper-triangular) inverse transpose of the diagonal block, used by the backward substitution
    const bool ident = (int)blockIdx.x >= n_row_wgs;
    const int rid = ((int)blockIdx.x - n_row_wgs) * TRSM_ROWS + r;
    const int row = k0 + NB + blockIdx.x * TRSM_ROWS + r;
    const int nv = min(NB, (lda - 1) - k0);     // rows of the diagonal block that belong to the system
    double acc[16];
#pragma unroll
    for (int m = 0; m < 16; ++m)
        acc[m] = ident ? ((cg + 8 * m == rid) ? 1.0 : 0.0) : A[(size_t)row * lda + k0 + cg + 8 * m];
    stage_block_and_inverses<256>(A, lda, k0, ident ? nv : NB, Ld, invd, Gi, t);
    chol_trsm_block<0>(acc, Ld, Gi, lane, cg);   chol_trsm_block<1>(acc, Ld, Gi, lane, cg);
    chol_trsm_block<2>(acc, Ld, Gi, lane, cg);   chol_trsm_block<3>(acc, Ld, Gi, lane, cg);
    chol_trsm_block<4>(acc, Ld, Gi, lane, cg);   chol_trsm_block<5>(acc, Ld, Gi, lane, cg);
    chol_trsm_block<6>(acc, Ld, Gi, lane, cg);   chol_trsm_block<7>(acc, Ld, Gi, lane, cg);
    chol_trsm_block<8>(acc, Ld, Gi, lane, cg);   chol_trsm_block<9>(acc, Ld, Gi, lane, cg);
    chol_trsm_block<10>(acc, Ld, Gi, lane, cg);  chol_trsm_block<11>(acc, Ld, Gi, lane, cg);
    chol_trsm_block<12>(acc, Ld, Gi, lane, cg);  chol_trsm_block<13>(acc, Ld, Gi, lane, cg);
    chol_trsm_block<14>(acc, Ld, Gi, lane, cg);  chol_trsm_block<15>(acc, Ld, Gi, lane, cg);
    if (ident) {
#pragma unroll
        for (int m = 0; m < 16; ++m) Xinv[(size_t)rid * NB + cg + 8 * m] = acc[m];
    } else {
#pragma unroll
        for (int m = 0; m < 16; ++m) A[(size_t)row * lda + k0 + cg + 8 * m] = acc[m];
    }
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
    static const int force = [] { const char* e = getenv("STBA_SYRK_TM"); return e ? atoi(e) : 0; }();
    static const int big = [] { const char* e = getenv("STBA_SYRK_BIG"); return e ? atoi(e) : 1000000; }();
    const bool use64 = force == 64 || (force != 128 && tiles128 < big);
    if (use64) hipLaunchKernelGGL(chol_syrk_kernel<64>, dim3(2 * tiles128), dim3(256), 0, st, A, lda, k0, tile_mode);
    else hipLaunchKernelGGL(chol_syrk_kernel<128>, dim3(tiles128), dim3(256), 0, st, A, lda, k0, tile_mode);
}

// ------------------------------------------------------------------------------------------
// backward substitution, one launch per 128-unknown block b (from the last block up):
//   workgroup 0      applies the previous block's solution x_{b+1} to ITS OWN 128 right-hand-side
//                    entries, then x_b = (L_bb^-T) y_b as a 128x128 GEMV with the inverse transpose
//                    the panel-solve kernel produced during the factorisation;
//   workgroups 1..   apply x_{b+1} to the remaining entries y[0 : k0)   (GEMV with the row panel).
// y lives in row lda-1 (it was forward-substituted for free by the factorisation).
constexpr int BWD_ROW_CHUNKS = 8;   // the 128 panel rows are split 8 ways so that enough CUs pull the panel
__global__ __launch_bounds__(1024) void chol_bwd_step_kernel(double* __restrict__ A, int lda, int k0, int has_next,
                                                             const double* __restrict__ Xinv, double* __restrict__ x) {
    __shared__ double xs[NB];                       // previous block's solution
    __shared__ double part[BWD_ROW_CHUNKS][NB];     // partial sums of this block's own update
    __shared__ double ybuf[NB];                     // this block's right-hand side
    const int t = threadIdx.x;
    const int kn = k0 + NB;                                  // first row of the previous (next-lower) block
    const int nvn = has_next ? min(NB, (lda - 1) - kn) : 0;   // its rows that belong to the system
    if (t < NB) xs[t] = (t < nvn) ? x[kn + t] : 0.0;
    __syncthreads();
    constexpr int RPC = NB / BWD_ROW_CHUNKS;        // rows per chunk (16)
    if (blockIdx.x > 0) {
        // update role: workgroup = (column chunk of 1024, row chunk of 16); FP64 atomics into y
        const int id = blockIdx.x - 1;
        const int cchunk = id / BWD_ROW_CHUNKS, rchunk = id % BWD_ROW_CHUNKS;
        const int c = cchunk * 1024 + t;
        if (c < k0) {
            double s = 0.0;
            const double* col = A + (size_t)(kn + rchunk * RPC) * lda + c;
#pragma unroll
            for (int j = 0; j < RPC; ++j) s = fma(col[(size_t)j * lda], xs[rchunk * RPC + j], s);
            if (s != 0.0) unsafeAtomicAdd(&A[(size_t)(lda - 1) * lda + c], -s);
        }
        return;
    }
    const int nv = min(NB, (lda - 1) - k0);     // rows of this block that belong to the system
    {   // own update: thread (rchunk, column) = (t >> 7, t & 127)
        const int rchunk = t >> 7, c = t & 127;
        double s = 0.0;
        if (nvn > 0) {
            const double* col = A + (size_t)(kn + rchunk * RPC) * lda + k0 + c;
#pragma unroll
            for (int j = 0; j < RPC; ++j) s = fma(col[(size_t)j * lda], xs[rchunk * RPC + j], s);
        }
        part[rchunk][c] = s;
    }
    __syncthreads();
    if (t < NB) {
        double yv = (t < nv) ? A[(size_t)(lda - 1) * lda + k0 + t] : 0.0;
#pragma unroll
        for (int q = 0; q < BWD_ROW_CHUNKS; ++q) yv -= part[q][t];
        ybuf[t] = (t < nv) ? yv : 0.0;
    }
    __syncthreads();
    // x_i = sum_{j >= i} X[i][j] y_j : 8 threads per row, 16 columns each, shuffle tree
    const int i = t >> 3, p8 = t & 7;
    const double* xr = Xinv + (size_t)i * NB + p8 * 16;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s = fma(xr[j], ybuf[p8 * 16 + j], s);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (p8 == 0) x[k0 + i] = (i < nv) ? s : 0.0;
}

// ------------------------------------------------------------------------------------------
static int chol_run(double* A, int lda, int n, double* x_dev, int* flag_dev, hipStream_t st, CholProfile* prof) {
    if (lda % NB != 0 || lda < n + 1) return fail(STBA_ERR_INVALID_ARGUMENT, "chol: bad padded dimension");
    const int nblk = lda / NB;
    const size_t trsm_lds = sizeof(double) * (NB * LDS_LD + 16 * 64 + NB);
    static bool attr_set = false;
    if (!attr_set) {
        STBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_trsm_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)trsm_lds));
        attr_set = true;
    }
    // inverse transposes of the diagonal blocks (written by the panel solve, read by the backward pass)
    static thread_local double* linv = nullptr;
    static thread_local int linv_blocks = 0;
    if (linv_blocks < nblk) {
        if (linv) (void)hipFree(linv);
        linv = nullptr; linv_blocks = 0;
        STBA_HIP(hipMalloc(reinterpret_cast<void**>(&linv), (size_t)nblk * NB * NB * sizeof(double)));
        linv_blocks = nblk;
    }
    std::vector<hipEvent_t> ev;
    if (prof) {
        ev.resize((size_t)nblk * 4 + 2);
        for (auto& e : ev) STBA_HIP(hipEventCreate(&e));
        memset(prof, 0, sizeof *prof);
    }
    auto mark = [&](size_t k) -> int { if (prof) STBA_HIP(hipEventRecord(ev[k], st)); return STBA_OK; };
    STBA_HIP(hipMemsetAsync(flag_dev, 0, sizeof(int), st));
    if (prof) {
        // serial schedule: one kernel class at a time, so the per-class event times are clean
        for (int b = 0; b < nblk; ++b) {
            const int k0 = b * NB;
            const int mt = nblk - b - 1;
            STBA_TRY(mark(4 * (size_t)b + 0));
            hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), 0, st, A, lda, k0, n, flag_dev);
            STBA_TRY(mark(4 * (size_t)b + 1));
            hipLaunchKernelGGL(chol_trsm_kernel, dim3(mt * (NB / TRSM_ROWS) + NB / TRSM_ROWS), dim3(256), trsm_lds, st, A, lda, k0,
                               mt * (NB / TRSM_ROWS), linv + (size_t)b * NB * NB);
            STBA_TRY(mark(4 * (size_t)b + 2));
            if (mt > 0) launch_syrk(A, lda, k0, 0, mt * (mt + 1) / 2, st);
            STBA_TRY(mark(4 * (size_t)b + 3));
            if (mt > 0) {
                const double m = std::max(0, n - (k0 + NB));
                prof->syrk_flops += m * (m + 1.0) * NB;
                prof->syrk_flops_padded += (double)(mt * (mt + 1) / 2) * 2.0 * NB * NB * NB;
                prof->syrk_launches += 1;
            }
        }
    } else {
        // look-ahead schedule: the panel of step b+1 (diagonal block + panel solve, on `st`) overlaps
        // the bulk of the trailing update of step b (on a side stream); only the update of the
        // next panel's tile column sits on the critical path.
        static thread_local hipStream_t su = nullptr;
        static thread_local std::vector<hipEvent_t> evP, evN;
        static thread_local hipEvent_t evU = nullptr;
        if (!su) {
            // bulk trailing updates run at the LOWEST priority so that panel kernels win dispatch slots
            int lo_prio = 0, hi_prio = 0;
            STBA_HIP(hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio));
            STBA_HIP(hipStreamCreateWithPriority(&su, hipStreamNonBlocking, lo_prio));
            STBA_HIP(hipEventCreateWithFlags(&evU, hipEventDisableTiming));
        }
        while ((int)evP.size() < nblk) {
            hipEvent_t e1, e2;
            STBA_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
            STBA_HIP(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
            evP.push_back(e1); evN.push_back(e2);
        }
        hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), 0, st, A, lda, 0, n, flag_dev);
        hipLaunchKernelGGL(chol_trsm_kernel, dim3((nblk - 1) * (NB / TRSM_ROWS) + NB / TRSM_ROWS), dim3(256), trsm_lds, st, A, lda, 0,
                           (nblk - 1) * (NB / TRSM_ROWS), linv);
        STBA_HIP(hipEventRecord(evP[0], st));
        // look-ahead only pays while the bulk update is longer than the panel chain (measured on
        // MI355X: cross-stream hand-offs cost ~7-14 us each and the panel is ~65 us under contention)
        static const int LOOKAHEAD_MIN_MT = [] { const char* e = getenv("STBA_LOOKAHEAD_MIN_MT"); return e ? atoi(e) : 36; }();
        int b = 0;
        for (; b + 1 < nblk && (nblk - b - 1) >= LOOKAHEAD_MIN_MT; ++b) {
            const int k0 = b * NB;
            const int mt = nblk - b - 1;
            STBA_HIP(hipStreamWaitEvent(su, evP[b], 0));
            launch_syrk(A, lda, k0, 1, mt, su);
            STBA_HIP(hipEventRecord(evN[b], su));
            if (mt > 1)
                launch_syrk(A, lda, k0, 2, mt * (mt - 1) / 2, su);
            STBA_HIP(hipStreamWaitEvent(st, evN[b], 0));
            const int k1 = k0 + NB;
            hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), 0, st, A, lda, k1, n, flag_dev);
            hipLaunchKernelGGL(chol_trsm_kernel, dim3((mt - 1) * (NB / TRSM_ROWS) + NB / TRSM_ROWS), dim3(256), trsm_lds, st, A, lda, k1,
                               (mt - 1) * (NB / TRSM_ROWS), linv + (size_t)(b + 1) * NB * NB);
            STBA_HIP(hipEventRecord(evP[b + 1], st));
        }
        if (b > 0) {
            STBA_HIP(hipEventRecord(evU, su));
            STBA_HIP(hipStreamWaitEvent(st, evU, 0));
        }
        for (; b + 1 < nblk; ++b) {   // serial tail on the caller's stream
            const int k0 = b * NB;
            const int mt = nblk - b - 1;
            launch_syrk(A, lda, k0, 0, mt * (mt + 1) / 2, st);
            const int k1 = k0 + NB;
            hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), 0, st, A, lda, k1, n, flag_dev);
            hipLaunchKernelGGL(chol_trsm_kernel, dim3((mt - 1) * (NB / TRSM_ROWS) + NB / TRSM_ROWS), dim3(256), trsm_lds, st, A, lda, k1,
                               (mt - 1) * (NB / TRSM_ROWS), linv + (size_t)(b + 1) * NB * NB);
        }
    }
    STBA_TRY(mark((size_t)nblk * 4));
    for (int b = nblk - 1; b >= 0; --b) {
        const int k0 = b * NB;
        const int has_next = (b < nblk - 1) ? 1 : 0;
        const int grid = 1 + (has_next ? ((k0 + 1023) / 1024) * BWD_ROW_CHUNKS : 0);
        hipLaunchKernelGGL(chol_bwd_step_kernel, dim3(grid), dim3(1024), 0, st, A, lda, k0, has_next, linv + (size_t)b * NB * NB, x_dev);
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

int chol_factor_solve_dev(double* A, int lda, int n, double* x_dev, int* flag_dev, hipStream_t st) {
    return chol_run(A, lda, n, x_dev, flag_dev, st, nullptr);
}

int chol_factor_solve_profiled(double* A, int lda, int n, double* x_dev, int* flag_dev, hipStream_t st,
                               CholProfile* prof) {
    return chol_run(A, lda, n, x_dev, flag_dev, st, prof);
}

}  // namespace stba

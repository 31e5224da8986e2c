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
// optional phase timestamps for tools/exp/diag_timing.hip (compiled out of the library)
#ifdef STBA_DIAG_TS
__device__ long long g_diag_ts[4][16][6];
#define DIAG_TS(slot) do { if ((t & 63) == 0) g_diag_ts[t >> 6][2 * Jt + h][slot] = __builtin_readcyclecounter(); } while (0)
#else
#define DIAG_TS(slot) do { } while (0)
#endif

template <int Jt>
__device__ __forceinline__ void chol_diag_tilecol(double4v (&acc)[2][8], double (*P)[9], double (*Lp)[9], double* rd, int t, int lr,
                                                  int lc, const int (&Irow)[2], int k0, int n_real, int* flag) {
    // tile column Jt is a compile-time constant so that every accumulator index is static (a runtime
    // tile index makes the compiler spill the accumulators to scratch)
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        const int j0 = 16 * Jt + 8 * h;
        DIAG_TS(0);
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
        DIAG_TS(1);
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
            if (i == j0) {
#pragma unroll
                for (int c = 0; c < 8; ++c) rd[j0 + c] = y[c];      // 1 / L[c][c], for the inverse blocks
            }
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
        DIAG_TS(2);
        __syncthreads();
        DIAG_TS(3);
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
        DIAG_TS(4);
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
        DIAG_TS(5);
    }
}

__global__ __launch_bounds__(256) void chol_diag_kernel(double* __restrict__ A, int lda, int k0,
                                                         int n_real, int* __restrict__ flag,
                                                         double* __restrict__ dinv) {
    __shared__ double P[NB][9];
    __shared__ double Lp[NB][9];
    __shared__ double rd[NB];
    __shared__ double Tl[8][16][17];
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
    chol_diag_tilecol<0>(acc, P, Lp, rd, t, lr, lc, Irow, k0, n_real, flag);
    chol_diag_tilecol<1>(acc, P, Lp, rd, t, lr, lc, Irow, k0, n_real, flag);
    chol_diag_tilecol<2>(acc, P, Lp, rd, t, lr, lc, Irow, k0, n_real, flag);
    chol_diag_tilecol<3>(acc, P, Lp, rd, t, lr, lc, Irow, k0, n_real, flag);
    chol_diag_tilecol<4>(acc, P, Lp, rd, t, lr, lc, Irow, k0, n_real, flag);
    chol_diag_tilecol<5>(acc, P, Lp, rd, t, lr, lc, Irow, k0, n_real, flag);
    chol_diag_tilecol<6>(acc, P, Lp, rd, t, lr, lc, Irow, k0, n_real, flag);
    chol_diag_tilecol<7>(acc, P, Lp, rd, t, lr, lc, Irow, k0, n_real, flag);
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
    // inverses of the eight 16x16 diagonal tiles (the panel solve multiplies by them on the matrix
    // cores): thread (b, m) forward-substitutes column m of tile b's inverse
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int J = 0; J < 8; ++J)
            if (J == Irow[s]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Tl[J][lr + 4 * r][lc] = acc[s][J][r];
            }
    __syncthreads();
    if (t < NB) {
        const int b = t >> 4, m = t & 15;
        double x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            double sum = (k == m) ? 1.0 : 0.0;
#pragma unroll
            for (int q = 0; q < k; ++q) sum = fma(-Tl[b][k][q], x[q], sum);
            x[k] = (k >= m) ? sum * rd[16 * b + k] : 0.0;
            dinv[(b * 16 + k) * 16 + m] = x[k];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Panel solve X = A21 * L11^-T on the matrix cores, one wave per 16 panel rows, computed in the
// transposed form  Y = X^T = L11^-1 * A21^T  so that a finished 16x16 tile Y_I, sitting in the
// accumulator layout, IS the B operand of the next v_mfma_f64_16x16x4_f64 (k-step q <-> register q):
// no LDS, no shuffles, no barriers.
//   for I = 0..7:   Y_I  = Inv_II * W_I                      (4 MFMAs; Inv_II from chol_diag_kernel)
//                   W_J -= L_JI * Y_I   for J > I            (4 MFMAs per tile, independent chains)
// Row order inside a tile: accumulator register r of lane group g holds LOGICAL row 4g + r (physical
// MFMA row g + 4r), so a lane's four registers are four consecutive matrix columns of X: one 32 B
// load/store per lane per tile, and the A operands (L_JI, Inv_II) are one 32 B load per lane too.
// The last 8 workgroups run the same solve on the rows of the identity: X = I * L11^-T, the
// inverse transpose of the diagonal block, used by the backward substitution.
__global__ __launch_bounds__(64) void chol_trsm_kernel(double* __restrict__ A, int lda, int k0, int n_groups,
                                                       double* __restrict__ Xinv, const double* __restrict__ dinv) {
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    const int pm = 4 * (n & 3) + (n >> 2);          // logical tile row this lane feeds as an A operand
    const int grp = blockIdx.x;
    const bool ident = grp >= n_groups;
    const int e = grp - n_groups;
    const int nv = ident ? min(NB, (lda - 1) - k0) : NB;   // rows of the block that belong to the system
    double* rowp = ident ? Xinv + (size_t)(16 * e + n) * NB
                         : A + (size_t)(k0 + NB + 16 * grp + n) * lda + k0;
    const double* Lb = A + (size_t)k0 * lda + k0;
    double4v W[8];
#pragma unroll
    for (int J = 0; J < 8; ++J) {
        if (ident) {
#pragma unroll
            for (int r = 0; r < 4; ++r) W[J][r] = (16 * J + 4 * g + r == 16 * e + n) ? 1.0 : 0.0;
        } else {
            W[J] = *reinterpret_cast<const double4v*>(rowp + 16 * J + 4 * g);
        }
    }
    auto load_l = [&](int J, int I) -> double4v {
        const int row = 16 * J + pm;
        double4v v = *reinterpret_cast<const double4v*>(Lb + (size_t)row * lda + 16 * I + 4 * g);
        if (row >= nv) v = double4v{0.0, 0.0, 0.0, 0.0};
        return -v;
    };
    auto load_inv = [&](int J) -> double4v {
        double4v v = *reinterpret_cast<const double4v*>(dinv + (J * 16 + pm) * 16 + 4 * g);
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
    for (int J = 0; J < 8; ++J) *reinterpret_cast<double4v*>(rowp + 16 * J + 4 * g) = W[J];
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
    // per diagonal block: its inverse transpose (written by the panel solve, read by the backward pass)
    // followed by the inverses of its eight 16x16 diagonal tiles (written by the diagonal kernel, read
    // by the panel solve)
    constexpr size_t LINV_STRIDE = (size_t)NB * NB + 8 * 256;
    static thread_local double* linv = nullptr;
    static thread_local int linv_blocks = 0;
    if (linv_blocks < nblk) {
        if (linv) (void)hipFree(linv);
        linv = nullptr; linv_blocks = 0;
        STBA_HIP(hipMalloc(reinterpret_cast<void**>(&linv), (size_t)nblk * LINV_STRIDE * sizeof(double)));
        linv_blocks = nblk;
    }
    // panel of block b: diagonal kernel + panel solve of the (nblk - b - 1) * 8 row groups below it
    auto launch_panel_diag = [&](int b, hipStream_t s_) {
        double* li = linv + (size_t)b * LINV_STRIDE;
        hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), 0, s_, A, lda, b * NB, n, flag_dev, li + NB * NB);
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
    STBA_HIP(hipMemsetAsync(flag_dev, 0, sizeof(int), st));
    if (prof) {
        // serial schedule: one kernel class at a time, so the per-class event times are clean
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
            if (mt > 0) {
                const double m = std::max(0, n - (k0 + NB));
                prof->syrk_flops += m * (m + 1.0) * NB;
                prof->syrk_flops_padded += (double)(mt * (mt + 1) / 2) * 2.0 * NB * NB * NB;
                prof->syrk_launches += 1;
            }
        }
    } else {
        // look-ahead schedule on a partitioned chip.  The panel chain (diagonal block, panel solve,
        // update of the next panel's tile column) is latency-bound and needs few CUs; the bulk of the
        // trailing update is throughput-bound.  Two streams with DISJOINT CU masks
        // (hipExtStreamCreateWithCUMask) keep them from competing for the same CUs: without the
        // partition a 324-VGPR diagonal-block workgroup cannot start until a CU has drained ALL its
        // trailing-update workgroups, which serialises the two.  Dependencies:
        //   sp:  D(0) T(0) | U1(0) D(1) T(1) | [wait bulk(0)] U1(1) D(2) T(2) | ...
        //   su:             [wait T(0)] U2(0)  [wait T(1)] U2(1) ...
        // U1(b) = update of tile column b+1 by panel b; U2(b) = the columns right of it.  The only
        // cross-stream wait on the panel chain is for U2(b-1), which had a whole panel time to finish.
        static thread_local hipStream_t sp = nullptr, su = nullptr;
        static thread_local std::vector<hipEvent_t> evP, evU;
        static thread_local hipEvent_t evStart = nullptr, evEnd = nullptr;
        static const int PANEL_CUS = [] { const char* e = getenv("STBA_PANEL_CUS"); return e ? atoi(e) : 32; }();
        static const int SERIAL_MT = [] { const char* e = getenv("STBA_SERIAL_MT"); return e ? atoi(e) : 0; }();
        if (!sp) {
            int dev = 0;
            hipDeviceProp_t prop;
            STBA_HIP(hipGetDevice(&dev));
            STBA_HIP(hipGetDeviceProperties(&prop, dev));
            const int ncu = prop.multiProcessorCount;
            const int words = (ncu + 31) / 32;
            std::vector<uint32_t> mp((size_t)words, 0u), mu((size_t)words, 0u);
            for (int c = 0; c < ncu; ++c) {
                if (c < PANEL_CUS) mp[(size_t)c / 32] |= 1u << (c % 32);
                else mu[(size_t)c / 32] |= 1u << (c % 32);
            }
            if (PANEL_CUS <= 0 || PANEL_CUS >= ncu) {   // no partition: plain streams
                STBA_HIP(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
                STBA_HIP(hipStreamCreateWithFlags(&su, hipStreamNonBlocking));
            } else {
                STBA_HIP(hipExtStreamCreateWithCUMask(&sp, (uint32_t)words, mp.data()));
                STBA_HIP(hipExtStreamCreateWithCUMask(&su, (uint32_t)words, mu.data()));
            }
            STBA_HIP(hipEventCreateWithFlags(&evStart, hipEventDisableTiming));
            STBA_HIP(hipEventCreateWithFlags(&evEnd, hipEventDisableTiming));
        }
        while ((int)evP.size() < nblk) {
            hipEvent_t e1, e2;
            STBA_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
            STBA_HIP(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
            evP.push_back(e1); evU.push_back(e2);
        }
        STBA_HIP(hipEventRecord(evStart, st));
        STBA_HIP(hipStreamWaitEvent(sp, evStart, 0));
        launch_panel_diag(0, sp);
        launch_panel_trsm(0, sp);
        STBA_HIP(hipEventRecord(evP[0], sp));
        int b = 0, last_bulk = -1;
        for (; b + 1 < nblk && (nblk - b - 1) > SERIAL_MT; ++b) {
            const int k0 = b * NB;
            const int mt = nblk - b - 1;
            if (last_bulk >= 0) STBA_HIP(hipStreamWaitEvent(sp, evU[last_bulk], 0));
            launch_syrk(A, lda, k0, 1, mt, sp);
            if (mt > 1) {
                STBA_HIP(hipStreamWaitEvent(su, evP[b], 0));
                launch_syrk(A, lda, k0, 2, mt * (mt - 1) / 2, su);
                STBA_HIP(hipEventRecord(evU[b], su));
                last_bulk = b;
            }
            launch_panel_diag(b + 1, sp);
            launch_panel_trsm(b + 1, sp);
            STBA_HIP(hipEventRecord(evP[b + 1], sp));
        }
        if (last_bulk >= 0) STBA_HIP(hipStreamWaitEvent(sp, evU[last_bulk], 0));
        STBA_HIP(hipEventRecord(evEnd, sp));
        STBA_HIP(hipStreamWaitEvent(st, evEnd, 0));
        for (; b + 1 < nblk; ++b) {   // serial tail on the caller's stream (whole chip)
            const int k0 = b * NB;
            const int mt = nblk - b - 1;
            launch_syrk(A, lda, k0, 0, mt * (mt + 1) / 2, st);
            launch_panel_diag(b + 1, st);
            launch_panel_trsm(b + 1, st);
        }
    }
    STBA_TRY(mark((size_t)nblk * 4));
    for (int b = nblk - 1; b >= 0; --b) {
        const int k0 = b * NB;
        const int has_next = (b < nblk - 1) ? 1 : 0;
        const int grid = 1 + (has_next ? ((k0 + 1023) / 1024) * BWD_ROW_CHUNKS : 0);
        hipLaunchKernelGGL(chol_bwd_step_kernel, dim3(grid), dim3(1024), 0, st, A, lda, k0, has_next, linv + (size_t)b * LINV_STRIDE, x_dev);
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

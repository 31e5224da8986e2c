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
// diagonal block.  Thread (ty, tx) owns rows ty + 32a, columns tx + 32b (a, b < 4).  Columns are
// processed in four groups A0 = j >> 5 so that every register index is a compile-time constant
// and whole sub-blocks outside the active trailing part are skipped.
template <int A0>
__device__ __forceinline__ void chol_diag_steps(double (&acc)[4][4], double (*colbuf)[NB], int tx, int ty,
                                                int k0, int n_real, int* flag) {
    for (int jj = 0; jj < 32; ++jj) {
        const int j = 32 * A0 + jj;
        double* cb = colbuf[j & 1];
        if (tx == jj) {
#pragma unroll
            for (int a = A0; a < 4; ++a) cb[ty + 32 * a] = acc[a][A0];
        }
        __syncthreads();
        const double d = cb[j];
        double ljj, inv;
        if (d > 0.0) {
            // division-free pivot: l = d * rsqrt(d) plus one FMA Newton correction (~1 ulp); the
            // column is scaled by rsqrt(d) itself, keeping the dependent chain short
            inv = rsqrt(d);
            ljj = d * inv;
            ljj = fma(fma(-ljj, ljj, d), 0.5 * inv, ljj);
        } else {
            ljj = 1.0; inv = 1.0;
            if (threadIdx.x == 0 && (k0 + j) < n_real) atomicCAS(flag, 0, k0 + j + 1);
        }
        double li[4], lc[4];
#pragma unroll
        for (int a = A0; a < 4; ++a) li[a] = cb[ty + 32 * a] * inv;
#pragma unroll
        for (int b = A0; b < 4; ++b) lc[b] = cb[tx + 32 * b] * inv;
#pragma unroll
        for (int a = A0; a < 4; ++a)
#pragma unroll
            for (int b = A0; b <= a; ++b) {
                bool on = true;
                if (b == A0) on = on && (tx > jj);     // column strictly right of j
                if (b == a) on = on && (tx <= ty);     // lower triangle of a diagonal sub-block
                if (on) acc[a][b] -= li[a] * lc[b];
            }
        if (tx == jj) {
#pragma unroll
            for (int a = A0; a < 4; ++a) {
                const int i = ty + 32 * a;
                acc[a][A0] = (i > j) ? li[a] : ((i == j) ? ljj : acc[a][A0]);
            }
        }
    }
}

__global__ __launch_bounds__(1024) void chol_diag_kernel(double* __restrict__ A, int lda, int k0,
                                                         int n_real, int* __restrict__ flag) {
    __shared__ double colbuf[2][NB];
    const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = ty + 32 * a, c = tx + 32 * b;
            acc[a][b] = (c <= i) ? A[(size_t)(k0 + i) * lda + k0 + c] : 0.0;
        }
    chol_diag_steps<0>(acc, colbuf, tx, ty, k0, n_real, flag);
    chol_diag_steps<1>(acc, colbuf, tx, ty, k0, n_real, flag);
    chol_diag_steps<2>(acc, colbuf, tx, ty, k0, n_real, flag);
    chol_diag_steps<3>(acc, colbuf, tx, ty, k0, n_real, flag);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = ty + 32 * a, c = tx + 32 * b;
            if (c <= i) A[(size_t)(k0 + i) * lda + k0 + c] = acc[a][b];
        }
}

// ------------------------------------------------------------------------------------------
// X = A21 * L11^-T for TRSM_ROWS rows per workgroup, in place.  256 threads: thread (r, cg) owns
// row r = t >> 3 and columns cg + 8m (m < 16).  Column groups M = j >> 3 are unrolled so that
// the owner's register index is static.
template <int M>
__device__ __forceinline__ void chol_trsm_steps(double (&acc)[16], const double* __restrict__ Ld,
                                                const double* __restrict__ invd, double (*xbuf)[TRSM_ROWS],
                                                int r, int cg) {
#pragma unroll 1
    for (int jj = 0; jj < 8; ++jj) {
        const int j = 8 * M + jj;
        double* xb = xbuf[j & 1];
        if (cg == jj) {
            const double x = acc[M] * invd[j];
            acc[M] = x;
            xb[r] = x;
        }
        __syncthreads();
        const double xi = xb[r];
        const double* lcol = Ld + j;   // L[c][j] at Ld[c*(NB+1) + j]
        if (cg > jj) acc[M] -= xi * lcol[(cg + 8 * M) * (NB + 1)];
#pragma unroll
        for (int m = M + 1; m < 16; ++m) acc[m] -= xi * lcol[(cg + 8 * m) * (NB + 1)];
    }
}

__global__ __launch_bounds__(256) void chol_trsm_kernel(double* __restrict__ A, int lda, int k0) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Ld = smem;                          // [128][129]
    double* invd = smem + NB * (NB + 1);        // [128]
    double(*xbuf)[TRSM_ROWS] = reinterpret_cast<double(*)[TRSM_ROWS]>(invd + NB);   // [2][32]
    const int t = threadIdx.x, r = t >> 3, cg = t & 7;
    const int row = k0 + NB + blockIdx.x * TRSM_ROWS + r;
    for (int idx = t; idx < NB * NB; idx += 256) {
        const int i = idx >> 7, c = idx & 127;
        Ld[i * (NB + 1) + c] = (c <= i) ? A[(size_t)(k0 + i) * lda + k0 + c] : 0.0;
    }
    if (t < NB) invd[t] = 1.0 / A[(size_t)(k0 + t) * lda + k0 + t];
    double acc[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) acc[m] = A[(size_t)row * lda + k0 + cg + 8 * m];
    __syncthreads();
    chol_trsm_steps<0>(acc, Ld, invd, xbuf, r, cg);   chol_trsm_steps<1>(acc, Ld, invd, xbuf, r, cg);
    chol_trsm_steps<2>(acc, Ld, invd, xbuf, r, cg);   chol_trsm_steps<3>(acc, Ld, invd, xbuf, r, cg);
    chol_trsm_steps<4>(acc, Ld, invd, xbuf, r, cg);   chol_trsm_steps<5>(acc, Ld, invd, xbuf, r, cg);
    chol_trsm_steps<6>(acc, Ld, invd, xbuf, r, cg);   chol_trsm_steps<7>(acc, Ld, invd, xbuf, r, cg);
    chol_trsm_steps<8>(acc, Ld, invd, xbuf, r, cg);   chol_trsm_steps<9>(acc, Ld, invd, xbuf, r, cg);
    chol_trsm_steps<10>(acc, Ld, invd, xbuf, r, cg);  chol_trsm_steps<11>(acc, Ld, invd, xbuf, r, cg);
    chol_trsm_steps<12>(acc, Ld, invd, xbuf, r, cg);  chol_trsm_steps<13>(acc, Ld, invd, xbuf, r, cg);
    chol_trsm_steps<14>(acc, Ld, invd, xbuf, r, cg);  chol_trsm_steps<15>(acc, Ld, invd, xbuf, r, cg);
#pragma unroll
    for (int m = 0; m < 16; ++m) A[(size_t)row * lda + k0 + cg + 8 * m] = acc[m];
}

// ------------------------------------------------------------------------------------------
// Trailing update C(i,j) -= P_i P_j^T over the lower-triangle 128x128 tiles, P = panel columns
// [k0, k0+128).  256 threads = 4 waves in a 2x2 grid, each wave 64x64 = 4x4 MFMA tiles.
__global__ __launch_bounds__(256) void chol_syrk_kernel(double* __restrict__ A, int lda, int k0) {
    __shared__ __attribute__((aligned(16))) double sA[2][2048];
    __shared__ __attribute__((aligned(16))) double sB[2][2048];
    const int id = blockIdx.x;
    int ti = (int)((sqrt(8.0 * (double)id + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > id) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= id) ++ti;
    const int tj = id - ti * (ti + 1) / 2;
    const int r0 = k0 + NB;
    const int row_i = r0 + ti * NB, row_j = r0 + tj * NB;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = w >> 1, wc = w & 1;

    double4v acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (double4v){0.0, 0.0, 0.0, 0.0};

    // staging map: pass p, half h -> row = (lane&15) + 16*(w + 4p), k = 2*((lane>>4) + 4h)
    double2 ga[2][2], gb[2][2];
    const int lrow = lane & 15, lkp = lane >> 4;
    auto gload = [&](int kc) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = lrow + 16 * (w + 4 * p);
                const int k = 2 * (lkp + 4 * h);
                ga[p][h] = *reinterpret_cast<const double2*>(&A[(size_t)(row_i + row) * lda + k0 + kc * 16 + k]);
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
                const int pos0 = (((k >> 2) * 8 + rb) << 6) + ((k & 3) << 4) + lrow;
                sA[buf][pos0] = ga[p][h].x;
                sA[buf][pos0 + 16] = ga[p][h].y;          // k+1: (k&3) is even so +1 -> +16
                sB[buf][pos0] = gb[p][h].x;
                sB[buf][pos0 + 16] = gb[p][h].y;
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
            double a[4], b[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = sA[buf][((kq * 8 + wr * 4 + m) << 6) + lane];
#pragma unroll
            for (int n = 0; n < 4; ++n) b[n] = sB[buf][((kq * 8 + wc * 4 + n) << 6) + lane];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        if (kc + 1 < KC) lstore(buf ^ 1);
        __syncthreads();
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*reg
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row_i + wr * 64 + m * 16 + (lane >> 4) + 4 * r;
                const int col = row_j + wc * 64 + n * 16 + (lane & 15);
                double* p = &A[(size_t)row * lda + col];
                *p -= acc[m][n][r];
            }
}

// ------------------------------------------------------------------------------------------
// backward substitution, block b: x_b = L_bb^-T y_b   (y lives in row lda-1).
// The block is staged in LDS by all 1024 threads, then ONE wave runs the 128 dependent steps with
// a shuffle broadcast per step (no barriers): lane l owns unknowns l and l + 64.
__global__ __launch_bounds__(1024) void chol_bwd_diag_kernel(double* __restrict__ A, int lda, int k0,
                                                             double* __restrict__ x) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Ld = smem;                 // [128][129]
    const int t = threadIdx.x;
    const int nv = min(NB, (lda - 1) - k0);   // rows of this block that belong to the system
    for (int idx = t; idx < NB * NB; idx += 1024) {
        const int i = idx >> 7, c = idx & 127;
        Ld[i * (NB + 1) + c] = (c <= i && i < nv) ? A[(size_t)(k0 + i) * lda + k0 + c] : ((c == i) ? 1.0 : 0.0);
    }
    __syncthreads();
    if (t >= 64) return;
    const int l = t;
    double y0 = (l < nv) ? A[(size_t)(lda - 1) * lda + k0 + l] : 0.0;
    double y1 = (l + 64 < nv) ? A[(size_t)(lda - 1) * lda + k0 + l + 64] : 0.0;
    const double id0 = 1.0 / Ld[l * (NB + 1) + l], id1 = 1.0 / Ld[(l + 64) * (NB + 1) + l + 64];
    for (int j = NB - 1; j >= 64; --j) {
        const double xj = __shfl(y1 * id1, j - 64, 64);
        const double* rowj = Ld + j * (NB + 1);
        if (l + 64 == j) y1 = xj;
        else if (l + 64 < j) y1 -= rowj[l + 64] * xj;
        y0 -= rowj[l] * xj;
    }
    for (int j = 63; j >= 0; --j) {
        const double xj = __shfl(y0 * id0, j, 64);
        if (l == j) y0 = xj;
        else if (l < j) y0 -= Ld[j * (NB + 1) + l] * xj;
    }
    x[k0 + l] = (l < nv) ? y0 : 0.0;
    x[k0 + l + 64] = (l + 64 < nv) ? y1 : 0.0;
}

// y[0:k0] -= L[k0:k0+nv, 0:k0]^T x_b
__global__ __launch_bounds__(256) void chol_bwd_update_kernel(double* __restrict__ A, int lda, int k0,
                                                              const double* __restrict__ x) {
    __shared__ double xs[NB];
    const int nv = min(NB, (lda - 1) - k0);
    if (threadIdx.x < NB) xs[threadIdx.x] = (threadIdx.x < nv) ? x[k0 + threadIdx.x] : 0.0;
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= k0) return;
    double s = 0.0;
    const double* col = A + (size_t)k0 * lda + c;
#pragma unroll 8
    for (int j = 0; j < nv; ++j) s += col[(size_t)j * lda] * xs[j];
    A[(size_t)(lda - 1) * lda + c] -= s;
}

// ------------------------------------------------------------------------------------------
static int chol_run(double* A, int lda, int n, double* x_dev, int* flag_dev, hipStream_t st, CholProfile* prof) {
    if (lda % NB != 0 || lda < n + 1) return fail(STBA_ERR_INVALID_ARGUMENT, "chol: bad padded dimension");
    const int nblk = lda / NB;
    const size_t trsm_lds = sizeof(double) * (NB * (NB + 1) + NB + 2 * TRSM_ROWS);
    const size_t bwd_lds = sizeof(double) * (NB * (NB + 1));
    static bool attr_set = false;
    if (!attr_set) {
        STBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_trsm_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)trsm_lds));
        STBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_bwd_diag_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_lds));
        attr_set = true;
    }
    std::vector<hipEvent_t> ev;
    if (prof) {
        ev.resize((size_t)nblk * 4 + 2);
        for (auto& e : ev) STBA_HIP(hipEventCreate(&e));
        memset(prof, 0, sizeof *prof);
    }
    auto mark = [&](size_t k) -> int { if (prof) STBA_HIP(hipEventRecord(ev[k], st)); return STBA_OK; };
    STBA_HIP(hipMemsetAsync(flag_dev, 0, sizeof(int), st));
    for (int b = 0; b < nblk; ++b) {
        const int k0 = b * NB;
        const int mt = nblk - b - 1;
        STBA_TRY(mark(4 * (size_t)b + 0));
        hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(1024), 0, st, A, lda, k0, n, flag_dev);
        STBA_TRY(mark(4 * (size_t)b + 1));
        if (mt > 0)
            hipLaunchKernelGGL(chol_trsm_kernel, dim3(mt * (NB / TRSM_ROWS)), dim3(256), trsm_lds, st, A, lda, k0);
        STBA_TRY(mark(4 * (size_t)b + 2));
        if (mt > 0) hipLaunchKernelGGL(chol_syrk_kernel, dim3(mt * (mt + 1) / 2), dim3(256), 0, st, A, lda, k0);
        STBA_TRY(mark(4 * (size_t)b + 3));
        if (prof && mt > 0) {
            const double m = std::max(0, n - (k0 + NB));
            prof->syrk_flops += m * (m + 1.0) * NB;
            prof->syrk_flops_padded += (double)(mt * (mt + 1) / 2) * 2.0 * NB * NB * NB;
            prof->syrk_launches += 1;
        }
    }
    STBA_TRY(mark((size_t)nblk * 4));
    for (int b = nblk - 1; b >= 0; --b) {
        const int k0 = b * NB;
        hipLaunchKernelGGL(chol_bwd_diag_kernel, dim3(1), dim3(1024), bwd_lds, st, A, lda, k0, x_dev);
        if (k0 > 0)
            hipLaunchKernelGGL(chol_bwd_update_kernel, dim3((k0 + 255) / 256), dim3(256), 0, st, A, lda, k0, x_dev);
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

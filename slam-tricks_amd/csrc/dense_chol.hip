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
//                      (4x4 per thread, 32-cyclic distribution), one LDS column broadcast and one
//                      barrier per column.
//   chol_trsm_kernel   one workgroup per 128 rows of the panel; L11 staged in LDS (129-padded),
//                      panel rows in registers, same column-broadcast scheme.
//   chol_syrk_kernel   128x128 output tiles of the trailing lower triangle, 4 waves x (4x4)
//                      16x16 MFMA tiles, K = 128 streamed through double-buffered LDS in
//                      fragment order (conflict-free ds_read_b64 / ds_write_b64).
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
    for (int j = 0; j < NB; ++j) {
        const int bj = j >> 5, txj = j & 31;
        double* cb = colbuf[j & 1];
        if (tx == txj) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                double v = acc[a][0];
#pragma unroll
                for (int b = 1; b < 4; ++b) v = (b == bj) ? acc[a][b] : v;
                cb[ty + 32 * a] = v;
            }
        }
        __syncthreads();
        const double d = cb[j];
        double ljj;
        if (d > 0.0) {
            ljj = sqrt(d);
        } else {
            ljj = 1.0;
            if (t == 0 && (k0 + j) < n_real) atomicCAS(flag, 0, k0 + j + 1);
        }
        const double inv = 1.0 / ljj;
        double li[4], lc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) li[a] = cb[ty + 32 * a] * inv;
#pragma unroll
        for (int b = 0; b < 4; ++b) lc[b] = cb[tx + 32 * b] * inv;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int i = ty + 32 * a, c = tx + 32 * b;
                if (c > j && c <= i) acc[a][b] -= li[a] * lc[b];
                if (c == j) acc[a][b] = (i > j) ? li[a] : ((i == j) ? ljj : acc[a][b]);
            }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = ty + 32 * a, c = tx + 32 * b;
            if (c <= i) A[(size_t)(k0 + i) * lda + k0 + c] = acc[a][b];
        }
}

// ------------------------------------------------------------------------------------------
// X = A21 * L11^-T for 128 rows per workgroup (rows r0 + 128*blockIdx.x ...), in place.
__global__ __launch_bounds__(1024) void chol_trsm_kernel(double* __restrict__ A, int lda, int k0) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Ld = smem;                      // [128][129]
    double* colbuf = smem + NB * (NB + 1);  // [2][128]
    const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
    const int r0 = k0 + NB + blockIdx.x * NB;
    for (int idx = t; idx < NB * NB; idx += 1024) {
        const int i = idx >> 7, c = idx & 127;
        Ld[i * (NB + 1) + c] = (c <= i) ? A[(size_t)(k0 + i) * lda + k0 + c] : 0.0;
    }
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
            acc[a][b] = A[(size_t)(r0 + ty + 32 * a) * lda + k0 + tx + 32 * b];
    __syncthreads();
    for (int j = 0; j < NB; ++j) {
        const int bj = j >> 5, txj = j & 31;
        double* cb = colbuf + (j & 1) * NB;
        if (tx == txj) {
            const double inv = 1.0 / Ld[j * (NB + 1) + j];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                double v = acc[a][0];
#pragma unroll
                for (int b = 1; b < 4; ++b) v = (b == bj) ? acc[a][b] : v;
                v *= inv;
                cb[ty + 32 * a] = v;
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = (b == bj) ? v : acc[a][b];
            }
        }
        __syncthreads();
        double xi[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) xi[a] = cb[ty + 32 * a];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int c = tx + 32 * b;
            if (c > j) {
                const double l = Ld[c * (NB + 1) + j];
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a][b] -= xi[a] * l;
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
            A[(size_t)(r0 + ty + 32 * a) * lda + k0 + tx + 32 * b] = acc[a][b];
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
// backward substitution, block b: x_b = L_bb^-T y_b   (y lives in row lda-1)
__global__ __launch_bounds__(1024) void chol_bwd_diag_kernel(double* __restrict__ A, int lda, int k0,
                                                             double* __restrict__ x) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Ld = smem;                 // [128][129]
    double* xs = smem + NB * (NB + 1); // [128]
    const int t = threadIdx.x;
    const int nv = min(NB, (lda - 1) - k0);   // rows of this block that belong to the system
    for (int idx = t; idx < NB * NB; idx += 1024) {
        const int i = idx >> 7, c = idx & 127;
        Ld[i * (NB + 1) + c] = (c <= i && i < nv) ? A[(size_t)(k0 + i) * lda + k0 + c] : 0.0;
    }
    double y = 0.0;
    if (t < nv) y = A[(size_t)(lda - 1) * lda + k0 + t];
    __syncthreads();
    for (int j = nv - 1; j >= 0; --j) {
        if (t == j) xs[j] = y / Ld[j * (NB + 1) + j];
        __syncthreads();
        if (t < j) y -= Ld[j * (NB + 1) + t] * xs[j];
    }
    __syncthreads();
    if (t < NB) x[k0 + t] = (t < nv) ? xs[t] : 0.0;
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

// same as chol_factor_solve_dev with hipEvent pairs around every kernel class (profiling only)
int chol_factor_solve_profiled(double* A, int lda, int n, double* x_dev, int* flag_dev, hipStream_t st,
                               CholProfile* prof) {
    if (lda % NB != 0 || lda < n + 1) return fail(STBA_ERR_INVALID_ARGUMENT, "chol: bad padded dimension");
    // precondition: chol_factor_solve_dev ran once before (it sets the dynamic-LDS attributes)
    STBA_HIP(hipMemsetAsync(flag_dev, 0, sizeof(int), st));
    const int nblk = lda / NB;
    const size_t trsm_lds = sizeof(double) * (NB * (NB + 1) + 2 * NB);
    const size_t bwd_lds = sizeof(double) * (NB * (NB + 1) + NB);
    std::vector<hipEvent_t> ev((size_t)nblk * 4 + 2);
    for (auto& e : ev) STBA_HIP(hipEventCreate(&e));
    memset(prof, 0, sizeof *prof);
    for (int b = 0; b < nblk; ++b) {
        const int k0 = b * NB;
        const int mt = nblk - b - 1;
        STBA_HIP(hipEventRecord(ev[4 * b + 0], st));
        hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(1024), 0, st, A, lda, k0, n, flag_dev);
        STBA_HIP(hipEventRecord(ev[4 * b + 1], st));
        if (mt > 0) hipLaunchKernelGGL(chol_trsm_kernel, dim3(mt), dim3(1024), trsm_lds, st, A, lda, k0);
        STBA_HIP(hipEventRecord(ev[4 * b + 2], st));
        if (mt > 0) hipLaunchKernelGGL(chol_syrk_kernel, dim3(mt * (mt + 1) / 2), dim3(256), 0, st, A, lda, k0);
        STBA_HIP(hipEventRecord(ev[4 * b + 3], st));
        if (mt > 0) {
            const double m = std::max(0, n - (k0 + NB));
            prof->syrk_flops += m * (m + 1.0) * NB;
            prof->syrk_flops_padded += (double)(mt * (mt + 1) / 2) * 2.0 * NB * NB * NB;
            prof->syrk_launches += 1;
        }
    }
    STBA_HIP(hipEventRecord(ev[(size_t)nblk * 4], st));
    for (int b = nblk - 1; b >= 0; --b) {
        const int k0 = b * NB;
        hipLaunchKernelGGL(chol_bwd_diag_kernel, dim3(1), dim3(1024), bwd_lds, st, A, lda, k0, x_dev);
        if (k0 > 0)
            hipLaunchKernelGGL(chol_bwd_update_kernel, dim3((k0 + 255) / 256), dim3(256), 0, st, A, lda, k0, x_dev);
    }
    STBA_HIP(hipEventRecord(ev[(size_t)nblk * 4 + 1], st));
    STBA_HIP(hipStreamSynchronize(st));
    float ms = 0.f;
    for (int b = 0; b < nblk; ++b) {
        (void)hipEventElapsedTime(&ms, ev[4 * b + 0], ev[4 * b + 1]); prof->ms_diag += ms;
        (void)hipEventElapsedTime(&ms, ev[4 * b + 1], ev[4 * b + 2]); prof->ms_trsm += ms;
        (void)hipEventElapsedTime(&ms, ev[4 * b + 2], ev[4 * b + 3]); prof->ms_syrk += ms;
    }
    (void)hipEventElapsedTime(&ms, ev[(size_t)nblk * 4], ev[(size_t)nblk * 4 + 1]); prof->ms_bwd = ms;
    for (auto& e : ev) (void)hipEventDestroy(e);
    return STBA_OK;
}

int chol_factor_solve_dev(double* A, int lda, int n, double* x_dev, int* flag_dev, hipStream_t st) {
    if (lda % NB != 0 || lda < n + 1) return fail(STBA_ERR_INVALID_ARGUMENT, "chol: bad padded dimension");
    const int nblk = lda / NB;
    const size_t trsm_lds = sizeof(double) * (NB * (NB + 1) + 2 * NB);
    const size_t bwd_lds = sizeof(double) * (NB * (NB + 1) + NB);
    static bool attr_set = false;
    if (!attr_set) {
        STBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_trsm_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)trsm_lds));
        STBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_bwd_diag_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_lds));
        attr_set = true;
    }
    STBA_HIP(hipMemsetAsync(flag_dev, 0, sizeof(int), st));
    for (int b = 0; b < nblk; ++b) {
        const int k0 = b * NB;
        hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(1024), 0, st, A, lda, k0, n, flag_dev);
        const int mt = nblk - b - 1;
        if (mt > 0) {
            hipLaunchKernelGGL(chol_trsm_kernel, dim3(mt), dim3(1024), trsm_lds, st, A, lda, k0);
            hipLaunchKernelGGL(chol_syrk_kernel, dim3(mt * (mt + 1) / 2), dim3(256), 0, st, A, lda, k0);
        }
    }
    for (int b = nblk - 1; b >= 0; --b) {
        const int k0 = b * NB;
        hipLaunchKernelGGL(chol_bwd_diag_kernel, dim3(1), dim3(1024), bwd_lds, st, A, lda, k0, x_dev);
        if (k0 > 0)
            hipLaunchKernelGGL(chol_bwd_update_kernel, dim3((k0 + 255) / 256), dim3(256), 0, st, A, lda, k0, x_dev);
    }
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

}  // namespace stba

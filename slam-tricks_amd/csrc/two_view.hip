// two_view.hip -- the two-view initialiser, the step BEFORE the bundle-adjustment path:
// from n pixel correspondences and K to the relative pose and the landmarks that seed the BA of config C2.
// Follows st22-two-view/src/src/two_view_geometry.cpp:18-126:
//   ComputeFunctionMatrix (:18-41)   null vector of the n x 9 system x1^T F x2 = 0 (no normalisation)
//   DecomposeFMat (:43-81)           E = K^T F K = U S V^T; t = +-u3; R = U W V^T | U W^T V^T (det fixed);
//                                    the ONE of the four hypotheses that puts EVERY point in front of both
//                                    cameras wins, otherwise the call fails
//   Triangulate (:105-126)           DLT, null vector of the 6 x 4 system [hat(x1) P1; hat(x2) P2]
// Device work (n up to 10^5..10^6): the n x 9 system is reduced to its 9 x 9 triangular factor R by
// Givens rotations (each lane folds its rows into a private R in registers, then a tree over LDS):
// A^T A = R^T R, so R has the singular values and right singular vectors of A without squaring the
// condition number; the cheirality test triangulates every pair under all four hypotheses and counts the
// failures; the final triangulation writes the landmarks.  The 9x9 / 3x3 decompositions are host work.
#include <cstring>
#include <vector>

#include "common.hpp"
#include "small_linalg.hpp"

namespace stba {
namespace {

constexpr int QR_THREADS = 128;

// fold the row a (9 entries) into the packed upper-triangular R (row k holds R[k][k..8])
__host__ __device__ inline void givens_fold(double (&R)[45], double (&a)[9]) {
    int idx = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const double ak = a[k];
        if (ak != 0.0) {
            const double d = R[idx];
            const double r = sqrt(d * d + ak * ak);
            const double c = d / r, s = ak / r;
            R[idx] = r;
#pragma unroll
            for (int j = k + 1; j < 9; ++j) {
                const double x = R[idx + j - k], y = a[j];
                R[idx + j - k] = c * x + s * y;
                a[j] = c * y - s * x;
            }
        }
        idx += 9 - k;
    }
}

__global__ __launch_bounds__(QR_THREADS) void tv_qr_kernel(int n, const double* __restrict__ f1, const double* __restrict__ f2,
                                                           double* __restrict__ Rout) {
    __shared__ double sh[QR_THREADS][45];
    double R[45];
#pragma unroll
    for (int k = 0; k < 45; ++k) R[k] = 0.0;
    for (int i = blockIdx.x * QR_THREADS + threadIdx.x; i < n; i += gridDim.x * QR_THREADS) {
        const double u1 = f1[2 * i], v1 = f1[2 * i + 1], u2 = f2[2 * i], v2 = f2[2 * i + 1];
        double a[9] = {u1 * u2, u1 * v2, u1, v1 * u2, v1 * v2, v1, u2, v2, 1.0};      // two_view_geometry.cpp:24-32
        givens_fold(R, a);
    }
    const int t = threadIdx.x;
    for (int off = QR_THREADS / 2; off >= 1; off >>= 1) {
        if (t >= off && t < 2 * off) {
#pragma unroll
            for (int k = 0; k < 45; ++k) sh[t][k] = R[k];
        }
        __syncthreads();
        if (t < off) {
            int idx = 0;
#pragma unroll
            for (int rr = 0; rr < 9; ++rr) {
                double a[9];
#pragma unroll
                for (int j = 0; j < 9; ++j) a[j] = (j >= rr) ? sh[t + off][idx + j - rr] : 0.0;
                givens_fold(R, a);
                idx += 9 - rr;
            }
        }
        __syncthreads();
    }
    if (t == 0) {
#pragma unroll
        for (int k = 0; k < 45; ++k) Rout[(size_t)blockIdx.x * 45 + k] = R[k];
    }
}

// DLT triangulation of one correspondence (two_view_geometry.cpp:105-126): the right singular vector of the
// smallest singular value of the 6x4 matrix, by one-sided Jacobi in registers
__device__ inline void dlt_triangulate(double u1, double v1, double u2, double v2, const double* P1, const double* P2,
                                       double (&X)[3]) {
    double A[6][4], V[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        A[0][c] = -P1[4 + c] + v1 * P1[8 + c];
        A[1][c] = P1[c] - u1 * P1[8 + c];
        A[2][c] = -v1 * P1[c] + u1 * P1[4 + c];
        A[3][c] = -P2[4 + c] + v2 * P2[8 + c];
        A[4][c] = P2[c] - u2 * P2[8 + c];
        A[5][c] = -v2 * P2[c] + u2 * P2[4 + c];
#pragma unroll
        for (int r = 0; r < 4; ++r) V[r][c] = (r == c) ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                double app = 0, aqq = 0, apq = 0;
#pragma unroll
                for (int r = 0; r < 6; ++r) { app += A[r][p] * A[r][p]; aqq += A[r][q] * A[r][q]; apq += A[r][p] * A[r][q]; }
                if (apq != 0.0) {
                    off = fmax(off, fabs(apq) / sqrt(fmax(app * aqq, 1e-300)));
                    const double zeta = (aqq - app) / (2.0 * apq);
                    const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double c = 1.0 / sqrt(1.0 + tt * tt), s = c * tt;
#pragma unroll
                    for (int r = 0; r < 6; ++r) { const double x = A[r][p], y = A[r][q]; A[r][p] = c * x - s * y; A[r][q] = s * x + c * y; }
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const double x = V[r][p], y = V[r][q]; V[r][p] = c * x - s * y; V[r][q] = s * x + c * y; }
                }
            }
        if (off < 1e-15) break;
    }
    int best = 0;
    double bn = 1e300;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double nn = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) nn += A[r][c] * A[r][c];
        if (nn < bn) { bn = nn; best = c; }
    }
    double h[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = (best == 0) ? V[r][0] : (best == 1) ? V[r][1] : (best == 2) ? V[r][2] : V[r][3];
    X[0] = h[0] / h[3]; X[1] = h[1] / h[3]; X[2] = h[2] / h[3];
}

struct TvHyp { double P1[12]; double P2[4][12]; double Rt[4][9]; double t[4][3]; };   // Rt = R^T (frame 1 -> frame 2)

__global__ __launch_bounds__(256) void tv_cheirality_kernel(int n, const double* __restrict__ f1, const double* __restrict__ f2,
                                                            TvHyp h, int* __restrict__ fails) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double u1 = f1[2 * i], v1 = f1[2 * i + 1], u2 = f2[2 * i], v2 = f2[2 * i + 1];
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        double X[3];
        dlt_triangulate(u1, v1, u2, v2, h.P1, h.P2[k], X);
        bool ok = X[2] > 0.0;                                   // two_view_geometry.cpp:91
        const double d0 = X[0] - h.t[k][0], d1 = X[1] - h.t[k][1], d2 = X[2] - h.t[k][2];
        const double z2 = h.Rt[k][6] * d0 + h.Rt[k][7] * d1 + h.Rt[k][8] * d2;
        ok = ok && (z2 > 0.0);                                  // :96
        if (!ok) atomicAdd(&fails[k], 1);
    }
}

__global__ __launch_bounds__(256) void tv_triangulate_kernel(int n, const double* __restrict__ f1, const double* __restrict__ f2,
                                                             TvHyp h, int k, double* __restrict__ pts) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double X[3];
    dlt_triangulate(f1[2 * i], f1[2 * i + 1], f2[2 * i], f2[2 * i + 1], h.P1, h.P2[k], X);
    pts[3 * i] = X[0]; pts[3 * i + 1] = X[1]; pts[3 * i + 2] = X[2];
}

void mat3_mul(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
void mat3_t(const double* A, double* T) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) T[i * 3 + j] = A[j * 3 + i];
}
double det3(const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}
// P = K [R^T | -R^T t]   (camera with pose (R, t) in frame 1; two_view_geometry.cpp:108-116)
void projection(const double* K, const double* R, const double* t, double* P) {
    double Rt[9], E[12];
    mat3_t(R, Rt);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) E[i * 4 + j] = Rt[i * 3 + j];
        E[i * 4 + 3] = -(Rt[i * 3] * t[0] + Rt[i * 3 + 1] * t[1] + Rt[i * 3 + 2] * t[2]);
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) P[i * 4 + j] = K[i * 3] * E[j] + K[i * 3 + 1] * E[4 + j] + K[i * 3 + 2] * E[8 + j];
}

}  // namespace
}  // namespace stba

using namespace stba;

extern "C" int stba_two_view_init(int n, const double* f1, const double* f2, const double* K, double* F_out, double* R_out,
                                  double* t_out, double* pts_out, int* fails_out, void* hip_stream) {
    if (n < 8 || !f1 || !f2 || !K || !R_out || !t_out) return fail(STBA_ERR_INVALID_ARGUMENT, "two_view_init: bad argument");
    STBA_TRY(require_device());
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    double *d1 = nullptr, *d2 = nullptr, *dR = nullptr, *dP = nullptr;
    int* dF = nullptr;
    auto cleanup = [&]() { (void)hipFree(d1); (void)hipFree(d2); (void)hipFree(dR); (void)hipFree(dP); (void)hipFree(dF); };
    const int qr_blocks = std::max(1, std::min(64, (n + QR_THREADS * 4 - 1) / (QR_THREADS * 4)));
#define TV_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { cleanup(); return fail(STBA_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } } while (0)
    TV_HIP(hipMalloc(reinterpret_cast<void**>(&d1), (size_t)n * 2 * sizeof(double)));
    TV_HIP(hipMalloc(reinterpret_cast<void**>(&d2), (size_t)n * 2 * sizeof(double)));
    TV_HIP(hipMalloc(reinterpret_cast<void**>(&dR), (size_t)qr_blocks * 45 * sizeof(double)));
    TV_HIP(hipMalloc(reinterpret_cast<void**>(&dP), (size_t)n * 3 * sizeof(double)));
    TV_HIP(hipMalloc(reinterpret_cast<void**>(&dF), 4 * sizeof(int)));
    TV_HIP(hipMemcpyAsync(d1, f1, (size_t)n * 2 * sizeof(double), hipMemcpyHostToDevice, st));
    TV_HIP(hipMemcpyAsync(d2, f2, (size_t)n * 2 * sizeof(double), hipMemcpyHostToDevice, st));
    // 1. fundamental matrix: triangular factor on the device, 9x9 SVD on the host
    hipLaunchKernelGGL(tv_qr_kernel, dim3(qr_blocks), dim3(QR_THREADS), 0, st, n, d1, d2, dR);
    std::vector<double> hR((size_t)qr_blocks * 45);
    TV_HIP(hipMemcpyAsync(hR.data(), dR, hR.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    TV_HIP(hipStreamSynchronize(st));
    double Rm[45];
    for (int k = 0; k < 45; ++k) Rm[k] = hR[(size_t)k];
    for (int b = 1; b < qr_blocks; ++b) {
        int idx = 0;
        for (int rr = 0; rr < 9; ++rr) {
            double a[9];
            for (int j = 0; j < 9; ++j) a[j] = (j >= rr) ? hR[(size_t)b * 45 + idx + j - rr] : 0.0;
            givens_fold(Rm, a);
            idx += 9 - rr;
        }
    }
    std::vector<double> Rfull(81, 0.0);
    {
        int idx = 0;
        for (int k = 0; k < 9; ++k) { for (int j = k; j < 9; ++j) Rfull[(size_t)k * 9 + j] = Rm[idx + j - k]; idx += 9 - k; }
    }
    double F[9];
    smallest_right_singular_vector(Rfull, 9, 9, F);             // row-major F, x1^T F x2 = 0  (:36-38)
    if (F_out) std::memcpy(F_out, F, sizeof F);
    // 2. essential matrix and the four hypotheses
    double Kt[9], tmp[9], E[9], U[9], S[3], V[9];
    mat3_t(K, Kt); mat3_mul(Kt, F, tmp); mat3_mul(tmp, K, E);
    svd3(E, U, S, V);
    const double W[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1}, Wt[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
    double Vt[9], R1[9], R2[9];
    mat3_t(V, Vt);
    mat3_mul(U, W, tmp); mat3_mul(tmp, Vt, R1);
    mat3_mul(U, Wt, tmp); mat3_mul(tmp, Vt, R2);
    if (det3(R1) < 0) for (double& x : R1) x = -x;
    if (det3(R2) < 0) for (double& x : R2) x = -x;
    const double t1[3] = {U[2], U[5], U[8]}, t2[3] = {-U[2], -U[5], -U[8]};
    const double* Rs[4] = {R1, R1, R2, R2};
    const double* ts[4] = {t1, t2, t1, t2};                     // order b1..b4 of :61-64
    TvHyp h;
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, z3[3] = {0, 0, 0};
    projection(K, I3, z3, h.P1);
    for (int k = 0; k < 4; ++k) {
        projection(K, Rs[k], ts[k], h.P2[k]);
        mat3_t(Rs[k], h.Rt[k]);
        for (int j = 0; j < 3; ++j) h.t[k][j] = ts[k][j];
    }
    // 3. cheirality: every point in front of both cameras, for exactly one hypothesis
    TV_HIP(hipMemsetAsync(dF, 0, 4 * sizeof(int), st));
    hipLaunchKernelGGL(tv_cheirality_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, d1, d2, h, dF);
    int fails[4];
    TV_HIP(hipMemcpyAsync(fails, dF, sizeof fails, hipMemcpyDeviceToHost, st));
    TV_HIP(hipStreamSynchronize(st));
    if (fails_out) std::memcpy(fails_out, fails, sizeof fails);
    int winner = -1, n_ok = 0;
    for (int k = 0; k < 4; ++k)
        if (fails[k] == 0) { winner = k; ++n_ok; }
    if (n_ok != 1) { cleanup(); return fail(STBA_ERR_NO_SOLUTION, "two_view_init: no unique pose hypothesis passes the cheirality test"); }
    // AdjustRotationMatrix (two_view_simu.h:19-26): nearest rotation U V^T
    double Ua[9], Sa[3], Va[9], Vat[9];
    svd3(Rs[winner], Ua, Sa, Va);
    mat3_t(Va, Vat);
    mat3_mul(Ua, Vat, R_out);
    for (int j = 0; j < 3; ++j) t_out[j] = ts[winner][j];
    // 4. landmarks in frame 1
    if (pts_out) {
        hipLaunchKernelGGL(tv_triangulate_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, d1, d2, h, winner, dP);
        TV_HIP(hipMemcpyAsync(pts_out, dP, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
        TV_HIP(hipStreamSynchronize(st));
    }
    TV_HIP(hipGetLastError());
#undef TV_HIP
    cleanup();
    return STBA_OK;
}

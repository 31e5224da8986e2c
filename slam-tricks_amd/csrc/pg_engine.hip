// pg_engine.hip -- pose-graph SLAM on gfx950 (BASELINE config C4: 10k SE3 nodes, 40k relative-pose
// edges).  BUILD-DEFINED: the reference has no pose-graph code (SURVEY.md header fact 3); the
// conventions are the reference's Lie-group notes: right-multiplicative update T <- T exp(delta),
// tangent order [rho, theta] (st23-lie-group-v2/doc.tex:862-996, st21-lie/lie-group.tex:218-278),
// trajectory shape and ATE from st4-kalman/src/src/pose_simulation.cpp:17-88,198-209.
//
//   residual   r_ij = log(Z_ij^-1 T_i^-1 T_j)  in R^6
//   Jacobians  d r/d delta_j = Jr^-1(r),  d r/d delta_i = -Jr^-1(r) Ad(T_j^-1 T_i),
//              Jr^-1(r) = I + ad(r)/2 + ad(r)^2/12
//   solver     Levenberg-Marquardt (same control flow as the BA engine); the damped normal equations
//              (6 n_nodes unknowns, block-sparse) are solved matrix-free by block-Jacobi
//              preconditioned conjugate gradients: every kernel is HBM/latency bound, no MFMA.
//
// Data in HBM: poses [n][7] (two copies), edges (i, j) int32, meas [m][7], r [m][6], Ji/Jj [36][m] (COMPONENT-major: entry k
// of edge e at k m + e, so that the lanes of a wave -- consecutive edges -- read and write consecutive addresses; edge-major
// [m][36] made every load instruction touch 64 cache lines: the matrix-free product took 24 us for 25 MB),
// g [6n], Hd [n][36] (diagonal blocks), Minv [n][36], PCG vectors x r z p q [6n].
#include <algorithm>
#include <chrono>
#include <vector>

#include "ba_kernels.hpp"

namespace stba {
namespace {

// ---------------------------------------------------------------- SE3 helpers (7-double poses)
__host__ __device__ inline void quat_mul7(const double* a, const double* b, double* o) {
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
__host__ __device__ inline void rot_vec(const double* q, const double* v, double* o) {
    double R[9];
    quat_to_rot(q, R);
    for (int i = 0; i < 3; ++i) o[i] = R[i * 3] * v[0] + R[i * 3 + 1] * v[1] + R[i * 3 + 2] * v[2];
}
__host__ __device__ inline void se3_compose(const double* a, const double* b, double* out) {
    double q[4], t[3];
    quat_mul7(a, b, q);
    rot_vec(a, b + 4, t);
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) out[i] = q[i] / n;
    for (int i = 0; i < 3; ++i) out[4 + i] = t[i] + a[4 + i];
}
__host__ __device__ inline void se3_inverse(const double* a, double* out) {
    const double qc[4] = {-a[0], -a[1], -a[2], a[3]};
    double t[3];
    rot_vec(qc, a + 4, t);
    for (int i = 0; i < 4; ++i) out[i] = qc[i];
    for (int i = 0; i < 3; ++i) out[4 + i] = -t[i];
}
__host__ __device__ inline void hat3d(const double* v, double* M) {
    M[0] = 0; M[1] = -v[2]; M[2] = v[1]; M[3] = v[2]; M[4] = 0; M[5] = -v[0]; M[6] = -v[1]; M[7] = v[0]; M[8] = 0;
}
// Sophus SE3::log of (q, t) -> [rho, theta]
__host__ __device__ inline void se3_log7(const double* T, double* xi) {
    const double n2 = T[0] * T[0] + T[1] * T[1] + T[2] * T[2], qw = T[3];
    double k;
    if (n2 < 1e-20) k = 2.0 / qw - (2.0 / 3.0) * n2 / (qw * qw * qw);
    else { const double n = sqrt(n2); k = 2.0 * ((qw < 0) ? atan2(-n, -qw) : atan2(n, qw)) / n; }
    const double w[3] = {k * T[0], k * T[1], k * T[2]};
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double a, b;
    if (th2 < 1e-20) { a = 0.5 - th2 / 24.0; b = 1.0 / 6.0 - th2 / 120.0; }
    else { const double th = sqrt(th2); a = (1.0 - cos(th)) / th2; b = (th - sin(th)) / (th2 * th); }
    double K[9], V[9];
    hat3d(w, K);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const double k2 = K[i * 3] * K[j] + K[i * 3 + 1] * K[3 + j] + K[i * 3 + 2] * K[6 + j];
            V[i * 3 + j] = (i == j ? 1.0 : 0.0) + a * K[i * 3 + j] + b * k2;
        }
    // rho = V^-1 t
    const double aa = V[0], bb = V[1], cc = V[2], dd = V[3], ee = V[4], ff = V[5], gg = V[6], hh = V[7], ii = V[8];
    const double C0 = ee * ii - ff * hh, C1 = ff * gg - dd * ii, C2 = dd * hh - ee * gg;
    const double inv = 1.0 / (aa * C0 + bb * C1 + cc * C2);
    const double* t = T + 4;
    xi[0] = inv * (C0 * t[0] + (cc * hh - bb * ii) * t[1] + (bb * ff - cc * ee) * t[2]);
    xi[1] = inv * (C1 * t[0] + (aa * ii - cc * gg) * t[1] + (cc * dd - aa * ff) * t[2]);
    xi[2] = inv * (C2 * t[0] + (bb * gg - aa * hh) * t[1] + (aa * ee - bb * dd) * t[2]);
    xi[3] = w[0]; xi[4] = w[1]; xi[5] = w[2];
}
__host__ __device__ inline void se3_retract(const double* T, const double* d, double* out) {
    double e[7], R[9];
    so3_exp(d + 3, e);
    se3_exp_rt(d, R, e + 4);
    se3_compose(T, e, out);
}

// r, Ji, Jj of one edge
__device__ inline void pg_edge(const double* Ti, const double* Tj, const double* Z, double* r, double* Ji, double* Jj) {
    double Zi[7], Tii[7], A[7], E[7];
    se3_inverse(Z, Zi);
    se3_inverse(Ti, Tii);
    se3_compose(Tii, Tj, A);
    se3_compose(Zi, A, E);
    if (E[3] < 0) for (int k = 0; k < 4; ++k) E[k] = -E[k];
    se3_log7(E, r);
    if (!Ji) return;
    // Jr^-1 = I + ad/2 + ad^2/12, ad(xi) = [[hat(th), hat(rho)], [0, hat(th)]]
    double Hr[9], Ht[9], ad[36], Jr[36];
    hat3d(r, Hr); hat3d(r + 3, Ht);
    for (int k = 0; k < 36; ++k) ad[k] = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { ad[i * 6 + j] = Ht[i * 3 + j]; ad[i * 6 + 3 + j] = Hr[i * 3 + j]; ad[(3 + i) * 6 + 3 + j] = Ht[i * 3 + j]; }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0.0;
            for (int k = 0; k < 6; ++k) s += ad[i * 6 + k] * ad[k * 6 + j];
            Jr[i * 6 + j] = (i == j ? 1.0 : 0.0) + 0.5 * ad[i * 6 + j] + s / 12.0;
        }
    for (int k = 0; k < 36; ++k) Jj[k] = Jr[k];
    // Ad(T_j^-1 T_i) = [[R, hat(t) R], [0, R]]
    double Ainv[7], R[9], Hh[9], AdM[36];
    se3_inverse(A, Ainv);
    quat_to_rot(Ainv, R);
    hat3d(Ainv + 4, Hh);
    for (int k = 0; k < 36; ++k) AdM[k] = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += Hh[i * 3 + k] * R[k * 3 + j];
            AdM[i * 6 + j] = R[i * 3 + j]; AdM[i * 6 + 3 + j] = s; AdM[(3 + i) * 6 + 3 + j] = R[i * 3 + j];
        }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0.0;
            for (int k = 0; k < 6; ++k) s += Jr[i * 6 + k] * AdM[k * 6 + j];
            Ji[i * 6 + j] = -s;
        }
}

__device__ inline void block_sum2(double a, double b, double* out2) {
    __shared__ double s[2][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
    if (lane == 0) { s[0][w] = a; s[1][w] = b; }
    __syncthreads();
    if (threadIdx.x == 0) { out2[0] = s[0][0] + s[0][1] + s[0][2] + s[0][3]; out2[1] = s[1][0] + s[1][1] + s[1][2] + s[1][3]; }
}

// ---------------------------------------------------------------- kernels
__global__ __launch_bounds__(256) void pg_linearize_kernel(int n_edges, const double* __restrict__ poses,
                                                           const int* __restrict__ ei, const int* __restrict__ ej,
                                                           const double* __restrict__ meas,
                                                           const unsigned char* __restrict__ fixed, int with_jac,
                                                           double* __restrict__ r, double* __restrict__ Ji,
                                                           double* __restrict__ Jj, double* __restrict__ partial) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    double c = 0.0;
    if (e < n_edges) {
        const int i = ei[e], j = ej[e];
        double Ti[7], Tj[7], Z[7], re[6], ji[36], jj[36];
        for (int k = 0; k < 7; ++k) { Ti[k] = poses[(size_t)i * 7 + k]; Tj[k] = poses[(size_t)j * 7 + k]; Z[k] = meas[(size_t)e * 7 + k]; }
        pg_edge(Ti, Tj, Z, re, with_jac ? ji : nullptr, jj);
        for (int k = 0; k < 6; ++k) c += re[k] * re[k];
        if (r) for (int k = 0; k < 6; ++k) r[(size_t)e * 6 + k] = re[k];
        if (with_jac) {
            const bool fi = fixed && fixed[i], fj = fixed && fixed[j];
            for (int k = 0; k < 36; ++k) { Ji[(size_t)k * n_edges + e] = fi ? 0.0 : ji[k]; Jj[(size_t)k * n_edges + e] = fj ? 0.0 : jj[k]; }
        }
    }
    double out2[2];
    block_sum2(c, 0.0, out2);
    if (threadIdx.x == 0) partial[blockIdx.x] = out2[0];
}

// gradient and diagonal blocks: g_i += Ji^T r, Hd_i += Ji^T Ji (same for j), FP64 atomics
__global__ __launch_bounds__(256) void pg_accumulate_kernel(int n_edges, const int* __restrict__ ei, const int* __restrict__ ej,
                                                            const double* __restrict__ r, const double* __restrict__ Ji,
                                                            const double* __restrict__ Jj, double* __restrict__ g,
                                                            double* __restrict__ Hd) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int e = gid >> 1, side = gid & 1;
    if (e >= n_edges) return;
    const int node = side ? ej[e] : ei[e];
    const double* J = (side ? Jj : Ji) + e;
    const double* re = r + (size_t)e * 6;
    double Jl[36], rl[6];
    bool any = false;
    for (int k = 0; k < 36; ++k) { Jl[k] = J[(size_t)k * n_edges]; any |= (Jl[k] != 0.0); }
    if (!any) return;   // constant node
    for (int k = 0; k < 6; ++k) rl[k] = re[k];
    for (int a = 0; a < 6; ++a) {
        double s = 0.0;
        for (int k = 0; k < 6; ++k) s += Jl[k * 6 + a] * rl[k];
        unsafeAtomicAdd(&g[(size_t)node * 6 + a], s);
        for (int b = 0; b <= a; ++b) {
            double h = 0.0;
            for (int k = 0; k < 6; ++k) h += Jl[k * 6 + a] * Jl[k * 6 + b];
            unsafeAtomicAdd(&Hd[(size_t)node * 36 + a * 6 + b], h);
            if (b != a) unsafeAtomicAdd(&Hd[(size_t)node * 36 + b * 6 + a], h);
        }
    }
}

// LM diagonal + block-Jacobi preconditioner M_i = (Hd_i + diag(d_i))^-1 (6x6 Gauss-Jordan on an SPD block)
__global__ __launch_bounds__(256) void pg_precond_kernel(int n_nodes, const double* __restrict__ Hd, double* __restrict__ scale,
                                                         int init_scale, int use_scaling, double radius, double dmin, double dmax,
                                                         const unsigned char* __restrict__ fixed, double* __restrict__ d,
                                                         double* __restrict__ Minv) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_nodes) return;
    double A[36], B[36];
    for (int k = 0; k < 36; ++k) { A[k] = Hd[(size_t)i * 36 + k]; B[k] = 0.0; }
    const bool fx = fixed && fixed[i];
    for (int a = 0; a < 6; ++a) {
        const double h = A[a * 7];
        double s = 1.0;
        if (use_scaling) { if (init_scale) { s = 1.0 / (1.0 + sqrt(h)); scale[(size_t)i * 6 + a] = s; } else s = scale[(size_t)i * 6 + a]; }
        else if (init_scale) scale[(size_t)i * 6 + a] = 1.0;
        const double s2 = s * s;
        const double dv = fmin(fmax(h * s2, dmin), dmax) / radius / s2;
        d[(size_t)i * 6 + a] = dv;
        A[a * 7] += dv;
        B[a * 7] = 1.0;
    }
    if (fx) { for (int k = 0; k < 36; ++k) Minv[(size_t)i * 36 + k] = 0.0; return; }
    for (int c = 0; c < 6; ++c) {          // Gauss-Jordan without pivoting (SPD + damping)
        const double inv = 1.0 / A[c * 7];
        for (int k = 0; k < 6; ++k) { A[c * 6 + k] *= inv; B[c * 6 + k] *= inv; }
        for (int rr = 0; rr < 6; ++rr) {
            if (rr == c) continue;
            const double f = A[rr * 6 + c];
            for (int k = 0; k < 6; ++k) { A[rr * 6 + k] -= f * A[c * 6 + k]; B[rr * 6 + k] -= f * B[c * 6 + k]; }
        }
    }
    for (int k = 0; k < 36; ++k) Minv[(size_t)i * 36 + k] = B[k];
}

// q = D p   (then the edge kernel adds J^T J p)
__global__ __launch_bounds__(256) void pg_diag_mul_kernel(int n, const double* __restrict__ d, const double* __restrict__ p,
                                                          double* __restrict__ q, int use_d) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) q[i] = use_d ? d[i] * p[i] : 0.0;
}

__global__ __launch_bounds__(256) void pg_matvec_kernel(int n_edges, const int* __restrict__ ei, const int* __restrict__ ej,
                                                        const double* __restrict__ Ji, const double* __restrict__ Jj,
                                                        const double* __restrict__ p, double* __restrict__ q) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_edges) return;
    const int i = ei[e], j = ej[e];
    double A[36], B[36];
    for (int k = 0; k < 36; ++k) { A[k] = Ji[(size_t)k * n_edges + e]; B[k] = Jj[(size_t)k * n_edges + e]; }
    double pi[6], pj[6], t[6];
    for (int k = 0; k < 6; ++k) { pi[k] = p[(size_t)i * 6 + k]; pj[k] = p[(size_t)j * 6 + k]; }
    for (int a = 0; a < 6; ++a) {
        double s = 0.0;
        for (int k = 0; k < 6; ++k) s += A[a * 6 + k] * pi[k] + B[a * 6 + k] * pj[k];
        t[a] = s;
    }
    for (int k = 0; k < 6; ++k) {
        double si = 0.0, sj = 0.0;
        for (int a = 0; a < 6; ++a) { si += A[a * 6 + k] * t[a]; sj += B[a * 6 + k] * t[a]; }
        if (si != 0.0) unsafeAtomicAdd(&q[(size_t)i * 6 + k], si);
        if (sj != 0.0) unsafeAtomicAdd(&q[(size_t)j * 6 + k], sj);
    }
}

// partial[b] = {dot(a, b2), dot(a, a)}
__global__ __launch_bounds__(256) void pg_dot_kernel(int n, const double* __restrict__ a, const double* __restrict__ b2,
                                                     double* __restrict__ partial) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double x = 0.0, y = 0.0;
    if (i < n) { x = a[i] * b2[i]; y = a[i] * a[i]; }
    double out2[2];
    block_sum2(x, y, out2);
    if (threadIdx.x == 0) { partial[blockIdx.x * 2] = out2[0]; partial[blockIdx.x * 2 + 1] = out2[1]; }
}

// sum of nb per-workgroup partials, by the WHOLE workgroup (256 threads): a strided share per thread, a shuffle tree per
// wave, the four waves in order -- the same order in every workgroup and on every rank, so every one of them gets the same
// bits.  (Every thread summing all partials serially -- 275 dependent loads -- made pg_pcg_update_kernel 27.5 us long, 40 %
// of a PCG iteration: profiles/r3_a_c4_kernel_stats.csv.)
__device__ inline double sum_partials_dev(const double* partial, int nb, int stride, int off) {
    __shared__ double s_w[4];
    __shared__ double s_tot;
    double v = 0.0;
    for (int k = threadIdx.x; k < nb; k += 256) v += partial[k * stride + off];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();                                     // (the shared slots may still be read from a previous call)
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) s_tot = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
    __syncthreads();
    return s_tot;
}

// scal: [0] rz, [1] rr, [2] pq   (device-resident PCG scalars)
// init: r = b (b = -g), x = 0, z = M r, p = z; partial -> rz, rr
__global__ __launch_bounds__(256) void pg_pcg_init_kernel(int n_nodes, const double* __restrict__ g, const double* __restrict__ Minv,
                                                          double* __restrict__ x, double* __restrict__ r, double* __restrict__ z,
                                                          double* __restrict__ p, double* __restrict__ partial) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double a = 0.0, b = 0.0;
    if (i < n_nodes) {
        double rl[6];
        for (int k = 0; k < 6; ++k) { rl[k] = -g[(size_t)i * 6 + k]; r[(size_t)i * 6 + k] = rl[k]; x[(size_t)i * 6 + k] = 0.0; }
        for (int k = 0; k < 6; ++k) {
            double s = 0.0;
            for (int m = 0; m < 6; ++m) s += Minv[(size_t)i * 36 + k * 6 + m] * rl[m];
            z[(size_t)i * 6 + k] = s; p[(size_t)i * 6 + k] = s;
            a += rl[k] * s; b += rl[k] * rl[k];
        }
    }
    double out2[2];
    block_sum2(a, b, out2);
    if (threadIdx.x == 0) { partial[blockIdx.x * 2] = out2[0]; partial[blockIdx.x * 2 + 1] = out2[1]; }
}

// alpha = rz / pq; x += alpha p; r -= alpha q; z = M r; partial -> rz_new, rr
__global__ __launch_bounds__(256) void pg_pcg_update_kernel(int n_nodes, int nb_rz, const double* __restrict__ part_rz,
                                                            int nb_pq, const double* __restrict__ part_pq,
                                                            const double* __restrict__ Minv,
                                                            const double* __restrict__ p, const double* __restrict__ q,
                                                            double* __restrict__ x, double* __restrict__ r, double* __restrict__ z,
                                                            double* __restrict__ part_out) {
    const double rz = sum_partials_dev(part_rz, nb_rz, 2, 0);
    const double pq = sum_partials_dev(part_pq, nb_pq, 2, 0);
    const double alpha = (pq > 0.0) ? rz / pq : 0.0;
    const int i = blockIdx.x * 256 + threadIdx.x;
    double a = 0.0, b = 0.0;
    if (i < n_nodes) {
        double rl[6];
        for (int k = 0; k < 6; ++k) {
            const size_t o = (size_t)i * 6 + k;
            x[o] += alpha * p[o];
            rl[k] = r[o] - alpha * q[o];
            r[o] = rl[k];
        }
        for (int k = 0; k < 6; ++k) {
            double s = 0.0;
            for (int m = 0; m < 6; ++m) s += Minv[(size_t)i * 36 + k * 6 + m] * rl[m];
            z[(size_t)i * 6 + k] = s;
            a += rl[k] * s; b += rl[k] * rl[k];
        }
    }
    double out2[2];
    block_sum2(a, b, out2);
    if (threadIdx.x == 0) { part_out[blockIdx.x * 2] = out2[0]; part_out[blockIdx.x * 2 + 1] = out2[1]; }
}

// beta = rz_new / rz_old; p = z + beta p
__global__ __launch_bounds__(256) void pg_pcg_dir_kernel(int n, int nb, const double* __restrict__ part_new,
                                                         const double* __restrict__ part_old, const double* __restrict__ z,
                                                         double* __restrict__ p) {
    const double rzn = sum_partials_dev(part_new, nb, 2, 0), rzo = sum_partials_dev(part_old, nb, 2, 0);
    const double beta = (rzo > 0.0) ? rzn / rzo : 0.0;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = z[i] + beta * p[i];
}

// ---------------------------------------------------------------- one rank: a PCG iteration in THREE launches, no atomics
// (1) p^T q = p^T D p + |J p|^2 = sum_i d_i p_i^2 + sum_e |t_e|^2 with t_e = J_e [p_i; p_j]: the dot product needs no pass over
//     the finished q -- the edge kernel adds up |t_e|^2, the kernel that makes the direction p adds up d p^2.
// (2) q is never scattered: the edge kernel stores u_e = (Ji^T t_e | Jj^T t_e), 12 doubles per edge, component-major, and the
//     node kernel GATHERS q_i = d_i p_i + sum over the node's edge ends of u (a CSR of the ends built at create time) while it
//     updates x, r, z.  The scatter with FP64 atomics was the bound of the product: 480 k device-scope atomics per product
//     retire at ~20 G/s whatever the access pattern -- the product scaled linearly with the edge count at 1.65 G edges/s
//     (24.7 us at 40 k edges, 95 us at 160 k, 387 us at 640 k).  Measured on the way: one lane per edge END with a segmented
//     scan over a node's lanes and 6 atomics per node instead of 12 per edge (21.1 against 23.8 us: fewer atomics, but every
//     edge read twice and no coalescing); the whole loop as ONE persistent kernel with four grid barriers per iteration
//     (correct, 59 us per iteration against 40 us for the launches: a barrier across eight XCDs costs more than a kernel
//     boundary).
//     (Also measured: the direction p = z + beta p_old formed on the fly by the edge kernel and again by the node kernel, the
//     node term of the dot product added up by the edge ends -- TWO launches per iteration: 32 us against 30 us for three.
//     Every kernel that needs a scalar of the loop adds up its partial sums first, ~2 us each; two kernels that need three
//     each lose more than the launch they save.)
__global__ __launch_bounds__(256) void pg_edge_product_kernel(int n_edges, const int* __restrict__ ei, const int* __restrict__ ej,
                                                              const double* __restrict__ Ji, const double* __restrict__ Jj,
                                                              const double* __restrict__ p, double* __restrict__ u,
                                                              double* __restrict__ part_tt) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    double tt2 = 0.0;
    if (e < n_edges) {
        const int i = ei[e], j = ej[e];
        double A[36], B[36];
        for (int k = 0; k < 36; ++k) { A[k] = Ji[(size_t)k * n_edges + e]; B[k] = Jj[(size_t)k * n_edges + e]; }
        double pi[6], pj[6], t[6];
        for (int k = 0; k < 6; ++k) { pi[k] = p[(size_t)i * 6 + k]; pj[k] = p[(size_t)j * 6 + k]; }
        for (int a = 0; a < 6; ++a) {
            double s = 0.0;
            for (int k = 0; k < 6; ++k) s += A[a * 6 + k] * pi[k] + B[a * 6 + k] * pj[k];
            t[a] = s;
            tt2 += s * s;
        }
        for (int k = 0; k < 6; ++k) {
            double si = 0.0, sj = 0.0;
            for (int a = 0; a < 6; ++a) { si += A[a * 6 + k] * t[a]; sj += B[a * 6 + k] * t[a]; }
            u[(size_t)k * n_edges + e] = si;
            u[(size_t)(6 + k) * n_edges + e] = sj;
        }
    }
    double out2[2];
    block_sum2(tt2, 0.0, out2);
    if (threadIdx.x == 0) { part_tt[blockIdx.x * 2] = out2[0]; part_tt[blockIdx.x * 2 + 1] = 0.0; }
}

// q_i = d_i p_i + sum_ends u;  alpha = rz / (sum |t|^2 + sum d p^2);  x += alpha p;  r -= alpha q;  z = M r;  partial -> rz_new, rr
__global__ __launch_bounds__(256) void pg_pcg_update3_kernel(int n_nodes, int n_edges, int nb_rz, const double* __restrict__ part_rz,
                                                             int nb_tt, const double* __restrict__ part_tt, int nb_dp,
                                                             const double* __restrict__ part_dp, const double* __restrict__ Minv,
                                                             const double* __restrict__ d, const int* __restrict__ node_start,
                                                             const int* __restrict__ end_code, const double* __restrict__ u,
                                                             const double* __restrict__ p, double* __restrict__ x, double* __restrict__ r,
                                                             double* __restrict__ z, double* __restrict__ part_out) {
    const double rz = sum_partials_dev(part_rz, nb_rz, 2, 0);
    const double pq = sum_partials_dev(part_tt, nb_tt, 2, 0) + sum_partials_dev(part_dp, nb_dp, 2, 0);
    const double alpha = (pq > 0.0) ? rz / pq : 0.0;
    const int i = blockIdx.x * 256 + threadIdx.x;
    double a = 0.0, b = 0.0;
    if (i < n_nodes) {
        double ql[6], pl[6], rl[6];
        for (int k = 0; k < 6; ++k) { pl[k] = p[(size_t)i * 6 + k]; ql[k] = d[(size_t)i * 6 + k] * pl[k]; }
        const int e1 = node_start[i + 1];
        for (int c = node_start[i]; c < e1; ++c) {
            const int code = end_code[c], e = code >> 1, side = code & 1;
            const double* ue = u + (size_t)(6 * side) * n_edges + e;
            for (int k = 0; k < 6; ++k) ql[k] += ue[(size_t)k * n_edges];
        }
        for (int k = 0; k < 6; ++k) {
            const size_t o = (size_t)i * 6 + k;
            x[o] += alpha * pl[k];
            rl[k] = r[o] - alpha * ql[k];
            r[o] = rl[k];
        }
        for (int k = 0; k < 6; ++k) {
            double s = 0.0;
            for (int m = 0; m < 6; ++m) s += Minv[(size_t)i * 36 + k * 6 + m] * rl[m];
            z[(size_t)i * 6 + k] = s;
            a += rl[k] * s; b += rl[k] * rl[k];
        }
    }
    double out2[2];
    block_sum2(a, b, out2);
    if (threadIdx.x == 0) { part_out[blockIdx.x * 2] = out2[0]; part_out[blockIdx.x * 2 + 1] = out2[1]; }
}

// beta = rz_new / rz_old (first: p = z); p = z + beta p; partial -> sum d p^2
__global__ __launch_bounds__(256) void pg_pcg_dir3_kernel(int n, int nb, const double* __restrict__ part_new,
                                                          const double* __restrict__ part_old, int first, const double* __restrict__ z,
                                                          const double* __restrict__ d, double* __restrict__ p, double* __restrict__ part_dp) {
    double beta = 0.0;
    if (!first) {
        const double rzn = sum_partials_dev(part_new, nb, 2, 0), rzo = sum_partials_dev(part_old, nb, 2, 0);
        beta = (rzo > 0.0) ? rzn / rzo : 0.0;
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    double dp2 = 0.0;
    if (i < n) {
        const double pv = first ? z[i] : z[i] + beta * p[i];
        p[i] = pv;
        dp2 = d[i] * pv * pv;
    }
    double out2[2];
    block_sum2(dp2, 0.0, out2);
    if (threadIdx.x == 0) { part_dp[blockIdx.x * 2] = out2[0]; part_dp[blockIdx.x * 2 + 1] = 0.0; }
}

// trial poses + statistics: partial[b] = {|x_new - x|^2, |x|^2}
__global__ __launch_bounds__(256) void pg_update_kernel(int n_nodes, const double* __restrict__ poses, const double* __restrict__ dx,
                                                        const unsigned char* __restrict__ fixed, double* __restrict__ poses_new,
                                                        double* __restrict__ partial) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double a = 0.0, b = 0.0;
    if (i < n_nodes) {
        double T[7], d[6], Tn[7];
        for (int k = 0; k < 7; ++k) T[k] = poses[(size_t)i * 7 + k];
        const bool fx = fixed && fixed[i];
        for (int k = 0; k < 6; ++k) d[k] = fx ? 0.0 : dx[(size_t)i * 6 + k];
        if (fx) for (int k = 0; k < 7; ++k) Tn[k] = T[k];
        else se3_retract(T, d, Tn);
        for (int k = 0; k < 7; ++k) {
            poses_new[(size_t)i * 7 + k] = Tn[k];
            if (!fx) { a += (Tn[k] - T[k]) * (Tn[k] - T[k]); b += T[k] * T[k]; }
        }
    }
    double out2[2];
    block_sum2(a, b, out2);
    if (threadIdx.x == 0) { partial[blockIdx.x * 2] = out2[0]; partial[blockIdx.x * 2 + 1] = out2[1]; }
}

// model change terms: partial = {sum g x, sum x (Hx)}
__global__ __launch_bounds__(256) void pg_model_kernel(int n, const double* __restrict__ g, const double* __restrict__ x,
                                                       const double* __restrict__ hx, double* __restrict__ partial) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double a = 0.0, b = 0.0;
    if (i < n) { a = g[i] * x[i]; b = x[i] * hx[i]; }
    double out2[2];
    block_sum2(a, b, out2);
    if (threadIdx.x == 0) { partial[blockIdx.x * 2] = out2[0]; partial[blockIdx.x * 2 + 1] = out2[1]; }
}

template <class T>
int dalloc(T** p, size_t n) {
    *p = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(n, 1) * sizeof(T));
    if (e != hipSuccess) return fail(STBA_ERR_ALLOC, std::string("hipMalloc: ") + hipGetErrorString(e));
    return STBA_OK;
}
double wall() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace
}  // namespace stba

using namespace stba;

struct stba_pg {
    int n = 0, m = 0;
    hipStream_t st = nullptr;
    bool own = false;
    double* poses[2] = {nullptr, nullptr};
    int cur = 0;
    int *ei = nullptr, *ej = nullptr;
    double *meas = nullptr, *r = nullptr, *Ji = nullptr, *Jj = nullptr, *g = nullptr, *Hd = nullptr, *Minv = nullptr,
           *d = nullptr, *scale = nullptr, *x = nullptr, *rr = nullptr, *z = nullptr, *p = nullptr, *q = nullptr,
           *part_e = nullptr, *part_a = nullptr, *part_b = nullptr, *part_c = nullptr, *part_d = nullptr;
    unsigned char* fixed = nullptr;
    int nb_nodes = 1, nb_vec = 1, nb_edges = 1;
    // one rank: the edge ends (edge * 2 + side) of every node as a CSR, and the per-edge products u = (Ji^T t | Jj^T t) [12][m]
    int *node_start = nullptr, *end_code = nullptr;
    double* u = nullptr;
    // multi-GPU: this engine holds one shard of the EDGES, all nodes are replicated; the hook sums the
    // gradient | diagonal blocks, every matrix-vector product and the cost across ranks
    stba_allreduce_fn ar = nullptr;
    void* ar_user = nullptr;
    int rank = 0, world = 1;
    double* scalar = nullptr;           // one device double for the cost sums
};

namespace stba {
namespace {
void pg_free(stba_pg* g) {
    auto F = [](void* p) { if (p) (void)hipFree(p); };
    F(g->poses[0]); F(g->poses[1]); F(g->ei); F(g->ej); F(g->meas); F(g->r); F(g->Ji); F(g->Jj); F(g->g);
    F(g->Minv); F(g->d); F(g->scale); F(g->x); F(g->rr); F(g->z); F(g->p); F(g->q); F(g->part_e); F(g->part_a);
    F(g->part_b); F(g->part_c); F(g->part_d); F(g->fixed); F(g->scalar); F(g->node_start); F(g->end_code); F(g->u);
    if (g->own && g->st) (void)hipStreamDestroy(g->st);
    delete g;
}

double host_sum(hipStream_t st, const double* dev, int n, int stride, int off, std::vector<double>& buf) {
    buf.resize((size_t)n * stride);
    (void)hipMemcpyAsync(buf.data(), dev, buf.size() * sizeof(double), hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    double s = 0.0;
    for (int k = 0; k < n; ++k) s += buf[(size_t)k * stride + off];
    return s;
}

int pg_linearize(stba_pg* g, int which, bool jac) {
    hipLaunchKernelGGL(pg_linearize_kernel, dim3(g->nb_edges), dim3(256), 0, g->st, g->m, g->poses[which], g->ei, g->ej,
                       g->meas, g->fixed, jac ? 1 : 0, jac ? g->r : nullptr, g->Ji, g->Jj, g->part_e);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

// q = (J^T J [+ D]) v.  Sharded: every rank applies its edges, rank 0 alone adds the diagonal term, and the
// hook sums the 6n-vector (the only data-path collective of a PCG iteration: every other vector operation
// is replicated and bit-identical on all ranks).
int pg_apply(stba_pg* g, const double* v, double* q, bool with_d) {
    hipLaunchKernelGGL(pg_diag_mul_kernel, dim3(g->nb_vec), dim3(256), 0, g->st, 6 * g->n, g->d, v, q,
                       (with_d && g->rank == 0) ? 1 : 0);
    hipLaunchKernelGGL(pg_matvec_kernel, dim3(g->nb_edges), dim3(256), 0, g->st, g->m, g->ei, g->ej, g->Ji, g->Jj, v, q);
    STBA_HIP(hipGetLastError());
    if (g->ar && g->ar(g->ar_user, q, (size_t)6 * g->n, g->st) != 0) return fail(STBA_ERR_CALLBACK, "all-reduce hook failed");
    return STBA_OK;
}

// sum of a host scalar across ranks (through one device double and the hook)
int pg_sum_ranks(stba_pg* g, double* v) {
    if (!g->ar) return STBA_OK;
    STBA_HIP(hipMemcpyAsync(g->scalar, v, sizeof(double), hipMemcpyHostToDevice, g->st));
    if (g->ar(g->ar_user, g->scalar, 1, g->st) != 0) return fail(STBA_ERR_CALLBACK, "all-reduce hook failed");
    STBA_HIP(hipMemcpyAsync(v, g->scalar, sizeof(double), hipMemcpyDeviceToHost, g->st));
    STBA_HIP(hipStreamSynchronize(g->st));
    return STBA_OK;
}
}  // namespace
}  // namespace stba

extern "C" {

void stba_pcg_default_options(stba_pcg_options* o) {
    if (!o) return;
    o->max_iterations = 1000;
    o->relative_tolerance = 1e-12;
    o->check_every = 20;
}

int stba_pg_create(stba_pg** out, int n_nodes, int n_edges, const double* poses, const int* edge_i, const int* edge_j,
                   const double* meas, const unsigned char* node_fixed, void* hip_stream) {
    if (!out) return fail(STBA_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    if (n_nodes <= 0 || n_edges <= 0 || !poses || !edge_i || !edge_j || !meas)
        return fail(STBA_ERR_INVALID_ARGUMENT, "stba_pg_create: null or empty input");
    for (int e = 0; e < n_edges; ++e)
        if (edge_i[e] < 0 || edge_i[e] >= n_nodes || edge_j[e] < 0 || edge_j[e] >= n_nodes || edge_i[e] == edge_j[e])
            return fail(STBA_ERR_INVALID_ARGUMENT, "stba_pg_create: bad edge");
    STBA_TRY(require_device());
    stba_pg* g = new stba_pg();
    g->n = n_nodes; g->m = n_edges;
    if (hip_stream) g->st = reinterpret_cast<hipStream_t>(hip_stream);
    else { if (hipStreamCreate(&g->st) != hipSuccess) { delete g; return fail(STBA_ERR_HIP, "hipStreamCreate"); } g->own = true; }
    g->nb_nodes = (n_nodes + 255) / 256; g->nb_vec = (6 * n_nodes + 255) / 256; g->nb_edges = (n_edges + 255) / 256;
    int rc = STBA_OK;
    auto bail = [&](int c) { pg_free(g); return c; };
#define A_(call) do { rc = (call); if (rc != STBA_OK) return bail(rc); } while (0)
    const size_t n = (size_t)n_nodes, m = (size_t)n_edges;
    A_(dalloc(&g->poses[0], n * 7)); A_(dalloc(&g->poses[1], n * 7)); A_(dalloc(&g->ei, m)); A_(dalloc(&g->ej, m));
    A_(dalloc(&g->meas, m * 7)); A_(dalloc(&g->r, m * 6)); A_(dalloc(&g->Ji, m * 36)); A_(dalloc(&g->Jj, m * 36));
    A_(dalloc(&g->g, n * 42)); g->Hd = g->g + n * 6;      // [gradient 6n | diagonal blocks 36n]: one cross-rank sum
    A_(dalloc(&g->Minv, n * 36)); A_(dalloc(&g->d, n * 6)); A_(dalloc(&g->scalar, 1));
    A_(dalloc(&g->scale, n * 6)); A_(dalloc(&g->x, n * 6)); A_(dalloc(&g->rr, n * 6)); A_(dalloc(&g->z, n * 6));
    A_(dalloc(&g->p, n * 6)); A_(dalloc(&g->q, n * 6));
    const size_t np_ = (size_t)std::max(g->nb_vec, std::max(g->nb_nodes, g->nb_edges)) * 2 + 2;
    A_(dalloc(&g->part_e, np_)); A_(dalloc(&g->part_a, np_)); A_(dalloc(&g->part_b, np_)); A_(dalloc(&g->part_c, np_));
    A_(dalloc(&g->part_d, np_));
    if (node_fixed) A_(dalloc(&g->fixed, n));
    A_(dalloc(&g->node_start, n + 1)); A_(dalloc(&g->end_code, 2 * m)); A_(dalloc(&g->u, 12 * m));
#undef A_
    {   // edge ends sorted by node (counting sort; stable: a node's ends in edge order)
        std::vector<int> start((size_t)n_nodes + 1, 0), code(2 * m);
        for (int e = 0; e < n_edges; ++e) { ++start[(size_t)edge_i[e] + 1]; ++start[(size_t)edge_j[e] + 1]; }
        for (int i = 0; i < n_nodes; ++i) start[(size_t)i + 1] += start[(size_t)i];
        std::vector<int> fill(start.begin(), start.end() - 1);
        for (int e = 0; e < n_edges; ++e) {
            code[(size_t)fill[(size_t)edge_i[e]]++] = 2 * e;
            code[(size_t)fill[(size_t)edge_j[e]]++] = 2 * e + 1;
        }
        if (hipMemcpyAsync(g->node_start, start.data(), (n + 1) * sizeof(int), hipMemcpyHostToDevice, g->st) != hipSuccess ||
            hipMemcpyAsync(g->end_code, code.data(), 2 * m * sizeof(int), hipMemcpyHostToDevice, g->st) != hipSuccess ||
            hipStreamSynchronize(g->st) != hipSuccess)
            return bail(fail(STBA_ERR_HIP, "stba_pg_create: upload failed"));
    }
    if (hipMemcpyAsync(g->poses[0], poses, n * 7 * sizeof(double), hipMemcpyHostToDevice, g->st) != hipSuccess ||
        hipMemcpyAsync(g->ei, edge_i, m * sizeof(int), hipMemcpyHostToDevice, g->st) != hipSuccess ||
        hipMemcpyAsync(g->ej, edge_j, m * sizeof(int), hipMemcpyHostToDevice, g->st) != hipSuccess ||
        hipMemcpyAsync(g->meas, meas, m * 7 * sizeof(double), hipMemcpyHostToDevice, g->st) != hipSuccess ||
        (node_fixed && hipMemcpyAsync(g->fixed, node_fixed, n, hipMemcpyHostToDevice, g->st) != hipSuccess) ||
        hipStreamSynchronize(g->st) != hipSuccess)
        return bail(fail(STBA_ERR_HIP, "stba_pg_create: upload failed"));
    *out = g;
    return STBA_OK;
}

int stba_pg_set_allreduce(stba_pg* g, stba_allreduce_fn fn, void* user, int rank, int world_size) {
    if (!g || rank < 0 || world_size < 1 || rank >= world_size) return fail(STBA_ERR_INVALID_ARGUMENT, "bad argument");
    if (!fn && world_size > 1) return fail(STBA_ERR_INVALID_ARGUMENT, "stba_pg_set_allreduce: world_size > 1 needs a hook");
    g->ar = (world_size > 1 || fn) ? fn : nullptr;
    g->ar_user = user; g->rank = rank; g->world = world_size;
    return STBA_OK;
}

int stba_pg_destroy(stba_pg* g) {
    if (!g) return STBA_OK;
    if (g->st) (void)hipStreamSynchronize(g->st);
    pg_free(g);
    return STBA_OK;
}

int stba_pg_get_poses(stba_pg* g, double* poses) {
    if (!g || !poses) return fail(STBA_ERR_INVALID_ARGUMENT, "null argument");
    STBA_HIP(hipMemcpyAsync(poses, g->poses[g->cur], (size_t)g->n * 7 * sizeof(double), hipMemcpyDeviceToHost, g->st));
    STBA_HIP(hipStreamSynchronize(g->st));
    return STBA_OK;
}

int stba_pg_evaluate(stba_pg* g, double* cost, double* r, double* Ji, double* Jj) {
    if (!g) return fail(STBA_ERR_INVALID_ARGUMENT, "null engine");
    STBA_TRY(pg_linearize(g, g->cur, true));
    std::vector<double> buf;
    double c2 = host_sum(g->st, g->part_e, g->nb_edges, 1, 0, buf);
    STBA_TRY(pg_sum_ranks(g, &c2));
    if (cost) *cost = 0.5 * c2;
    if (r) STBA_HIP(hipMemcpyAsync(r, g->r, (size_t)g->m * 6 * sizeof(double), hipMemcpyDeviceToHost, g->st));
    // (the device keeps the Jacobians component-major, [36][m]; the caller gets them edge-major, [m][6][6])
    std::vector<double> ti, tj;
    if (Ji) { ti.resize((size_t)g->m * 36); STBA_HIP(hipMemcpyAsync(ti.data(), g->Ji, ti.size() * sizeof(double), hipMemcpyDeviceToHost, g->st)); }
    if (Jj) { tj.resize((size_t)g->m * 36); STBA_HIP(hipMemcpyAsync(tj.data(), g->Jj, tj.size() * sizeof(double), hipMemcpyDeviceToHost, g->st)); }
    STBA_HIP(hipStreamSynchronize(g->st));
    for (int k = 0; k < 36; ++k)
        for (int e = 0; e < g->m; ++e) {
            if (Ji) Ji[(size_t)e * 36 + k] = ti[(size_t)k * g->m + e];
            if (Jj) Jj[(size_t)e * 36 + k] = tj[(size_t)k * g->m + e];
        }
    return STBA_OK;
}

// measurement (bench.py --config c4): hipEvent-timed averages of the two kernels a solve is made of -- the residual +
// Jacobian kernel and one matrix-free product q = (J^T J + D) p (diagonal term + edge kernel) -- on the engine's stream
int stba_pg_time_kernels(stba_pg* g, int reps, double* ms_linearize, double* ms_matvec) {
    if (!g || reps <= 0 || !ms_linearize || !ms_matvec) return fail(STBA_ERR_INVALID_ARGUMENT, "bad argument");
    hipEvent_t e0, e1, e2;
    STBA_HIP(hipEventCreate(&e0)); STBA_HIP(hipEventCreate(&e1)); STBA_HIP(hipEventCreate(&e2));
    STBA_TRY(pg_linearize(g, g->cur, true));            // warm-up; also makes the Jacobians the products use
    // (a non-zero direction and damping -- the gradient of this linearisation, unit damping -- so that the product does the
    // atomics a real one does)
    STBA_HIP(hipMemsetAsync(g->g, 0, (size_t)g->n * 42 * sizeof(double), g->st));
    hipLaunchKernelGGL(pg_accumulate_kernel, dim3((2 * g->m + 255) / 256), dim3(256), 0, g->st, g->m, g->ei, g->ej, g->r, g->Ji, g->Jj, g->g, g->Hd);
    STBA_HIP(hipMemcpyAsync(g->p, g->g, (size_t)6 * g->n * sizeof(double), hipMemcpyDeviceToDevice, g->st));
    STBA_HIP(hipMemcpyAsync(g->d, g->g, (size_t)6 * g->n * sizeof(double), hipMemcpyDeviceToDevice, g->st));
    // the product as the one-rank solve runs it: the edge kernel (t_e, u_e = J_e^T t_e, |t_e|^2) -- the node kernel gathers u
    // while it updates x, r, z and is not a product kernel of its own
    int rc = STBA_OK;
    hipLaunchKernelGGL(pg_edge_product_kernel, dim3(g->nb_edges), dim3(256), 0, g->st, g->m, g->ei, g->ej, g->Ji, g->Jj, g->p, g->u, g->part_c);
    if (hipEventRecord(e0, g->st) != hipSuccess) rc = fail(STBA_ERR_HIP, "hipEventRecord");
    for (int k = 0; k < reps && rc == STBA_OK; ++k) rc = pg_linearize(g, g->cur, true);
    if (rc == STBA_OK && hipEventRecord(e1, g->st) != hipSuccess) rc = fail(STBA_ERR_HIP, "hipEventRecord");
    for (int k = 0; k < reps && rc == STBA_OK; ++k)
        hipLaunchKernelGGL(pg_edge_product_kernel, dim3(g->nb_edges), dim3(256), 0, g->st, g->m, g->ei, g->ej, g->Ji, g->Jj, g->p, g->u, g->part_c);
    if (rc == STBA_OK && hipEventRecord(e2, g->st) != hipSuccess) rc = fail(STBA_ERR_HIP, "hipEventRecord");
    if (rc == STBA_OK && hipStreamSynchronize(g->st) != hipSuccess) rc = fail(STBA_ERR_HIP, "hipStreamSynchronize");
    float a = 0.f, b = 0.f;
    if (rc == STBA_OK) { (void)hipEventElapsedTime(&a, e0, e1); (void)hipEventElapsedTime(&b, e1, e2); }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    *ms_linearize = a / reps; *ms_matvec = b / reps;
    return rc;
}

int stba_pg_solve(stba_pg* g, const stba_lm_options* opt_in, const stba_pcg_options* pcg_in, stba_lm_summary* summary,
                  double* trace, int* pcg_iterations_total) {
    if (!g) return fail(STBA_ERR_INVALID_ARGUMENT, "null engine");
    stba_lm_options opt;
    if (opt_in) opt = *opt_in; else stba_lm_default_options(&opt);
    stba_pcg_options pcg;
    if (pcg_in) pcg = *pcg_in; else stba_pcg_default_options(&pcg);
    stba_lm_summary s;
    memset(&s, 0, sizeof s);
    const double t0 = wall();
    std::vector<double> buf;
    const int N = 6 * g->n;
    int pcg_total = 0;

    auto linearize_full = [&](double* cost, double* gmax) -> int {
        STBA_TRY(pg_linearize(g, g->cur, true));
        double c2 = host_sum(g->st, g->part_e, g->nb_edges, 1, 0, buf);
        STBA_TRY(pg_sum_ranks(g, &c2));
        *cost = 0.5 * c2;
        STBA_HIP(hipMemsetAsync(g->g, 0, (size_t)g->n * 42 * sizeof(double), g->st));
        hipLaunchKernelGGL(pg_accumulate_kernel, dim3((2 * g->m + 255) / 256), dim3(256), 0, g->st, g->m, g->ei, g->ej, g->r,
                           g->Ji, g->Jj, g->g, g->Hd);
        if (g->ar && g->ar(g->ar_user, g->g, (size_t)g->n * 42, g->st) != 0) return fail(STBA_ERR_CALLBACK, "all-reduce hook failed");
        STBA_TRY(launch_absmax(g->g, (size_t)N, nullptr, 0, g->part_c, g->part_a, g->nb_vec, g->st));
        double gm = 0.0;
        STBA_HIP(hipMemcpyAsync(&gm, g->part_c, sizeof(double), hipMemcpyDeviceToHost, g->st));
        STBA_HIP(hipStreamSynchronize(g->st));
        *gmax = gm;
        return STBA_OK;
    };

    double cost = 0.0, gmax = 0.0;
    STBA_TRY(linearize_full(&cost, &gmax));
    s.initial_cost = cost;
    double radius = opt.initial_trust_region_radius, decrease = 2.0;
    bool scale_init = false;
    int iter = 0;
    if (trace) { memset(trace, 0, sizeof(double) * STBA_TRACE_COLS); trace[0] = cost; trace[2] = gmax; trace[5] = radius; trace[6] = 1; }
    s.termination_type = STBA_NO_CONVERGENCE; s.termination_reason = STBA_TERM_MAX_ITER;
    bool done = gmax <= opt.gradient_tolerance;
    if (done) { s.termination_type = STBA_CONVERGENCE; s.termination_reason = STBA_TERM_GRADIENT; }
    while (!done) {
        if (iter >= opt.max_num_iterations) break;
        if (radius < opt.min_trust_region_radius) { s.termination_type = STBA_CONVERGENCE; s.termination_reason = STBA_TERM_MIN_RADIUS; break; }
        ++iter;
        hipLaunchKernelGGL(pg_precond_kernel, dim3(g->nb_nodes), dim3(256), 0, g->st, g->n, g->Hd, g->scale, scale_init ? 0 : 1,
                           opt.jacobi_scaling, radius, opt.min_lm_diagonal, opt.max_lm_diagonal, g->fixed, g->d, g->Minv);
        scale_init = true;
        // ---- PCG on (J^T J + D) x = -g
        hipLaunchKernelGGL(pg_pcg_init_kernel, dim3(g->nb_nodes), dim3(256), 0, g->st, g->n, g->g, g->Minv, g->x, g->rr, g->z, g->p,
                           g->part_a);
        double* part_rz = g->part_a;
        double* part_new = g->part_b;
        const double rr0 = host_sum(g->st, part_rz, g->nb_nodes, 2, 1, buf);
        const double tol2 = pcg.relative_tolerance * pcg.relative_tolerance * rr0;
        int k = 0;
        bool ok = std::isfinite(rr0);
        // (one rank: three launches per iteration, the dot product p.q gathered on the way -- see pg_matvec_dot_kernel; with
        // several ranks the product needs a cross-rank sum between the edge kernel and the dot product: five launches)
        const bool fused3 = (g->ar == nullptr);
        if (fused3 && ok && rr0 > 0.0)
            hipLaunchKernelGGL(pg_pcg_dir3_kernel, dim3(g->nb_vec), dim3(256), 0, g->st, N, g->nb_nodes, part_rz, part_rz, 1, g->z, g->d, g->p,
                               g->part_d);
        while (ok && rr0 > 0.0 && k < pcg.max_iterations) {
            if (fused3) {
                hipLaunchKernelGGL(pg_edge_product_kernel, dim3(g->nb_edges), dim3(256), 0, g->st, g->m, g->ei, g->ej, g->Ji, g->Jj, g->p, g->u,
                                   g->part_c);
                hipLaunchKernelGGL(pg_pcg_update3_kernel, dim3(g->nb_nodes), dim3(256), 0, g->st, g->n, g->m, g->nb_nodes, part_rz, g->nb_edges,
                                   g->part_c, g->nb_vec, g->part_d, g->Minv, g->d, g->node_start, g->end_code, g->u, g->p, g->x, g->rr, g->z,
                                   part_new);
                hipLaunchKernelGGL(pg_pcg_dir3_kernel, dim3(g->nb_vec), dim3(256), 0, g->st, N, g->nb_nodes, part_new, part_rz, 0, g->z, g->d,
                                   g->p, g->part_d);
            } else {
                STBA_TRY(pg_apply(g, g->p, g->q, true));
                hipLaunchKernelGGL(pg_dot_kernel, dim3(g->nb_vec), dim3(256), 0, g->st, N, g->p, g->q, g->part_c);
                hipLaunchKernelGGL(pg_pcg_update_kernel, dim3(g->nb_nodes), dim3(256), 0, g->st, g->n, g->nb_nodes, part_rz, g->nb_vec,
                                   g->part_c, g->Minv, g->p, g->q, g->x, g->rr, g->z, part_new);
                hipLaunchKernelGGL(pg_pcg_dir_kernel, dim3(g->nb_vec), dim3(256), 0, g->st, N, g->nb_nodes, part_new, part_rz, g->z, g->p);
            }
            std::swap(part_rz, part_new);
            ++k;
            if (k % std::max(1, pcg.check_every) == 0) {
                const double rrk = host_sum(g->st, part_rz, g->nb_nodes, 2, 1, buf);
                if (!(rrk > tol2)) break;
                if (!std::isfinite(rrk)) { ok = false; break; }
            }
        }
        pcg_total += k;
        STBA_HIP(hipGetLastError());
        // ---- model change, trial point
        STBA_TRY(pg_apply(g, g->x, g->q, false));
        hipLaunchKernelGGL(pg_model_kernel, dim3(g->nb_vec), dim3(256), 0, g->st, N, g->g, g->x, g->q, g->part_c);
        const int nxt = g->cur ^ 1;
        double* part_upd = g->part_a;     // the PCG scalars are no longer needed
        hipLaunchKernelGGL(pg_update_kernel, dim3(g->nb_nodes), dim3(256), 0, g->st, g->n, g->poses[g->cur], g->x, g->fixed,
                           g->poses[nxt], part_upd);
        STBA_TRY(pg_linearize(g, nxt, false));
        const double gx = host_sum(g->st, g->part_c, g->nb_vec, 2, 0, buf);
        double xhx = 0.0;
        for (int b = 0; b < g->nb_vec; ++b) xhx += buf[(size_t)b * 2 + 1];
        const double step2 = host_sum(g->st, part_upd, g->nb_nodes, 2, 0, buf);
        double x2 = 0.0;
        for (int b = 0; b < g->nb_nodes; ++b) x2 += buf[(size_t)b * 2 + 1];
        double nc2 = host_sum(g->st, g->part_e, g->nb_edges, 1, 0, buf);
        STBA_TRY(pg_sum_ranks(g, &nc2));
        const double new_cost = 0.5 * nc2;
        const double model_change = -gx - 0.5 * xhx;
        const double step_norm = std::sqrt(step2), x_norm = std::sqrt(x2);
        ok = ok && model_change > 0.0 && std::isfinite(model_change) && std::isfinite(new_cost);
        double cost_change = 0.0, rho = 0.0;
        bool accepted = false, stop = false;
        if (ok) {
            cost_change = cost - new_cost;
            rho = cost_change / model_change;
            if (step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance)) {
                s.termination_type = STBA_CONVERGENCE; s.termination_reason = STBA_TERM_PARAMETER; stop = true;
            } else if (std::fabs(cost_change) <= opt.function_tolerance * cost) {
                if (rho > opt.min_relative_decrease) { g->cur = nxt; cost = new_cost; ++s.num_successful_steps; accepted = true; }
                s.termination_type = STBA_CONVERGENCE; s.termination_reason = STBA_TERM_FUNCTION; stop = true;
            }
            if (!stop) accepted = rho > opt.min_relative_decrease;
        }
        if (trace) {
            double* tr = trace + (size_t)iter * STBA_TRACE_COLS;
            tr[0] = ok ? new_cost : cost; tr[1] = cost_change; tr[2] = gmax; tr[3] = ok ? step_norm : 0.0; tr[4] = rho; tr[5] = radius;
            tr[6] = accepted ? 1 : 0;
        }
        if (stop) break;
        if (accepted) {
            g->cur = nxt;
            ++s.num_successful_steps;
            const double t = 2.0 * rho - 1.0;
            radius = std::min(opt.max_trust_region_radius, radius / std::max(1.0 / 3.0, 1.0 - t * t * t));
            decrease = 2.0;
            double c2;
            STBA_TRY(linearize_full(&c2, &gmax));
            cost = c2;
            if (trace) { trace[(size_t)iter * STBA_TRACE_COLS + 2] = gmax; trace[(size_t)iter * STBA_TRACE_COLS + 5] = radius; }
            if (gmax <= opt.gradient_tolerance) { s.termination_type = STBA_CONVERGENCE; s.termination_reason = STBA_TERM_GRADIENT; break; }
        } else {
            ++s.num_unsuccessful_steps;
            radius /= decrease; decrease *= 2.0;
            if (trace) trace[(size_t)iter * STBA_TRACE_COLS + 5] = radius;
        }
        if (opt.minimizer_progress_to_stdout)
            printf("%4d  %.6e   % .2e    %.2e   %.2e  % .2e  %.2e  pcg %d\n", iter, cost, cost_change, gmax, step_norm, rho, radius, k);
    }
    s.num_iterations = iter; s.final_cost = cost; s.final_radius = radius; s.final_gradient_max_norm = gmax;
    s.seconds_total = wall() - t0;
    if (summary) *summary = s;
    if (pcg_iterations_total) *pcg_iterations_total = pcg_total;
    return STBA_OK;
}

}  // extern "C"

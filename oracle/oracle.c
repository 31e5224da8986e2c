/*
 * oracle.c -- CPU restatement of the slam-tricks NLS path.  TEST INFRASTRUCTURE ONLY
 * (see oracle.h for the scope, conventions and parity status; every function cites the
 * reference file:line it follows).
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ======================================================================================
 * SO3 / SE3 (Sophus semantics; quaternion storage x,y,z,w)
 * ==================================================================================== */

static void hat3(const double v[3], double M[9]) {
    M[0] = 0;     M[1] = -v[2]; M[2] = v[1];
    M[3] = v[2];  M[4] = 0;     M[5] = -v[0];
    M[6] = -v[1]; M[7] = v[0];  M[8] = 0;
}

void orc_quat_to_rot(const double q[4], double R[9]) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double xx = x * x, yy = y * y, zz = z * z;
    const double xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    R[0] = 1 - 2 * (yy + zz); R[1] = 2 * (xy - wz);     R[2] = 2 * (xz + wy);
    R[3] = 2 * (xy + wz);     R[4] = 1 - 2 * (xx + zz); R[5] = 2 * (yz - wx);
    R[6] = 2 * (xz - wy);     R[7] = 2 * (yz + wx);     R[8] = 1 - 2 * (xx + yy);
}

void orc_rot_to_quat(const double R[9], double q[4]) {
    /* Shepperd's method; returns unit quaternion with arbitrary sign */
    const double tr = R[0] + R[4] + R[8];
    double x, y, z, w;
    if (tr > 0) {
        double s = sqrt(tr + 1.0) * 2;
        w = 0.25 * s; x = (R[7] - R[5]) / s; y = (R[2] - R[6]) / s; z = (R[3] - R[1]) / s;
    } else if (R[0] > R[4] && R[0] > R[8]) {
        double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
        w = (R[7] - R[5]) / s; x = 0.25 * s; y = (R[1] + R[3]) / s; z = (R[2] + R[6]) / s;
    } else if (R[4] > R[8]) {
        double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
        w = (R[2] - R[6]) / s; x = (R[1] + R[3]) / s; y = 0.25 * s; z = (R[5] + R[7]) / s;
    } else {
        double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
        w = (R[3] - R[1]) / s; x = (R[2] + R[6]) / s; y = (R[5] + R[7]) / s; z = 0.25 * s;
    }
    const double n = sqrt(x * x + y * y + z * z + w * w);
    q[0] = x / n; q[1] = y / n; q[2] = z / n; q[3] = w / n;
}

void orc_quat_mul(const double a[4], const double b[4], double o[4]) {
    const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
    const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
    o[0] = aw * bx + ax * bw + ay * bz - az * by;
    o[1] = aw * by - ax * bz + ay * bw + az * bx;
    o[2] = aw * bz + ax * by - ay * bx + az * bw;
    o[3] = aw * bw - ax * bx - ay * by - az * bz;
}

/* Sophus SO3::exp: quaternion (sin(th/2)/th * w, cos(th/2)) with Taylor branch near 0 */
void orc_so3_exp(const double w[3], double q[4]) {
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double imag, real;
    if (th2 < 1e-20) {
        const double th4 = th2 * th2;
        imag = 0.5 - th2 / 48.0 + th4 / 3840.0;
        real = 1.0 - th2 / 8.0 + th4 / 384.0;
    } else {
        const double th = sqrt(th2);
        imag = sin(0.5 * th) / th;
        real = cos(0.5 * th);
    }
    q[0] = imag * w[0]; q[1] = imag * w[1]; q[2] = imag * w[2]; q[3] = real;
}

/* Sophus SO3::log */
void orc_so3_log(const double q[4], double w[3]) {
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
    const double qw = q[3];
    double k;
    if (n2 < 1e-20) {
        k = 2.0 / qw - (2.0 / 3.0) * n2 / (qw * qw * qw);
    } else {
        const double n = sqrt(n2);
        /* atan2(-n,-w) branch of Sophus keeps the angle in (-pi, pi] */
        const double at = (qw < 0) ? atan2(-n, -qw) : atan2(n, qw);
        k = 2.0 * at / n;
    }
    w[0] = k * q[0]; w[1] = k * q[1]; w[2] = k * q[2];
}

void orc_so3_plus(const double q[4], const double d[3], double out[4]) {
    double e[4], o[4];
    orc_so3_exp(d, e);
    orc_quat_mul(q, e, o);
    /* Sophus normalises after group multiplication when the norm drifts */
    const double n = sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
    out[0] = o[0] / n; out[1] = o[1] / n; out[2] = o[2] / n; out[3] = o[3] / n;
}

void orc_so3_plus_jacobian(const double q[4], double J[12]) {
    /* notes.tex:131-144: c0=w/2 c1=z/2 c2=-c1 c3=y/2 c4=x/2 c5=-c4 c6=-c3 */
    const double c0 = 0.5 * q[3], c1 = 0.5 * q[2], c2 = -c1, c3 = 0.5 * q[1], c4 = 0.5 * q[0],
                 c5 = -c4, c6 = -c3;
    J[0] = c0; J[1] = c2;  J[2] = c3;
    J[3] = c1; J[4] = c0;  J[5] = c5;
    J[6] = c6; J[7] = c4;  J[8] = c0;
    J[9] = c5; J[10] = c6; J[11] = c2;
}

void orc_so3r3_plus(const double x[3], const double d[3], double out[3]) {
    double qa[4], qb[4], qc[4];
    orc_so3_exp(x, qa);
    orc_so3_exp(d, qb);
    orc_quat_mul(qa, qb, qc);
    const double n = sqrt(qc[0] * qc[0] + qc[1] * qc[1] + qc[2] * qc[2] + qc[3] * qc[3]);
    for (int i = 0; i < 4; ++i) qc[i] /= n;
    orc_so3_log(qc, out);
}

/* V(theta) of SE3::exp:  I + (1-cos)/th^2 K + (th - sin)/th^3 K^2 */
static void so3_left_jacobian(const double w[3], double V[9]) {
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double K[9], K2[9];
    hat3(w, K);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += K[i * 3 + k] * K[k * 3 + j];
            K2[i * 3 + j] = s;
        }
    double a, b;
    if (th2 < 1e-20) {
        a = 0.5 - th2 / 24.0;
        b = 1.0 / 6.0 - th2 / 120.0;
    } else {
        const double th = sqrt(th2);
        a = (1.0 - cos(th)) / th2;
        b = (th - sin(th)) / (th2 * th);
    }
    for (int i = 0; i < 9; ++i) V[i] = a * K[i] + b * K2[i];
    V[0] += 1; V[4] += 1; V[8] += 1;
}

void orc_se3_exp(const double xi[6], double q[4], double t[3]) {
    double V[9];
    orc_so3_exp(xi + 3, q);
    so3_left_jacobian(xi + 3, V);
    for (int i = 0; i < 3; ++i) t[i] = V[i * 3] * xi[0] + V[i * 3 + 1] * xi[1] + V[i * 3 + 2] * xi[2];
}

static int solve3(const double A[9], const double b[3], double x[3]) {
    const double a = A[0], bb = A[1], c = A[2], d = A[3], e = A[4], f = A[5], g = A[6], h = A[7], i = A[8];
    const double C0 = e * i - f * h, C1 = f * g - d * i, C2 = d * h - e * g;
    const double det = a * C0 + bb * C1 + c * C2;
    if (det == 0.0 || !isfinite(det)) return 1;
    const double inv = 1.0 / det;
    x[0] = inv * (C0 * b[0] + (c * h - bb * i) * b[1] + (bb * f - c * e) * b[2]);
    x[1] = inv * (C1 * b[0] + (a * i - c * g) * b[1] + (c * d - a * f) * b[2]);
    x[2] = inv * (C2 * b[0] + (bb * g - a * h) * b[1] + (a * e - bb * d) * b[2]);
    return 0;
}

void orc_se3_log(const double q[4], const double t[3], double xi[6]) {
    double V[9];
    orc_so3_log(q, xi + 3);
    so3_left_jacobian(xi + 3, V);
    solve3(V, t, xi);
}

/* ======================================================================================
 * Reprojection factor
 * ==================================================================================== */

void orc_reproj_residual(const double q[4], const double t[3], const double L[3],
                         const double f[2], double r[2]) {
    double R[9];
    orc_quat_to_rot(q, R);
    const double d0 = L[0] - t[0], d1 = L[1] - t[1], d2 = L[2] - t[2];
    const double x = R[0] * d0 + R[3] * d1 + R[6] * d2;   /* R^T (L - t) */
    const double y = R[1] * d0 + R[4] * d1 + R[7] * d2;
    const double z = R[2] * d0 + R[5] * d1 + R[8] * d2;
    r[0] = x / z - f[0];
    r[1] = y / z - f[1];
}

void orc_reproj_residual_ambient(const double q[4], const double t[3], const double L[3],
                                 const double f[2], double r[2]) {
    /* SE3_CtoW.inverse() * L = conj(q) * (L - t) evaluated as Eigen's
     * quaternion _transformVector: v + w*uv + u x uv, uv = 2 u x v, with u = -q.xyz, w = q.w */
    const double u[3] = {-q[0], -q[1], -q[2]};
    const double w = q[3];
    const double v[3] = {L[0] - t[0], L[1] - t[1], L[2] - t[2]};
    const double uv[3] = {2 * (u[1] * v[2] - u[2] * v[1]), 2 * (u[2] * v[0] - u[0] * v[2]),
                          2 * (u[0] * v[1] - u[1] * v[0])};
    const double x = v[0] + w * uv[0] + (u[1] * uv[2] - u[2] * uv[1]);
    const double y = v[1] + w * uv[1] + (u[2] * uv[0] - u[0] * uv[2]);
    const double z = v[2] + w * uv[2] + (u[0] * uv[1] - u[1] * uv[0]);
    r[0] = x / z - f[0];
    r[1] = y / z - f[1];
}

void orc_reproj_jacobian(const double q[4], const double t[3], const double L[3],
                         double Jc[12], double Jp[6], int rot_mode) {
    double R[9];
    orc_quat_to_rot(q, R);
    const double d0 = L[0] - t[0], d1 = L[1] - t[1], d2 = L[2] - t[2];
    const double x = R[0] * d0 + R[3] * d1 + R[6] * d2;
    const double y = R[1] * d0 + R[4] * d1 + R[7] * d2;
    const double z = R[2] * d0 + R[5] * d1 + R[8] * d2;
    const double iz = 1.0 / z;
    /* pn_pc, solver.hpp:190-192 */
    const double A[6] = {iz, 0, -x * iz * iz, 0, iz, -y * iz * iz};
    /* rotation block: A * hat(h) */
    double h[3];
    if (rot_mode == 0) {
        h[0] = x; h[1] = y; h[2] = z;            /* hat(pInC): right-perturbation derivative */
    } else {
        /* solver.hpp:195: pn_pc * (R^-1 hat(Pw) R) = pn_pc * hat(R^-1 Pw) */
        h[0] = R[0] * L[0] + R[3] * L[1] + R[6] * L[2];
        h[1] = R[1] * L[0] + R[4] * L[1] + R[7] * L[2];
        h[2] = R[2] * L[0] + R[5] * L[1] + R[8] * L[2];
    }
    double H[9];
    hat3(h, H);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0, sp = 0;
            for (int k = 0; k < 3; ++k) {
                s += A[i * 3 + k] * H[k * 3 + j];
                sp += A[i * 3 + k] * R[j * 3 + k];   /* (A R^T)_{ij} = sum_k A_ik R_jk */
            }
            if (Jc) { Jc[i * 6 + j] = s; Jc[i * 6 + 3 + j] = -sp; }   /* solver.hpp:198 */
            if (Jp) Jp[i * 3 + j] = sp;
        }
}

/* ======================================================================================
 * options
 * ==================================================================================== */
void orc_lm_default_options(orc_lm_options* o) {
    o->max_num_iterations = 50;
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->function_tolerance = 1e-6;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8;
    o->jacobi_scaling = 1;
    o->num_threads = 1;
    o->fixed_iterations = 0;
    o->function_tolerance_takes_step = 1;
}

/* ======================================================================================
 * dense Cholesky (row-major, lower, in place), blocked right-looking
 * ==================================================================================== */
#define CH_NB 64

static inline double dotk(const double* a, const double* b, int k) {
    double s = 0;
#pragma omp simd reduction(+ : s)
    for (int m = 0; m < k; ++m) s += a[m] * b[m];
    return s;
}

int orc_cholesky_lower(double* A, int n, int num_threads) {
    (void)num_threads;
    for (int k0 = 0; k0 < n; k0 += CH_NB) {
        const int kb = (n - k0 < CH_NB) ? n - k0 : CH_NB;
        /* diagonal block, unblocked */
        for (int j = k0; j < k0 + kb; ++j) {
            double d = A[(size_t)j * n + j] - dotk(&A[(size_t)j * n + k0], &A[(size_t)j * n + k0], j - k0);
            if (!(d > 0.0) || !isfinite(d)) return j + 1;
            d = sqrt(d);
            A[(size_t)j * n + j] = d;
            const double inv = 1.0 / d;
            for (int i = j + 1; i < k0 + kb; ++i) {
                double s = A[(size_t)i * n + j] - dotk(&A[(size_t)i * n + k0], &A[(size_t)j * n + k0], j - k0);
                A[(size_t)i * n + j] = s * inv;
            }
        }
        const int r0 = k0 + kb;
        if (r0 >= n) break;
        /* panel solve: rows r0..n */
#pragma omp parallel for schedule(static) num_threads(num_threads > 0 ? num_threads : 1)
        for (int i = r0; i < n; ++i) {
            double* ai = &A[(size_t)i * n + k0];
            for (int j = 0; j < kb; ++j) {
                const double* lj = &A[(size_t)(k0 + j) * n + k0];
                double s = ai[j] - dotk(ai, lj, j);
                ai[j] = s / lj[j];
            }
        }
        /* trailing update, lower triangle: A[i][j] -= P[i].P[j]; 4x4 register tiles */
#pragma omp parallel for schedule(dynamic, 4) num_threads(num_threads > 0 ? num_threads : 1)
        for (int ib = r0; ib < n; ib += 4) {
            const int ie = (ib + 4 < n) ? ib + 4 : n;
            for (int jb = r0; jb < ie; jb += 4) {
                const int je = (jb + 4 < ie) ? jb + 4 : ie;
                if (ie - ib == 4 && je - jb == 4) {
                    const double* a0 = &A[(size_t)(ib + 0) * n + k0];
                    const double* a1 = &A[(size_t)(ib + 1) * n + k0];
                    const double* a2 = &A[(size_t)(ib + 2) * n + k0];
                    const double* a3 = &A[(size_t)(ib + 3) * n + k0];
                    const double* b0 = &A[(size_t)(jb + 0) * n + k0];
                    const double* b1 = &A[(size_t)(jb + 1) * n + k0];
                    const double* b2 = &A[(size_t)(jb + 2) * n + k0];
                    const double* b3 = &A[(size_t)(jb + 3) * n + k0];
                    double c00 = 0, c01 = 0, c02 = 0, c03 = 0, c10 = 0, c11 = 0, c12 = 0, c13 = 0;
                    double c20 = 0, c21 = 0, c22 = 0, c23 = 0, c30 = 0, c31 = 0, c32 = 0, c33 = 0;
#pragma omp simd reduction(+ : c00, c01, c02, c03, c10, c11, c12, c13, c20, c21, c22, c23, c30, c31, c32, c33)
                    for (int m = 0; m < kb; ++m) {
                        const double x0 = a0[m], x1 = a1[m], x2 = a2[m], x3 = a3[m];
                        const double y0 = b0[m], y1 = b1[m], y2 = b2[m], y3 = b3[m];
                        c00 += x0 * y0; c01 += x0 * y1; c02 += x0 * y2; c03 += x0 * y3;
                        c10 += x1 * y0; c11 += x1 * y1; c12 += x1 * y2; c13 += x1 * y3;
                        c20 += x2 * y0; c21 += x2 * y1; c22 += x2 * y2; c23 += x2 * y3;
                        c30 += x3 * y0; c31 += x3 * y1; c32 += x3 * y2; c33 += x3 * y3;
                    }
                    double* r0p = &A[(size_t)(ib + 0) * n + jb];
                    double* r1p = &A[(size_t)(ib + 1) * n + jb];
                    double* r2p = &A[(size_t)(ib + 2) * n + jb];
                    double* r3p = &A[(size_t)(ib + 3) * n + jb];
                    r0p[0] -= c00; r0p[1] -= c01; r0p[2] -= c02; r0p[3] -= c03;
                    r1p[0] -= c10; r1p[1] -= c11; r1p[2] -= c12; r1p[3] -= c13;
                    r2p[0] -= c20; r2p[1] -= c21; r2p[2] -= c22; r2p[3] -= c23;
                    r3p[0] -= c30; r3p[1] -= c31; r3p[2] -= c32; r3p[3] -= c33;
                } else {
                    for (int i = ib; i < ie; ++i)
                        for (int j = jb; j < je && j <= i; ++j)
                            A[(size_t)i * n + j] -= dotk(&A[(size_t)i * n + k0], &A[(size_t)j * n + k0], kb);
                }
            }
        }
    }
    return 0;
}

void orc_cholesky_solve(const double* L, int n, double* b) {
    for (int i = 0; i < n; ++i) {
        double s = b[i] - dotk(&L[(size_t)i * n], b, i);
        b[i] = s / L[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * n + i] * b[k];
        b[i] = s / L[(size_t)i * n + i];
    }
}

/* column-oriented back substitution variant for big n (cache friendly): L^T x = y */
static void chol_solve_big(const double* L, int n, double* b) {
    for (int i = 0; i < n; ++i) {
        double s = b[i] - dotk(&L[(size_t)i * n], b, i);
        b[i] = s / L[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        const double xi = b[i] / L[(size_t)i * n + i];
        b[i] = xi;
        const double* row = &L[(size_t)i * n];
#pragma omp simd
        for (int k = 0; k < i; ++k) b[k] -= row[k] * xi;
    }
}

/* ======================================================================================
 * Bundle adjustment
 * ==================================================================================== */

static inline int cam_dof_fixed(const orc_ba_problem* p, int c, int d) {
    return p->cam_fixed ? p->cam_fixed[c * 6 + d] : 0;
}
static inline int pt_is_fixed(const orc_ba_problem* p, int j) { return p->pt_fixed ? p->pt_fixed[j] : 0; }

static int g_threads = 1;   /* set by orc_ba_solve from options->num_threads */

/* Optional external dense solver for the reduced camera system (bench.py's cpu_baseline leg: LAPACK dpotrf/dpotrs of
 * the OpenBLAS that numpy/scipy ship, installed from oracle_py.use_lapack()).  fn(S, n, x): S row-major, lower triangle
 * valid, may be overwritten; x holds the right-hand side on entry and the solution on return; returns 0, or k > 0 if
 * the leading minor of order k is not positive definite.  NULL: the blocked C factorisation above. */
typedef int (*orc_dense_solver_fn)(double* S, int n, double* x);
static orc_dense_solver_fn g_dense_solver = 0;
void orc_set_dense_solver(orc_dense_solver_fn fn) { g_dense_solver = fn; }

/* Index of a problem's observation structure: the observations of every camera and of every landmark (ascending, i.e. in the order
 * every sum below has always run in), and the maximal runs of consecutive observations of one landmark.  ROUND 6: until round 5
 * every thread of the block assembly, of the Schur complement and of the back-substitution walked ALL observations and kept the
 * cameras / landmarks with index % threads == its id -- thread-count independent sums, but O(observations x threads) memory
 * traffic: 16 / 32 / 64 / 128 threads ran 2.30 / 2.00 / 1.34 / 0.67 LM it/s at C5 on the 256-thread host.  With the
 * index a thread visits only what it owns; every sum keeps its order (ascending observation index), so the results are the same
 * bits as before and still do not depend on the thread count.  Cached: rebuilt when the observation arrays change (checksum). */
typedef struct {
    int n_obs, n_cams, n_pts;
    unsigned long long sum;
    int *cam_start, *cam_obs, *pt_start, *pt_obs, *run_start, *run_of;
    int n_runs;
} ba_index;
static ba_index g_ix = {0, 0, 0, 0ull, 0, 0, 0, 0, 0, 0, 0};

static unsigned long long ba_index_checksum(const orc_ba_problem* p) {
    unsigned long long s = 1469598103934665603ull;
    for (int i = 0; i < p->n_obs; ++i)
        s += ((unsigned long long)(unsigned)p->obs_cam[i] + 1ull) * (2ull * (unsigned long long)i + 1ull) +
             (((unsigned long long)(unsigned)p->obs_pt[i] + 1ull) << 21) * (2ull * (unsigned long long)i + 7ull);
    return s;
}
static const ba_index* ba_index_get(const orc_ba_problem* p) {
    const unsigned long long sum = ba_index_checksum(p);
    ba_index* ix = &g_ix;
    if (ix->cam_start && ix->n_obs == p->n_obs && ix->n_cams == p->n_cams && ix->n_pts == p->n_pts && ix->sum == sum) return ix;
    free(ix->cam_start); free(ix->cam_obs); free(ix->pt_start); free(ix->pt_obs); free(ix->run_start); free(ix->run_of);
    const int no = p->n_obs, nc = p->n_cams, np = p->n_pts;
    ix->n_obs = no; ix->n_cams = nc; ix->n_pts = np; ix->sum = sum;
    ix->cam_start = calloc((size_t)nc + 1, sizeof(int)); ix->cam_obs = malloc(sizeof(int) * (size_t)(no > 0 ? no : 1));
    ix->pt_start = calloc((size_t)np + 1, sizeof(int)); ix->pt_obs = malloc(sizeof(int) * (size_t)(no > 0 ? no : 1));
    ix->run_start = malloc(sizeof(int) * ((size_t)no + 1)); ix->run_of = malloc(sizeof(int) * (size_t)(no > 0 ? no : 1));
    for (int i = 0; i < no; ++i) { ++ix->cam_start[p->obs_cam[i] + 1]; ++ix->pt_start[p->obs_pt[i] + 1]; }
    for (int c = 0; c < nc; ++c) ix->cam_start[c + 1] += ix->cam_start[c];
    for (int j = 0; j < np; ++j) ix->pt_start[j + 1] += ix->pt_start[j];
    int* fc = malloc(sizeof(int) * ((size_t)nc + 1)); int* fp = malloc(sizeof(int) * ((size_t)np + 1));
    memcpy(fc, ix->cam_start, sizeof(int) * ((size_t)nc + 1)); memcpy(fp, ix->pt_start, sizeof(int) * ((size_t)np + 1));
    int nr = 0;
    for (int i = 0; i < no; ++i) {
        ix->cam_obs[fc[p->obs_cam[i]]++] = i; ix->pt_obs[fp[p->obs_pt[i]]++] = i;
        if (i == 0 || p->obs_pt[i] != p->obs_pt[i - 1]) ix->run_start[nr++] = i;
        ix->run_of[i] = nr - 1;
    }
    ix->run_start[nr] = no; ix->n_runs = nr;
    free(fc); free(fp);
    return ix;
}

double orc_ba_evaluate(const orc_ba_problem* p, double* r, double* Jc, double* Jp) {
    double cost = 0;
#pragma omp parallel for schedule(static) reduction(+ : cost) num_threads(g_threads)
    for (int i = 0; i < p->n_obs; ++i) {
        const int c = p->obs_cam[i], j = p->obs_pt[i];
        const double* cam = &p->cams[c * 7];
        double ri[2];
        orc_reproj_residual(cam, cam + 4, &p->pts[j * 3], &p->obs_feat[i * 2], ri);
        if (r) { r[i * 2] = ri[0]; r[i * 2 + 1] = ri[1]; }
        cost += ri[0] * ri[0] + ri[1] * ri[1];
        if (Jc || Jp) {
            double jc[12], jp[6];
            orc_reproj_jacobian(cam, cam + 4, &p->pts[j * 3], jc, jp, 0);
            /* constant dofs: their Jacobian columns are dropped (test_ceres.h:127-130) */
            for (int d = 0; d < 6; ++d)
                if (cam_dof_fixed(p, c, d)) { jc[d] = 0; jc[6 + d] = 0; }
            if (pt_is_fixed(p, j)) memset(jp, 0, sizeof jp);
            if (Jc) memcpy(&Jc[(size_t)i * 12], jc, sizeof jc);
            if (Jp) memcpy(&Jp[(size_t)i * 6], jp, sizeof jp);
        }
    }
    return 0.5 * cost;
}

void orc_ba_normal_blocks(const orc_ba_problem* p, const double* r, const double* Jc,
                          const double* Jp, double* Hcc, double* gc, double* Hpp, double* gp) {
    const ba_index* ix = ba_index_get(p);
    /* one camera / one landmark per loop trip, its observations in ascending order: no write conflicts and a summation order
     * that does not depend on the thread count (the same order, hence the same bits, as the walk-everything form of rounds 1-5) */
#pragma omp parallel for schedule(dynamic, 4) num_threads(g_threads)
    for (int c = 0; c < p->n_cams; ++c) {
        double H[36] = {0}, g[6] = {0};
        for (int k = ix->cam_start[c]; k < ix->cam_start[c + 1]; ++k) {
            const int i = ix->cam_obs[k];
            const double* jc = &Jc[(size_t)i * 12];
            const double r0 = r[i * 2], r1 = r[i * 2 + 1];
            for (int a = 0; a < 6; ++a) {
                for (int b2 = 0; b2 < 6; ++b2) H[a * 6 + b2] += jc[a] * jc[b2] + jc[6 + a] * jc[6 + b2];
                g[a] += jc[a] * r0 + jc[6 + a] * r1;
            }
        }
        memcpy(&Hcc[c * 36], H, sizeof H); memcpy(&gc[c * 6], g, sizeof g);
    }
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int j = 0; j < p->n_pts; ++j) {
        double H[9] = {0}, g[3] = {0};
        for (int k = ix->pt_start[j]; k < ix->pt_start[j + 1]; ++k) {
            const int i = ix->pt_obs[k];
            const double* jp = &Jp[(size_t)i * 6];
            const double r0 = r[i * 2], r1 = r[i * 2 + 1];
            for (int a = 0; a < 3; ++a) {
                for (int b2 = 0; b2 < 3; ++b2) H[a * 3 + b2] += jp[a] * jp[b2] + jp[3 + a] * jp[3 + b2];
                g[a] += jp[a] * r0 + jp[3 + a] * r1;
            }
        }
        memcpy(&Hpp[j * 9], H, sizeof H); memcpy(&gp[j * 3], g, sizeof g);
    }
}

static int inv3_sym(const double A[9], double Ai[9]) {
    const double a = A[0], b = A[1], c = A[2], d = A[4], e = A[5], f = A[8];
    const double C00 = d * f - e * e, C01 = c * e - b * f, C02 = b * e - c * d;
    const double det = a * C00 + b * C01 + c * C02;
    if (!(det > 0) || !isfinite(det)) return 1;
    const double inv = 1.0 / det;
    Ai[0] = C00 * inv; Ai[1] = C01 * inv; Ai[2] = C02 * inv;
    Ai[3] = Ai[1]; Ai[4] = (a * f - c * c) * inv; Ai[5] = (b * c - a * e) * inv;
    Ai[6] = Ai[2]; Ai[7] = Ai[5]; Ai[8] = (a * d - b * b) * inv;
    return 0;
}

void orc_ba_reduced_system(const orc_ba_problem* p, const double* Jc, const double* Jp,
                           const double* r, const double* dc, const double* dp,
                           int pt_begin, int pt_end, double* S, double* rhs) {
    const int n = 6 * p->n_cams;
    const ba_index* ix = ba_index_get(p);
    /* (a) per run of observations of one landmark: the damped inverse landmark block and the landmark's gradient, once */
    double* HiAll = malloc(sizeof(double) * 9 * (size_t)(ix->n_runs > 0 ? ix->n_runs : 1));
    double* gpAll = malloc(sizeof(double) * 3 * (size_t)(ix->n_runs > 0 ? ix->n_runs : 1));
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int q = 0; q < ix->n_runs; ++q) {
        const int i0 = ix->run_start[q], i1 = ix->run_start[q + 1], j = p->obs_pt[i0];
        double* Hi = &HiAll[(size_t)q * 9];
        double* gpv = &gpAll[(size_t)q * 3];
        if (j < pt_begin || j >= pt_end) { memset(Hi, 0, sizeof(double) * 9); gpv[0] = gpv[1] = gpv[2] = 0; continue; }
        double Hpp[9] = {0};
        gpv[0] = gpv[1] = gpv[2] = 0;
        for (int i = i0; i < i1; ++i) {
            const double* jp = &Jp[(size_t)i * 6];
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < 3; ++b) Hpp[a * 3 + b] += jp[a] * jp[b] + jp[3 + a] * jp[3 + b];
                gpv[a] += jp[a] * r[i * 2] + jp[3 + a] * r[i * 2 + 1];
            }
        }
        if (!pt_is_fixed(p, j)) {
            Hpp[0] += dp[j * 3]; Hpp[4] += dp[j * 3 + 1]; Hpp[8] += dp[j * 3 + 2];
            if (inv3_sym(Hpp, Hi)) memset(Hi, 0, sizeof(double) * 9);
        } else {
            memset(Hi, 0, sizeof(double) * 9);
        }
    }
    /* (b) one camera row per loop trip: the thread zeroes the camera's six rows of S and adds the camera's observations in
     * ascending order (landmark-major observations: landmark by landmark, as the walk of rounds 1-5 did): no write conflicts,
     * thread-count independent sums */
#pragma omp parallel for schedule(dynamic, 2) num_threads(g_threads)
    for (int c = 0; c < p->n_cams; ++c) {
        memset(&S[(size_t)(c * 6) * n], 0, sizeof(double) * 6 * (size_t)n);
        for (int a = 0; a < 6; ++a) rhs[c * 6 + a] = 0.0;
        for (int kk = ix->cam_start[c]; kk < ix->cam_start[c + 1]; ++kk) {
            const int i = ix->cam_obs[kk];
            const int j = p->obs_pt[i];
            if (j < pt_begin || j >= pt_end) continue;
            const int q = ix->run_of[i];
            const int i0 = ix->run_start[q], i1 = ix->run_start[q + 1];
            const double* Hi = &HiAll[(size_t)q * 9];
            const double* gpv = &gpAll[(size_t)q * 3];
            const int fixed = pt_is_fixed(p, j);
            const double* jc = &Jc[(size_t)i * 12];
            const double* jp = &Jp[(size_t)i * 6];
            const double r0 = r[i * 2], r1 = r[i * 2 + 1];
            /* camera diagonal block + gradient */
            for (int a = 0; a < 6; ++a) {
                for (int b = 0; b <= a; ++b)
                    S[(size_t)(c * 6 + a) * n + c * 6 + b] += jc[a] * jc[b] + jc[6 + a] * jc[6 + b];
                rhs[c * 6 + a] -= jc[a] * r0 + jc[6 + a] * r1;
            }
            if (fixed) continue;
            /* W = Jc^T Jp (6x3); E = W Hpp^-1 */
            double W[18], E[18];
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 3; ++b) W[a * 3 + b] = jc[a] * jp[b] + jc[6 + a] * jp[3 + b];
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 3; ++b)
                    E[a * 3 + b] = W[a * 3] * Hi[b] + W[a * 3 + 1] * Hi[3 + b] + W[a * 3 + 2] * Hi[6 + b];
            for (int a = 0; a < 6; ++a)
                rhs[c * 6 + a] += E[a * 3] * gpv[0] + E[a * 3 + 1] * gpv[1] + E[a * 3 + 2] * gpv[2];
            for (int l = i0; l < i1; ++l) {
                const int c2 = p->obs_cam[l];
                if (c2 > c) continue;   /* lower triangle: block (c, c2), c2 <= c */
                const double* jc2 = &Jc[(size_t)l * 12];
                const double* jp2 = &Jp[(size_t)l * 6];
                for (int a = 0; a < 6; ++a)
                    for (int b = 0; b < 6; ++b) {
                        if (c2 == c && b > a) continue;
                        /* (E W2^T)_{ab} = sum_k E[a][k] W2[b][k] */
                        double s2 = 0;
                        for (int k = 0; k < 3; ++k)
                            s2 += E[a * 3 + k] * (jc2[b] * jp2[k] + jc2[6 + b] * jp2[3 + k]);
                        S[(size_t)(c * 6 + a) * n + c2 * 6 + b] -= s2;
                    }
            }
        }
    }
    free(HiAll); free(gpAll);
    /* damping + fixed dofs (only when this call owns the whole landmark range start:
     * the diagonal terms must be added exactly once across shards -> shard with pt_begin==0) */
    if (pt_begin == 0) {
        for (int c = 0; c < p->n_cams; ++c)
            for (int a = 0; a < 6; ++a) {
                const size_t d = (size_t)(c * 6 + a) * n + c * 6 + a;
                if (cam_dof_fixed(p, c, a)) { S[d] += 1.0; }
                else S[d] += dc[c * 6 + a];
            }
    }
}

typedef struct {
    double *r, *Jc, *Jp, *Hcc, *gc, *Hpp, *gp, *dc, *dp, *scale_c, *scale_p, *S, *rhs, *dxc, *dxp;
    double *cams_new, *pts_new;
} ba_ws;

static void ba_backsub(const orc_ba_problem* p, const ba_ws* w) {
    const ba_index* ix = ba_index_get(p);
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int q = 0; q < ix->n_runs; ++q) {
        const int i0 = ix->run_start[q], i1 = ix->run_start[q + 1], j = p->obs_pt[i0];
        double* dx = &w->dxp[j * 3];
        if (pt_is_fixed(p, j)) { dx[0] = dx[1] = dx[2] = 0; continue; }
        double H[9], Hi[9], v[3];
        memcpy(H, &w->Hpp[j * 9], sizeof H);
        H[0] += w->dp[j * 3]; H[4] += w->dp[j * 3 + 1]; H[8] += w->dp[j * 3 + 2];
        if (inv3_sym(H, Hi)) memset(Hi, 0, sizeof Hi);
        v[0] = -w->gp[j * 3]; v[1] = -w->gp[j * 3 + 1]; v[2] = -w->gp[j * 3 + 2];
        for (int i = i0; i < i1; ++i) {
            const int c = p->obs_cam[i];
            const double* jc = &w->Jc[(size_t)i * 12];
            const double* jp = &w->Jp[(size_t)i * 6];
            /* W^T dc = Jp^T (Jc dc) */
            double m0 = 0, m1 = 0;
            for (int a = 0; a < 6; ++a) { m0 += jc[a] * w->dxc[c * 6 + a]; m1 += jc[6 + a] * w->dxc[c * 6 + a]; }
            for (int b = 0; b < 3; ++b) v[b] -= jp[b] * m0 + jp[3 + b] * m1;
        }
        for (int a = 0; a < 3; ++a) dx[a] = Hi[a * 3] * v[0] + Hi[a * 3 + 1] * v[1] + Hi[a * 3 + 2] * v[2];
    }
}

static void ba_apply(const orc_ba_problem* p, const double* dxc, const double* dxp,
                     double* cams_new, double* pts_new) {
    for (int c = 0; c < p->n_cams; ++c) {
        orc_so3_plus(&p->cams[c * 7], &dxc[c * 6], &cams_new[c * 7]);
        for (int a = 0; a < 3; ++a) cams_new[c * 7 + 4 + a] = p->cams[c * 7 + 4 + a] + dxc[c * 6 + 3 + a];
    }
    for (int j = 0; j < p->n_pts * 3; ++j) pts_new[j] = p->pts[j] + dxp[j];
}

static int cam_rot_active(const orc_ba_problem* p, int c) {
    return !(cam_dof_fixed(p, c, 0) && cam_dof_fixed(p, c, 1) && cam_dof_fixed(p, c, 2));
}
static int cam_pos_active(const orc_ba_problem* p, int c) {
    return !(cam_dof_fixed(p, c, 3) && cam_dof_fixed(p, c, 4) && cam_dof_fixed(p, c, 5));
}

static double ba_x_norm2(const orc_ba_problem* p, const double* cams, const double* pts,
                         const double* cams0, const double* pts0) {
    /* squared norm of (x) or (x - x0) over the non-constant parameter blocks */
    double s = 0;
    for (int c = 0; c < p->n_cams; ++c) {
        if (cam_rot_active(p, c))
            for (int a = 0; a < 4; ++a) {
                const double d = cams[c * 7 + a] - (cams0 ? cams0[c * 7 + a] : 0.0);
                s += d * d;
            }
        if (cam_pos_active(p, c))
            for (int a = 4; a < 7; ++a) {
                const double d = cams[c * 7 + a] - (cams0 ? cams0[c * 7 + a] : 0.0);
                s += d * d;
            }
    }
    for (int j = 0; j < p->n_pts; ++j) {
        if (pt_is_fixed(p, j)) continue;
        for (int a = 0; a < 3; ++a) {
            const double d = pts[j * 3 + a] - (pts0 ? pts0[j * 3 + a] : 0.0);
            s += d * d;
        }
    }
    return s;
}

int orc_ba_solve(orc_ba_problem* p, const orc_lm_options* opt, orc_lm_summary* sum, double* trace) {
    const int nc = p->n_cams, np = p->n_pts, no = p->n_obs, n = 6 * nc;
    const int nt = opt->num_threads > 0 ? opt->num_threads : 1;
    g_threads = nt;
    ba_ws w;
    w.r = malloc(sizeof(double) * 2 * no);
    w.Jc = malloc(sizeof(double) * 12 * (size_t)no);
    w.Jp = malloc(sizeof(double) * 6 * (size_t)no);
    w.Hcc = malloc(sizeof(double) * 36 * nc);
    w.gc = malloc(sizeof(double) * 6 * nc);
    w.Hpp = malloc(sizeof(double) * 9 * np);
    w.gp = malloc(sizeof(double) * 3 * np);
    w.dc = calloc(6 * nc, sizeof(double));
    w.dp = calloc(3 * np, sizeof(double));
    w.scale_c = malloc(sizeof(double) * 6 * nc);
    w.scale_p = malloc(sizeof(double) * 3 * np);
    w.S = malloc(sizeof(double) * (size_t)n * n);
    w.rhs = malloc(sizeof(double) * n);
    w.dxc = calloc(6 * nc, sizeof(double));
    w.dxp = calloc(3 * np, sizeof(double));
    w.cams_new = malloc(sizeof(double) * 7 * nc);
    w.pts_new = malloc(sizeof(double) * 3 * np);
    memset(sum, 0, sizeof *sum);
    const double t_start = now_s();
    double t0 = now_s();

    double cost = orc_ba_evaluate(p, w.r, w.Jc, w.Jp);
    orc_ba_normal_blocks(p, w.r, w.Jc, w.Jp, w.Hcc, w.gc, w.Hpp, w.gp);
    sum->seconds_linearize += now_s() - t0;
    sum->initial_cost = cost;

    /* Jacobi scaling, computed once from the initial Jacobian (Ceres jacobi_scaling) */
    for (int i = 0; i < 6 * nc; ++i) {
        const double h = w.Hcc[(i / 6) * 36 + (i % 6) * 7];
        w.scale_c[i] = opt->jacobi_scaling ? 1.0 / (1.0 + sqrt(h)) : 1.0;
    }
    for (int i = 0; i < 3 * np; ++i) {
        const double h = w.Hpp[(i / 3) * 9 + (i % 3) * 4];
        w.scale_p[i] = opt->jacobi_scaling ? 1.0 / (1.0 + sqrt(h)) : 1.0;
    }

    double gmax = 0;
    for (int i = 0; i < 6 * nc; ++i) if (fabs(w.gc[i]) > gmax) gmax = fabs(w.gc[i]);
    for (int i = 0; i < 3 * np; ++i) if (fabs(w.gp[i]) > gmax) gmax = fabs(w.gp[i]);

    double radius = opt->initial_trust_region_radius;
    double decrease_factor = 2.0;
    double x_norm = sqrt(ba_x_norm2(p, p->cams, p->pts, NULL, NULL));
    int iter = 0;
    if (trace) {
        memset(trace, 0, sizeof(double) * ORC_TRACE_COLS);
        trace[0] = cost; trace[2] = gmax; trace[5] = radius; trace[6] = 1;
    }
    sum->termination_type = ORC_NO_CONVERGENCE;
    sum->termination_reason = ORC_TERM_MAX_ITER;
    const int fixed = opt->fixed_iterations;
    const int max_iter = fixed > 0 ? fixed : opt->max_num_iterations;

    /* Ceres: a residual block that returns a non-finite value fails its evaluation, and a failed evaluation of the START point
     * ends the solve as FAILURE before any step ("Initial residual and Jacobian evaluation failed", TrustRegionMinimizer::Init);
     * at a trial point the same failure is only an unsuccessful step -- the rho test below rejects a non-finite cost */
    if (!isfinite(cost)) {
        sum->termination_type = ORC_FAILURE; sum->termination_reason = ORC_TERM_SOLVER_FAIL;
        goto done;
    }
    if (!fixed && gmax <= opt->gradient_tolerance) {
        sum->termination_type = ORC_CONVERGENCE; sum->termination_reason = ORC_TERM_GRADIENT;
        goto done;
    }

    while (1) {
        if (iter >= max_iter) {
            sum->termination_type = fixed ? ORC_CONVERGENCE : ORC_NO_CONVERGENCE;
            sum->termination_reason = fixed ? ORC_TERM_FIXED : ORC_TERM_MAX_ITER;
            break;
        }
        if (!fixed && radius < opt->min_trust_region_radius) {
            sum->termination_type = ORC_CONVERGENCE; sum->termination_reason = ORC_TERM_MIN_RADIUS;
            break;
        }
        ++iter;
        /* LM diagonal: clamp(diag(Js^T Js)) / radius, mapped back to unscaled coordinates */
        for (int i = 0; i < 6 * nc; ++i) {
            const double s2 = w.scale_c[i] * w.scale_c[i];
            double d = w.Hcc[(i / 6) * 36 + (i % 6) * 7] * s2;
            d = fmin(fmax(d, opt->min_lm_diagonal), opt->max_lm_diagonal);
            w.dc[i] = d / radius / s2;
        }
        for (int i = 0; i < 3 * np; ++i) {
            const double s2 = w.scale_p[i] * w.scale_p[i];
            double d = w.Hpp[(i / 3) * 9 + (i % 3) * 4] * s2;
            d = fmin(fmax(d, opt->min_lm_diagonal), opt->max_lm_diagonal);
            w.dp[i] = d / radius / s2;
        }
        t0 = now_s();
        orc_ba_reduced_system(p, w.Jc, w.Jp, w.r, w.dc, w.dp, 0, np, w.S, w.rhs);
        sum->seconds_schur += now_s() - t0;
        t0 = now_s();
        int bad;
        if (g_dense_solver) {
            memcpy(w.dxc, w.rhs, sizeof(double) * n);
            bad = g_dense_solver(w.S, n, w.dxc);
        } else {
            bad = orc_cholesky_lower(w.S, n, nt);
        }
        int step_ok = (bad == 0);
        if (step_ok) {
            if (!g_dense_solver) {
                memcpy(w.dxc, w.rhs, sizeof(double) * n);
                chol_solve_big(w.S, n, w.dxc);
            }
            for (int c = 0; c < nc; ++c)
                for (int a = 0; a < 6; ++a)
                    if (cam_dof_fixed(p, c, a)) w.dxc[c * 6 + a] = 0;
        }
        sum->seconds_solve += now_s() - t0;
        double model_change = 0, new_cost = 0, step_norm = 0, rho = 0;
        if (step_ok) {
            t0 = now_s();
            ba_backsub(p, &w);
            sum->seconds_backsub += now_s() - t0;
            /* model_cost_change = -sum m.(r + m/2), m = J delta (Ceres trust_region_minimizer) */
#pragma omp parallel for schedule(static) reduction(+ : model_change) num_threads(g_threads)
            for (int i = 0; i < no; ++i) {
                const int c = p->obs_cam[i], j = p->obs_pt[i];
                const double* jc = &w.Jc[(size_t)i * 12];
                const double* jp = &w.Jp[(size_t)i * 6];
                double m0 = 0, m1 = 0;
                for (int a = 0; a < 6; ++a) { m0 += jc[a] * w.dxc[c * 6 + a]; m1 += jc[6 + a] * w.dxc[c * 6 + a]; }
                for (int a = 0; a < 3; ++a) { m0 += jp[a] * w.dxp[j * 3 + a]; m1 += jp[3 + a] * w.dxp[j * 3 + a]; }
                model_change -= m0 * (w.r[i * 2] + 0.5 * m0) + m1 * (w.r[i * 2 + 1] + 0.5 * m1);
            }
            if (!(model_change > 0) || !isfinite(model_change)) step_ok = 0;
        }
        int accepted = 0;
        if (step_ok) {
            ba_apply(p, w.dxc, w.dxp, w.cams_new, w.pts_new);
            t0 = now_s();
            orc_ba_problem q = *p;
            q.cams = w.cams_new; q.pts = w.pts_new;
            new_cost = orc_ba_evaluate(&q, NULL, NULL, NULL);
            sum->seconds_cost += now_s() - t0;
            step_norm = sqrt(ba_x_norm2(p, w.cams_new, w.pts_new, p->cams, p->pts));
            const double cost_change = cost - new_cost;
            rho = cost_change / model_change;
            if (trace) {
                double* tr = &trace[iter * ORC_TRACE_COLS];
                tr[0] = new_cost; tr[1] = cost_change; tr[3] = step_norm; tr[4] = rho;
            }
            if (!fixed) {
                if (step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) {
                    sum->termination_type = ORC_CONVERGENCE; sum->termination_reason = ORC_TERM_PARAMETER;
                    if (trace) { trace[iter * ORC_TRACE_COLS + 5] = radius; trace[iter * ORC_TRACE_COLS + 2] = gmax; }
                    break;
                }
                if (fabs(cost_change) <= opt->function_tolerance * cost) {
                    /* the step is taken if it is a decrease before convergence is reported -- or not at all
                     * (function_tolerance_takes_step = 0: the other reading of Ceres, oracle.h) */
                    if (opt->function_tolerance_takes_step && rho > opt->min_relative_decrease) {
                        memcpy(p->cams, w.cams_new, sizeof(double) * 7 * nc);
                        memcpy(p->pts, w.pts_new, sizeof(double) * 3 * np);
                        cost = new_cost; ++sum->num_successful_steps;
                        if (trace) trace[iter * ORC_TRACE_COLS + 6] = 1;
                    }
                    sum->termination_type = ORC_CONVERGENCE; sum->termination_reason = ORC_TERM_FUNCTION;
                    if (trace) { trace[iter * ORC_TRACE_COLS + 5] = radius; trace[iter * ORC_TRACE_COLS + 2] = gmax; }
                    break;
                }
            }
            accepted = rho > opt->min_relative_decrease;
        }
        if (accepted) {
            memcpy(p->cams, w.cams_new, sizeof(double) * 7 * nc);
            memcpy(p->pts, w.pts_new, sizeof(double) * 3 * np);
            cost = new_cost;
            x_norm = sqrt(ba_x_norm2(p, p->cams, p->pts, NULL, NULL));
            ++sum->num_successful_steps;
            t0 = now_s();
            orc_ba_evaluate(p, w.r, w.Jc, w.Jp);
            orc_ba_normal_blocks(p, w.r, w.Jc, w.Jp, w.Hcc, w.gc, w.Hpp, w.gp);
            sum->seconds_linearize += now_s() - t0;
            gmax = 0;
            for (int i = 0; i < 6 * nc; ++i) if (fabs(w.gc[i]) > gmax) gmax = fabs(w.gc[i]);
            for (int i = 0; i < 3 * np; ++i) if (fabs(w.gp[i]) > gmax) gmax = fabs(w.gp[i]);
            /* LevenbergMarquardtStrategy::StepAccepted */
            const double t = 2.0 * rho - 1.0;
            radius = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
            radius = fmin(opt->max_trust_region_radius, radius);
            decrease_factor = 2.0;
        } else {
            ++sum->num_unsuccessful_steps;
            radius = radius / decrease_factor;
            decrease_factor *= 2.0;
            if (fixed) {   /* fixed-work mode re-linearises every iteration */
                t0 = now_s();
                orc_ba_evaluate(p, w.r, w.Jc, w.Jp);
                orc_ba_normal_blocks(p, w.r, w.Jc, w.Jp, w.Hcc, w.gc, w.Hpp, w.gp);
                sum->seconds_linearize += now_s() - t0;
            }
        }
        if (trace) {
            double* tr = &trace[iter * ORC_TRACE_COLS];
            if (!step_ok) { tr[0] = cost; tr[1] = 0; tr[3] = 0; tr[4] = 0; }
            tr[2] = gmax; tr[5] = radius; tr[6] = accepted;
        }
        if (accepted && !fixed && gmax <= opt->gradient_tolerance) {
            sum->termination_type = ORC_CONVERGENCE; sum->termination_reason = ORC_TERM_GRADIENT;
            break;
        }
    }
done:
    sum->num_iterations = iter;
    sum->final_cost = cost;
    sum->final_radius = radius;
    sum->final_gradient_max_norm = gmax;
    sum->seconds_total = now_s() - t_start;
    free(w.r); free(w.Jc); free(w.Jp); free(w.Hcc); free(w.gc); free(w.Hpp); free(w.gp);
    free(w.dc); free(w.dp); free(w.scale_c); free(w.scale_p); free(w.S); free(w.rhs);
    free(w.dxc); free(w.dxp); free(w.cams_new); free(w.pts_new);
    return sum->termination_type;
}

/* sim_data.cpp:299-311: each landmark refined alone with cameras held fixed.  The reference
 * runs a default-option Ceres solve of Triangulation (sim_data.h:165-194, r = feature - proj,
 * same |r| as the BA residual); restated as damped Gauss-Newton on the 3x3 system. */
void orc_ba_triangulate(orc_ba_problem* p, int max_iter) {
    int i0 = 0;
    while (i0 < p->n_obs) {
        const int j = p->obs_pt[i0];
        int i1 = i0;
        while (i1 < p->n_obs && p->obs_pt[i1] == j) ++i1;
        double* L = &p->pts[j * 3];
        double lambda = 1e-4, cost = 0;
        for (int i = i0; i < i1; ++i) {
            double r[2];
            const double* cam = &p->cams[p->obs_cam[i] * 7];
            orc_reproj_residual(cam, cam + 4, L, &p->obs_feat[i * 2], r);
            cost += r[0] * r[0] + r[1] * r[1];
        }
        for (int it = 0; it < max_iter; ++it) {
            double H[9] = {0}, g[3] = {0};
            for (int i = i0; i < i1; ++i) {
                double r[2], jp[6];
                const double* cam = &p->cams[p->obs_cam[i] * 7];
                orc_reproj_residual(cam, cam + 4, L, &p->obs_feat[i * 2], r);
                orc_reproj_jacobian(cam, cam + 4, L, NULL, jp, 0);
                for (int a = 0; a < 3; ++a) {
                    for (int b = 0; b < 3; ++b) H[a * 3 + b] += jp[a] * jp[b] + jp[3 + a] * jp[3 + b];
                    g[a] -= jp[a] * r[0] + jp[3 + a] * r[1];
                }
            }
            double Hd[9], d[3], Ln[3];
            memcpy(Hd, H, sizeof H);
            Hd[0] += lambda * (H[0] + 1e-12); Hd[4] += lambda * (H[4] + 1e-12); Hd[8] += lambda * (H[8] + 1e-12);
            if (solve3(Hd, g, d)) break;
            for (int a = 0; a < 3; ++a) Ln[a] = L[a] + d[a];
            double nc = 0;
            for (int i = i0; i < i1; ++i) {
                double r[2];
                const double* cam = &p->cams[p->obs_cam[i] * 7];
                orc_reproj_residual(cam, cam + 4, Ln, &p->obs_feat[i * 2], r);
                nc += r[0] * r[0] + r[1] * r[1];
            }
            if (nc < cost && isfinite(nc)) {
                const double dn = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                L[0] = Ln[0]; L[1] = Ln[1]; L[2] = Ln[2];
                const double rel = (cost - nc) / (cost + 1e-300);
                cost = nc; lambda = fmax(lambda * 0.1, 1e-12);
                if (dn < 1e-12 || rel < 1e-14) break;
            } else {
                lambda *= 10.0;
                if (lambda > 1e12) break;
            }
        }
        i0 = i1;
    }
}

/* ======================================================================================
 * generic dense LM (same trust-region logic, dense normal equations + Cholesky).
 * Ceres DENSE_QR (solver.hpp:282) solves the same damped least squares by QR; the step is
 * identical in exact arithmetic.
 * ==================================================================================== */
int orc_dense_lm(orc_residual_fn fn, orc_plus_fn plus, void* user, int n_params, int n_local,
                 int n_res, double* x, const double* lower, const double* upper,
                 const orc_lm_options* opt, orc_lm_summary* sum, double* trace) {
    const int n = n_local;
    double* r = malloc(sizeof(double) * n_res);
    double* J = malloc(sizeof(double) * (size_t)n_res * n);
    double* H = malloc(sizeof(double) * n * n);
    double* Hd = malloc(sizeof(double) * n * n);
    double* g = malloc(sizeof(double) * n);
    double* dx = malloc(sizeof(double) * n);
    double* scale = malloc(sizeof(double) * n);
    double* xn = malloc(sizeof(double) * n_params);
    double* rn = malloc(sizeof(double) * n_res);
    memset(sum, 0, sizeof *sum);
    const double t_start = now_s();
    const int bounded = (lower != NULL) || (upper != NULL);
    int rc = ORC_NO_CONVERGENCE;
    sum->termination_reason = ORC_TERM_MAX_ITER;

#define DENSE_LINEARIZE()                                                                   \
    do {                                                                                    \
        for (int a = 0; a < n; ++a) {                                                       \
            g[a] = 0;                                                                       \
            for (int b = 0; b < n; ++b) H[a * n + b] = 0;                                   \
        }                                                                                   \
        for (int i = 0; i < n_res; ++i)                                                     \
            for (int a = 0; a < n; ++a) {                                                   \
                const double ja = J[(size_t)i * n + a];                                     \
                g[a] += ja * r[i];                                                          \
                for (int b = 0; b <= a; ++b) H[a * n + b] += ja * J[(size_t)i * n + b];     \
            }                                                                               \
        for (int a = 0; a < n; ++a)                                                         \
            for (int b = a + 1; b < n; ++b) H[a * n + b] = H[b * n + a];                    \
    } while (0)

    if (fn(user, x, r, J)) { rc = ORC_FAILURE; sum->termination_reason = ORC_TERM_SOLVER_FAIL; goto out; }
    double cost = 0;
    for (int i = 0; i < n_res; ++i) cost += r[i] * r[i];
    cost *= 0.5;
    sum->initial_cost = cost;
    if (!isfinite(cost)) {       /* (Ceres: initial evaluation failed -- see orc_ba_solve) */
        rc = ORC_FAILURE; sum->termination_reason = ORC_TERM_SOLVER_FAIL; sum->final_cost = cost;
        goto out;
    }
    DENSE_LINEARIZE();
    for (int a = 0; a < n; ++a) scale[a] = opt->jacobi_scaling ? 1.0 / (1.0 + sqrt(H[a * n + a])) : 1.0;

    double gmax;
#define DENSE_GMAX()                                                                        \
    do {                                                                                    \
        gmax = 0;                                                                           \
        if (!bounded) {                                                                     \
            for (int a = 0; a < n; ++a) if (fabs(g[a]) > gmax) gmax = fabs(g[a]);           \
        } else { /* |x - Proj(x - g)|_inf (Ceres projected gradient for bounds) */          \
            for (int a = 0; a < n; ++a) {                                                   \
                double y = x[a] - g[a];                                                     \
                if (lower && y < lower[a]) y = lower[a];                                    \
                if (upper && y > upper[a]) y = upper[a];                                    \
                if (fabs(x[a] - y) > gmax) gmax = fabs(x[a] - y);                           \
            }                                                                               \
        }                                                                                   \
    } while (0)
    DENSE_GMAX();
    double radius = opt->initial_trust_region_radius, decrease_factor = 2.0;
    double x_norm = 0;
    for (int a = 0; a < n_params; ++a) x_norm += x[a] * x[a];
    x_norm = sqrt(x_norm);
    int iter = 0;
    if (trace) { memset(trace, 0, sizeof(double) * ORC_TRACE_COLS); trace[0] = cost; trace[2] = gmax; trace[5] = radius; trace[6] = 1; }
    if (gmax <= opt->gradient_tolerance) { rc = ORC_CONVERGENCE; sum->termination_reason = ORC_TERM_GRADIENT; goto fin; }

    while (1) {
        if (iter >= opt->max_num_iterations) { rc = ORC_NO_CONVERGENCE; sum->termination_reason = ORC_TERM_MAX_ITER; break; }
        if (radius < opt->min_trust_region_radius) { rc = ORC_CONVERGENCE; sum->termination_reason = ORC_TERM_MIN_RADIUS; break; }
        ++iter;
        memcpy(Hd, H, sizeof(double) * n * n);
        for (int a = 0; a < n; ++a) {
            const double s2 = scale[a] * scale[a];
            double d = fmin(fmax(H[a * n + a] * s2, opt->min_lm_diagonal), opt->max_lm_diagonal);
            Hd[a * n + a] += d / radius / s2;
        }
        int ok = (orc_cholesky_lower(Hd, n, 1) == 0);
        double model_change = 0, new_cost = 0, step_norm = 0, rho = 0;
        if (ok) {
            for (int a = 0; a < n; ++a) dx[a] = -g[a];
            orc_cholesky_solve(Hd, n, dx);
            for (int i = 0; i < n_res; ++i) {
                double m = 0;
                for (int a = 0; a < n; ++a) m += J[(size_t)i * n + a] * dx[a];
                model_change -= m * (r[i] + 0.5 * m);
            }
            if (!(model_change > 0) || !isfinite(model_change)) ok = 0;
        }
        int accepted = 0;
        if (ok) {
            if (plus) plus(user, x, dx, xn);
            else for (int a = 0; a < n_params; ++a) xn[a] = x[a] + dx[a];
            if (bounded)   /* Ceres projects the candidate onto the box */
                for (int a = 0; a < n_params; ++a) {
                    if (lower && xn[a] < lower[a]) xn[a] = lower[a];
                    if (upper && xn[a] > upper[a]) xn[a] = upper[a];
                }
            if (fn(user, xn, rn, NULL)) ok = 0;
        }
        if (ok) {
            for (int i = 0; i < n_res; ++i) new_cost += rn[i] * rn[i];
            new_cost *= 0.5;
            for (int a = 0; a < n_params; ++a) step_norm += (xn[a] - x[a]) * (xn[a] - x[a]);
            step_norm = sqrt(step_norm);
            const double cost_change = cost - new_cost;
            rho = cost_change / model_change;
            if (trace) { double* tr = &trace[iter * ORC_TRACE_COLS]; tr[0] = new_cost; tr[1] = cost_change; tr[3] = step_norm; tr[4] = rho; }
            if (step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) {
                rc = ORC_CONVERGENCE; sum->termination_reason = ORC_TERM_PARAMETER;
                if (trace) { trace[iter * ORC_TRACE_COLS + 5] = radius; trace[iter * ORC_TRACE_COLS + 2] = gmax; }
                break;
            }
            if (fabs(cost_change) <= opt->function_tolerance * cost) {
                if (opt->function_tolerance_takes_step && rho > opt->min_relative_decrease) {
                    memcpy(x, xn, sizeof(double) * n_params); cost = new_cost; ++sum->num_successful_steps;
                    if (trace) trace[iter * ORC_TRACE_COLS + 6] = 1;
                }
                rc = ORC_CONVERGENCE; sum->termination_reason = ORC_TERM_FUNCTION;
                if (trace) { trace[iter * ORC_TRACE_COLS + 5] = radius; trace[iter * ORC_TRACE_COLS + 2] = gmax; }
                break;
            }
            accepted = rho > opt->min_relative_decrease;
        }
        if (accepted) {
            memcpy(x, xn, sizeof(double) * n_params);
            cost = new_cost;
            x_norm = 0;
            for (int a = 0; a < n_params; ++a) x_norm += x[a] * x[a];
            x_norm = sqrt(x_norm);
            ++sum->num_successful_steps;
            if (fn(user, x, r, J)) { rc = ORC_FAILURE; sum->termination_reason = ORC_TERM_SOLVER_FAIL; break; }
            DENSE_LINEARIZE();
            DENSE_GMAX();
            const double t = 2.0 * rho - 1.0;
            radius = fmin(opt->max_trust_region_radius, radius / fmax(1.0 / 3.0, 1.0 - t * t * t));
            decrease_factor = 2.0;
        } else {
            ++sum->num_unsuccessful_steps;
            radius /= decrease_factor;
            decrease_factor *= 2.0;
        }
        if (trace) {
            double* tr = &trace[iter * ORC_TRACE_COLS];
            if (!ok) { tr[0] = cost; tr[1] = 0; tr[3] = 0; tr[4] = 0; }
            tr[2] = gmax; tr[5] = radius; tr[6] = accepted;
        }
        if (accepted && gmax <= opt->gradient_tolerance) { rc = ORC_CONVERGENCE; sum->termination_reason = ORC_TERM_GRADIENT; break; }
    }
fin:
    sum->num_iterations = iter;
    sum->final_cost = cost;
    sum->final_radius = radius;
    sum->final_gradient_max_norm = gmax;
out:
    sum->termination_type = rc;
    sum->seconds_total = now_s() - t_start;
    free(r); free(J); free(H); free(Hd); free(g); free(dx); free(scale); free(xn); free(rn);
    return rc;
}

/* ======================================================================================
 * st17: SelfGaussNewton (solver.hpp:387-462)
 * ==================================================================================== */
int orc_pnp_gauss_newton(int n, const double* pts_w, const double* feats, double q[4], double t[3],
                         int rot_mode, int max_iter, double* change_trace) {
    int i = 0;
    for (; i != max_iter; ++i) {                       /* solver.hpp:401 (max_iter = 10) */
        double H[36] = {0}, g[6] = {0};
        for (int k = 0; k < n; ++k) {                  /* solver.hpp:405-436 */
            double r[2], jc[12];
            orc_reproj_residual(q, t, &pts_w[k * 3], &feats[k * 2], r);
            orc_reproj_jacobian(q, t, &pts_w[k * 3], jc, NULL, rot_mode);
            for (int a = 0; a < 6; ++a) {
                for (int b = 0; b < 6; ++b) H[a * 6 + b] += jc[a] * jc[b] + jc[6 + a] * jc[6 + b];
                g[a] -= jc[a] * r[0] + jc[6 + a] * r[1];
            }
        }
        /* hMat.ldlt().solve(gMat), solver.hpp:438 (SPD here -> Cholesky gives the same x) */
        if (orc_cholesky_lower(H, 6, 1)) return -1;
        orc_cholesky_solve(H, 6, g);
        double qn[4];
        orc_so3_plus(q, g, qn);                         /* solver.hpp:442 */
        memcpy(q, qn, sizeof qn);
        t[0] += g[3]; t[1] += g[4]; t[2] += g[5];       /* solver.hpp:443 */
        const double change = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]) +
                              sqrt(g[3] * g[3] + g[4] * g[4] + g[5] * g[5]);   /* :445 */
        if (change_trace) change_trace[i] = change;
        if (change < 1e-8) break;                       /* :452 */
    }
    return i;
}

/* ======================================================================================
 * st7: parabola (float arithmetic as in the reference, parabola.hpp:98-130)
 * ==================================================================================== */
static int solve3f(const float A[9], const float b[3], float x[3]) {
    double Ad[9], bd[3], xd[3];
    for (int i = 0; i < 9; ++i) Ad[i] = A[i];
    for (int i = 0; i < 3; ++i) bd[i] = b[i];
    if (solve3(Ad, bd, xd)) return 1;
    for (int i = 0; i < 3; ++i) x[i] = (float)xd[i];
    return 0;
}

void orc_parabola_least_square(int n, const float* xy, float abc[3]) {
    /* (B^T B)^-1 B^T l, parabola.hpp:98-108 */
    float N[9] = {0}, u[3] = {0};
    for (int i = 0; i < n; ++i) {
        const float x = xy[i * 2], y = xy[i * 2 + 1];
        const float B[3] = {x * x, x, 1.0f};
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) N[a * 3 + b] += B[a] * B[b];
            u[a] += B[a] * y;
        }
    }
    solve3f(N, u, abc);
}

int orc_parabola_gauss_newton(int n, const float* xy, int iters, float abc[3]) {
    float X[3] = {1.0f, 0.0f, 0.0f};                   /* parabola.hpp:112 */
    int i = 0;
    for (; i != iters; ++i) {
        float H[9] = {0}, g[3] = {0};
        for (int j = 0; j < n; ++j) {
            const float x = xy[j * 2], y = xy[j * 2 + 1];
            const float err = (X[0] * x * x + X[1] * x + X[2]) - y;   /* :118 */
            const float J[3] = {x * x, x, 1.0f};
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < 3; ++b) H[a * 3 + b] += J[a] * J[b];
                g[a] += -J[a] * err;
            }
        }
        float d[3];
        if (solve3f(H, g, d)) break;
        X[0] += d[0]; X[1] += d[1]; X[2] += d[2];
        if (sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) < 1e-6f) break;   /* :126 */
    }
    abc[0] = X[0]; abc[1] = X[1]; abc[2] = X[2];
    return i;
}

/* ======================================================================================
 * st6: planar point-to-point alignment with known correspondences -- Gauss-Newton on SE2 with a LEFT
 * multiplicative update, FLOAT arithmetic as the reference (st6-icp/src/include/icp.hpp:28-50).  The
 * reference holds the iterates of this loop after 1 and 2 iterations on a recorded 10-point pair
 * (st6-icp/log/binding/pc1_prime_{1,2}.csv): the only per-iteration trace of a manifold Gauss-Newton in the
 * repository.  T = {cos, sin, tx, ty}: p' = R p + t.
 * ==================================================================================== */
static void se2_exp_f(const float d[3], float T[4]) {
    /* Sophus::SE2f::exp: tangent (upsilon_x, upsilon_y, theta); translation = V(theta) upsilon */
    const float th = d[2];
    const float c = cosf(th), s = sinf(th);
    float sbt, omcbt;                                  /* sin(theta)/theta, (1 - cos(theta))/theta */
    if (fabsf(th) < 1e-5f) {                           /* Sophus: Constants<float>::epsilon() = 1e-5f */
        const float th2 = th * th;
        sbt = 1.0f - (1.0f / 6.0f) * th2;
        omcbt = 0.5f * th - (1.0f / 24.0f) * th * th2;
    } else {
        sbt = s / th;
        omcbt = (1.0f - c) / th;
    }
    T[0] = c; T[1] = s;
    T[2] = sbt * d[0] - omcbt * d[1];
    T[3] = omcbt * d[0] + sbt * d[1];
}

int orc_icp_se2_gauss_newton(int n, const float* pc1, const float* pc2, int iters, float T[4]) {
    T[0] = 1.0f; T[1] = 0.0f; T[2] = 0.0f; T[3] = 0.0f;      /* icp.hpp:30: identity */
    int it = 0;
    for (; it != iters; ++it) {
        float H[9] = {0}, g[3] = {0};
        for (int i = 0; i < n; ++i) {
            const float x = pc1[i * 2], y = pc1[i * 2 + 1];
            const float px = T[0] * x - T[1] * y + T[2], py = T[1] * x + T[0] * y + T[3];   /* :37 */
            const float ex = px - pc2[i * 2], ey = py - pc2[i * 2 + 1];                     /* :38 */
            const float J[2][3] = {{1.0f, 0.0f, -py}, {0.0f, 1.0f, px}};                    /* :40-41 */
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < 3; ++b) H[a * 3 + b] += J[0][a] * J[0][b] + J[1][a] * J[1][b];
                g[a] -= J[0][a] * ex + J[1][a] * ey;                                          /* :43 */
            }
        }
        float d[3];
        if (solve3f(H, g, d)) break;                                                          /* :45 (ldlt) */
        float E[4];
        se2_exp_f(d, E);
        /* T <- exp(delta) * T  (:46) */
        const float c = E[0] * T[0] - E[1] * T[1], s = E[1] * T[0] + E[0] * T[1];
        const float tx = E[0] * T[2] - E[1] * T[3] + E[2], ty = E[1] * T[2] + E[0] * T[3] + E[3];
        /* Sophus SO2 product: no square root -- the first-order factor 2 / (1 + |z|^2), applied only when |z|^2 != 1 */
        const float n2 = c * c + s * s;
        const float sc = (n2 != 1.0f) ? 2.0f / (1.0f + n2) : 1.0f;
        T[0] = c * sc; T[1] = s * sc; T[2] = tx; T[3] = ty;
    }
    return it;
}

/* ======================================================================================
 * st3: calibration (calib.cpp:247-262, 282-422)
 * ==================================================================================== */
double orc_calib_evaluate(int n_views, int n_corners, const double* params, const double* obj,
                          const double* img, double* e, double* Ji, double* Jx) {
    const double alpha = params[0], beta = params[1], u0 = params[2], v0 = params[3];
    const double k1 = params[4], k2 = params[5], k3 = params[6], p1 = params[7], p2 = params[8];
    double sse = 0;
    for (int v = 0; v < n_views; ++v) {
        double q[4], t[3], R[9];
        orc_se3_exp(&params[9 + v * 6], q, t);          /* calib.cpp:318 */
        orc_quat_to_rot(q, R);
        for (int c = 0; c < n_corners; ++c) {
            const size_t o = (size_t)v * n_corners + c;
            const double X = obj[o * 2], Y = obj[o * 2 + 1];          /* Z = 0, :322 */
            const double Xp = R[0] * X + R[1] * Y + t[0];
            const double Yp = R[3] * X + R[4] * Y + t[1];
            const double Zp = R[6] * X + R[7] * Y + t[2];
            const double xn = Xp / Zp, yn = Yp / Zp;                  /* :327 */
            const double r2 = xn * xn + yn * yn, r4 = r2 * r2, r6 = r4 * r2;
            const double rad = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
            /* distortNormPt :254-262 */
            const double xd = xn * rad + 2.0 * p1 * xn * yn + p2 * (r2 + 2.0 * xn * xn);
            const double yd = yn * rad + 2.0 * p2 * xn * yn + p1 * (r2 + 2.0 * yn * yn);
            const double u = alpha * xd + u0, vv = beta * yd + v0;    /* :247-252 */
            const double e0 = u - img[o * 2], e1 = vv - img[o * 2 + 1];   /* :334 */
            if (e) { e[o * 2] = e0; e[o * 2 + 1] = e1; }
            sse += e0 * e0 + e1 * e1;
            if (Ji) {
                double* J = &Ji[o * 18];   /* 2x9 row-major */
                /* intrinsics :337-339, distortion :342-348 (reference stores the transpose) */
                J[0] = xd; J[1] = 0;  J[2] = 1; J[3] = 0;
                J[9] = 0;  J[10] = yd; J[11] = 0; J[12] = 1;
                J[4] = alpha * xn * r2;  J[13] = beta * yn * r2;
                J[5] = alpha * xn * r4;  J[14] = beta * yn * r4;
                J[6] = alpha * xn * r6;  J[15] = beta * yn * r6;
                J[7] = 2.0 * alpha * xn * yn;           J[16] = beta * (r2 + 2.0 * yn * yn);
                J[8] = alpha * (r2 + 2.0 * xn * xn);    J[17] = 2.0 * beta * xn * yn;
            }
            if (Jx) {
                /* pd_pn :356-367 */
                const double dx = 2.0 * k1 * xn + 4.0 * k2 * r2 * xn + 6.0 * k3 * r4 * xn;
                const double dy = 2.0 * k1 * yn + 4.0 * k2 * r2 * yn + 6.0 * k3 * r4 * yn;
                const double d00 = rad + xn * dx + 2.0 * p1 * yn + 6.0 * p2 * xn;
                const double d01 = xn * dy + 2.0 * p1 * xn + 2.0 * p2 * yn;
                const double d10 = yn * dx + 2.0 * p1 * xn + 2.0 * p2 * yn;
                const double d11 = rad + yn * dy + 2.0 * p2 * xn + 6.0 * p1 * yn;
                /* pn_PPrime :369-376 */
                const double iz = 1.0 / Zp, iz2 = iz * iz;
                const double N[6] = {iz, 0, -Xp * iz2, 0, iz, -Yp * iz2};
                /* M = diag(alpha,beta) * pd_pn * pn_PPrime  (2x3) */
                double M[6];
                for (int b = 0; b < 3; ++b) {
                    M[b] = alpha * (d00 * N[b] + d01 * N[3 + b]);
                    M[3 + b] = beta * (d10 * N[b] + d11 * N[3 + b]);
                }
                /* PPrime_pos = [I | -hat(P')] :378-380 */
                const double P[3] = {Xp, Yp, Zp};
                double Hh[9];
                hat3(P, Hh);
                double* J = &Jx[o * 12];   /* 2x6 row-major */
                for (int a = 0; a < 2; ++a)
                    for (int b = 0; b < 3; ++b) {
                        J[a * 6 + b] = M[a * 3 + b];
                        J[a * 6 + 3 + b] = -(M[a * 3] * Hh[b] + M[a * 3 + 1] * Hh[3 + b] + M[a * 3 + 2] * Hh[6 + b]);
                    }
            }
        }
    }
    return sse;
}

int orc_calib_gauss_newton(int n_views, int n_corners, double* params, const double* obj,
                           const double* img, int max_iter, double* sse_trace) {
    const int n = 9 + 6 * n_views;
    const size_t no = (size_t)n_views * n_corners;
    double* e = malloc(sizeof(double) * 2 * no);
    double* Ji = malloc(sizeof(double) * 18 * no);
    double* Jx = malloc(sizeof(double) * 12 * no);
    double* H = malloc(sizeof(double) * n * n);
    double* g = malloc(sizeof(double) * n);
    int iter = 0;
    for (; iter != max_iter; ++iter) {                  /* calib.cpp:303 */
        const double sse = orc_calib_evaluate(n_views, n_corners, params, obj, img, e, Ji, Jx);
        if (sse_trace) sse_trace[iter] = sse;
        memset(H, 0, sizeof(double) * n * n);
        memset(g, 0, sizeof(double) * n);
        /* H += J J^T, g -= J e (:383-389) using the arrow structure: the full-width J of the
         * reference is zero outside [0,9) and the view's 6 columns, so the sums are identical */
        for (int v = 0; v < n_views; ++v)
            for (int c = 0; c < n_corners; ++c) {
                const size_t o = (size_t)v * n_corners + c;
                int idx[15];
                double j0[15], j1[15];
                for (int a = 0; a < 9; ++a) { idx[a] = a; j0[a] = Ji[o * 18 + a]; j1[a] = Ji[o * 18 + 9 + a]; }
                for (int a = 0; a < 6; ++a) { idx[9 + a] = 9 + v * 6 + a; j0[9 + a] = Jx[o * 12 + a]; j1[9 + a] = Jx[o * 12 + 6 + a]; }
                for (int a = 0; a < 15; ++a) {
                    g[idx[a]] -= j0[a] * e[o * 2] + j1[a] * e[o * 2 + 1];
                    for (int b = 0; b < 15; ++b) H[idx[a] * n + idx[b]] += j0[a] * j0[b] + j1[a] * j1[b];
                }
            }
        /* H.ldlt().solve(g) :393 */
        if (orc_cholesky_lower(H, n, 1)) break;
        orc_cholesky_solve(H, n, g);
        for (int a = 0; a < 9; ++a) params[a] += g[a];  /* :394 */
        double un = 0;
        for (int a = 0; a < n; ++a) un += g[a] * g[a];
        for (int v = 0; v < n_views; ++v) {             /* :397-402: exp(update) * exp(param) */
            double qa[4], ta[3], qb[4], tb[3], qc[4], tc[3], Ra[9];
            orc_se3_exp(&g[9 + v * 6], qa, ta);
            orc_se3_exp(&params[9 + v * 6], qb, tb);
            orc_quat_mul(qa, qb, qc);
            orc_quat_to_rot(qa, Ra);
            for (int a = 0; a < 3; ++a) tc[a] = Ra[a * 3] * tb[0] + Ra[a * 3 + 1] * tb[1] + Ra[a * 3 + 2] * tb[2] + ta[a];
            const double nn = sqrt(qc[0] * qc[0] + qc[1] * qc[1] + qc[2] * qc[2] + qc[3] * qc[3]);
            for (int a = 0; a < 4; ++a) qc[a] /= nn;
            orc_se3_log(qc, tc, &params[9 + v * 6]);
        }
        if (sqrt(un) < 1e-8) break;                     /* :404 */
    }
    free(e); free(Ji); free(Jx); free(H); free(g);
    return iter;
}

/* ======================================================================================
 * pose graph (C4, build-defined)
 * ==================================================================================== */
static void rot_apply(const double* q, const double* v, double* o) {
    double R[9];
    orc_quat_to_rot(q, R);
    for (int i = 0; i < 3; ++i) o[i] = R[i * 3] * v[0] + R[i * 3 + 1] * v[1] + R[i * 3 + 2] * v[2];
}

void orc_se3_compose(const double* a, const double* b, double* out) {
    double q[4], t[3];
    orc_quat_mul(a, b, q);
    rot_apply(a, b + 4, t);
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) out[i] = q[i] / n;
    for (int i = 0; i < 3; ++i) out[4 + i] = t[i] + a[4 + i];
}

void orc_se3_inverse(const double* a, double* out) {
    const double qc[4] = {-a[0], -a[1], -a[2], a[3]};
    double t[3];
    rot_apply(qc, a + 4, t);
    memcpy(out, qc, sizeof qc);
    for (int i = 0; i < 3; ++i) out[4 + i] = -t[i];
}

void orc_se3_retract(const double* T, const double* delta, double* out) {
    double e[7];
    orc_se3_exp(delta, e, e + 4);
    orc_se3_compose(T, e, out);
}

static void mat6_mul(const double* A, const double* B, double* C) {
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0;
            for (int k = 0; k < 6; ++k) s += A[i * 6 + k] * B[k * 6 + j];
            C[i * 6 + j] = s;
        }
}

/* ad(xi), xi = [rho, theta]: [[hat(theta), hat(rho)], [0, hat(theta)]] */
static void se3_ad(const double* xi, double* M) {
    double Hr[9], Ht[9];
    hat3(xi, Hr); hat3(xi + 3, Ht);
    memset(M, 0, sizeof(double) * 36);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            M[i * 6 + j] = Ht[i * 3 + j];
            M[i * 6 + 3 + j] = Hr[i * 3 + j];
            M[(3 + i) * 6 + 3 + j] = Ht[i * 3 + j];
        }
}

/* Ad(T), T = (R, t): [[R, hat(t) R], [0, R]] */
static void se3_Ad(const double* T, double* M) {
    double R[9], Ht[9];
    orc_quat_to_rot(T, R);
    hat3(T + 4, Ht);
    memset(M, 0, sizeof(double) * 36);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += Ht[i * 3 + k] * R[k * 3 + j];
            M[i * 6 + j] = R[i * 3 + j];
            M[i * 6 + 3 + j] = s;
            M[(3 + i) * 6 + 3 + j] = R[i * 3 + j];
        }
}

static void pg_edge(const double* Ti, const double* Tj, const double* Z, double* r, double* Ji, double* Jj) {
    double Zi[7], Tii[7], A[7], E[7];
    orc_se3_inverse(Z, Zi);
    orc_se3_inverse(Ti, Tii);
    orc_se3_compose(Tii, Tj, A);        /* T_i^-1 T_j */
    orc_se3_compose(Zi, A, E);
    if (E[3] < 0) for (int k = 0; k < 4; ++k) E[k] = -E[k];   /* shortest rotation */
    orc_se3_log(E, E + 4, r);
    if (Ji || Jj) {
        double ad[36], ad2[36], Jr[36];
        se3_ad(r, ad);
        mat6_mul(ad, ad, ad2);
        for (int k = 0; k < 36; ++k) Jr[k] = 0.5 * ad[k] + ad2[k] / 12.0;
        for (int k = 0; k < 6; ++k) Jr[k * 7] += 1.0;
        if (Jj) memcpy(Jj, Jr, sizeof Jr);
        if (Ji) {
            double Ainv[7], AdM[36];
            orc_se3_inverse(A, Ainv);   /* T_j^-1 T_i */
            se3_Ad(Ainv, AdM);
            mat6_mul(Jr, AdM, Ji);
            for (int k = 0; k < 36; ++k) Ji[k] = -Ji[k];
        }
    }
}

double orc_pg_evaluate(const orc_pg_problem* p, double* r, double* Ji, double* Jj) {
    double cost = 0;
    for (int e = 0; e < p->n_edges; ++e) {
        const int i = p->edge_i[e], j = p->edge_j[e];
        double re[6], ji[36], jj[36];
        pg_edge(&p->poses[i * 7], &p->poses[j * 7], &p->meas[e * 7], re, (Ji || Jj) ? ji : NULL, (Ji || Jj) ? jj : NULL);
        if (p->node_fixed) {
            if (p->node_fixed[i]) memset(ji, 0, sizeof ji);
            if (p->node_fixed[j]) memset(jj, 0, sizeof jj);
        }
        for (int k = 0; k < 6; ++k) cost += re[k] * re[k];
        if (r) memcpy(&r[(size_t)e * 6], re, sizeof re);
        if (Ji) memcpy(&Ji[(size_t)e * 36], ji, sizeof ji);
        if (Jj) memcpy(&Jj[(size_t)e * 36], jj, sizeof jj);
    }
    return 0.5 * cost;
}

double orc_pg_ate(int n, const double* truth, const double* est) {
    double s = 0;
    for (int i = 0; i < n; ++i) {
        double ti[7], d[7], xi[6];
        orc_se3_inverse(&truth[i * 7], ti);
        orc_se3_compose(ti, &est[i * 7], d);
        if (d[3] < 0) for (int k = 0; k < 4; ++k) d[k] = -d[k];
        orc_se3_log(d, d + 4, xi);
        for (int k = 0; k < 6; ++k) s += xi[k] * xi[k];
    }
    return sqrt(s / n);
}

typedef struct { orc_pg_problem* p; } pg_ctx;

static int pg_residual_cb(void* user, const double* x, double* r, double* J) {
    pg_ctx* c = (pg_ctx*)user;
    orc_pg_problem q = *c->p;
    q.poses = (double*)x;
    const int n = 6 * q.n_nodes;
    if (!J) { orc_pg_evaluate(&q, r, NULL, NULL); return 0; }
    double* Ji = malloc(sizeof(double) * 36 * q.n_edges);
    double* Jj = malloc(sizeof(double) * 36 * q.n_edges);
    orc_pg_evaluate(&q, r, Ji, Jj);
    memset(J, 0, sizeof(double) * (size_t)6 * q.n_edges * n);
    for (int e = 0; e < q.n_edges; ++e)
        for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 6; ++b) {
                J[(size_t)(e * 6 + a) * n + q.edge_i[e] * 6 + b] = Ji[(size_t)e * 36 + a * 6 + b];
                J[(size_t)(e * 6 + a) * n + q.edge_j[e] * 6 + b] = Jj[(size_t)e * 36 + a * 6 + b];
            }
    /* fixed nodes: unit regulariser rows are not needed -- their columns are zero and the LM
     * diagonal (clamped at 1e-6) keeps the system positive definite; the step there is 0 */
    free(Ji); free(Jj);
    return 0;
}

static void pg_plus_cb(void* user, const double* x, const double* d, double* out) {
    pg_ctx* c = (pg_ctx*)user;
    for (int i = 0; i < c->p->n_nodes; ++i) {
        if (c->p->node_fixed && c->p->node_fixed[i]) memcpy(&out[i * 7], &x[i * 7], sizeof(double) * 7);
        else orc_se3_retract(&x[i * 7], &d[i * 6], &out[i * 7]);
    }
}

int orc_pg_solve(orc_pg_problem* p, const orc_lm_options* opt, orc_lm_summary* sum, double* trace) {
    pg_ctx c = {p};
    return orc_dense_lm(pg_residual_cb, pg_plus_cb, &c, 7 * p->n_nodes, 6 * p->n_nodes, 6 * p->n_edges, p->poses,
                        NULL, NULL, opt, sum, trace);
}

/* --------------------------------------------------------------------------------------
 * C4-SIZE pose graph (10 000 nodes): the same Levenberg-Marquardt control flow as orc_dense_lm above (cost, Jacobi scaling
 * fixed at the first linearisation, clamped LM diagonal / radius, rho test, radius update, the three stopping rules, trace
 * columns), with the damped normal equations (J^T J + D) x = -g solved WITHOUT forming them: conjugate gradients on the
 * matrix-free product, run to 1e-13 of |g| and then CERTIFIED -- the explicit residual |(J^T J + D) x + g| <= 1e-10 |g| is
 * recomputed from scratch and the solve fails loudly if it does not hold.  So the step is the exact LM step to that accuracy
 * however the iteration got there; the preconditioner (6x6 block Jacobi + a coarse space of six rigid-body modes per group of
 * consecutive nodes, delta_k = Ad(T_k^-1 T_ref) xi, Galerkin coarse matrix factored densely) only decides how long it takes.
 * Build-defined like everything about C4 (the reference has no pose graph): conventions st23-lie-group-v2/doc.tex:862-996.
 * Checked against orc_pg_solve (dense normal equations) trace for trace on small graphs and against a sparse direct solve
 * (scipy splu) of one C4-size system: tests/test_oracle_pg.py, tests/golden/make_oracle_traces.py. */
/* OpenMP team size for a loop of `work` items: the GPU box has 256 logical cores, and a parallel region over a few hundred
 * items with 256 threads costs more than the loop (a 150-node solve took minutes there); at most 16 threads, 512 items each */
static int pgs_threads(int work) {
    int t = work / 512;
    if (t < 1) t = 1;
    if (t > 16) t = 16;
    return t;
}
typedef struct {
    int n, m;
    const int *ei, *ej;
    const double *Ji, *Jj, *D;
    int *nstart, *ncode;
    double* t;
} pgs_op;

static void pgs_apply(const pgs_op* A, const double* v, double* q) {
    const int n = A->n, m = A->m;
#pragma omp parallel for schedule(static) num_threads(pgs_threads(m))
    for (int e = 0; e < m; ++e) {
        const double *ji = &A->Ji[(size_t)e * 36], *jj = &A->Jj[(size_t)e * 36];
        const double *vi = &v[(size_t)A->ei[e] * 6], *vj = &v[(size_t)A->ej[e] * 6];
        for (int a = 0; a < 6; ++a) {
            double s = 0;
            for (int k = 0; k < 6; ++k) s += ji[a * 6 + k] * vi[k] + jj[a * 6 + k] * vj[k];
            A->t[(size_t)e * 6 + a] = s;
        }
    }
#pragma omp parallel for schedule(static) num_threads(pgs_threads(n))
    for (int i = 0; i < n; ++i) {
        double acc[6];
        for (int k = 0; k < 6; ++k) acc[k] = A->D ? A->D[(size_t)i * 6 + k] * v[(size_t)i * 6 + k] : 0.0;
        for (int c = A->nstart[i]; c < A->nstart[i + 1]; ++c) {
            const int e = A->ncode[c] >> 1, side = A->ncode[c] & 1;
            const double* J = side ? &A->Jj[(size_t)e * 36] : &A->Ji[(size_t)e * 36];
            const double* te = &A->t[(size_t)e * 6];
            for (int k = 0; k < 6; ++k) {
                double s = 0;
                for (int a = 0; a < 6; ++a) s += J[a * 6 + k] * te[a];
                acc[k] += s;
            }
        }
        for (int k = 0; k < 6; ++k) q[(size_t)i * 6 + k] = acc[k];
    }
}

static double vdot(const double* a, const double* b, int n) {
    double s = 0;
#pragma omp parallel for reduction(+ : s) schedule(static) num_threads(pgs_threads(n / 8))
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* 6x6 SPD inverse by Gauss-Jordan (no pivoting); returns 0 on success */
static int inv6_spd(const double* A, double* B) {
    double M[36];
    memcpy(M, A, sizeof M);
    memset(B, 0, sizeof(double) * 36);
    for (int k = 0; k < 6; ++k) B[k * 7] = 1.0;
    for (int c = 0; c < 6; ++c) {
        if (!(M[c * 7] > 0.0)) return 1;
        const double inv = 1.0 / M[c * 7];
        for (int k = 0; k < 6; ++k) { M[c * 6 + k] *= inv; B[c * 6 + k] *= inv; }
        for (int r = 0; r < 6; ++r) {
            if (r == c) continue;
            const double f = M[r * 6 + c];
            for (int k = 0; k < 6; ++k) { M[r * 6 + k] -= f * M[c * 6 + k]; B[r * 6 + k] -= f * B[c * 6 + k]; }
        }
    }
    return 0;
}

int orc_pg_solve_sparse(orc_pg_problem* p, const orc_lm_options* opt, orc_lm_summary* sum, double* trace,
                        int* cg_iterations_total, double* worst_linear_residual) {
    const int n = p->n_nodes, m = p->n_edges, N = 6 * n;
    const int agg = n >= 4000 ? 64 : n >= 1000 ? 32 : n >= 200 ? 16 : 8;
    const int na = (n + agg - 1) / agg, nc = 6 * na;
    double* r = malloc(sizeof(double) * 6 * (size_t)m);
    double* rn = malloc(sizeof(double) * 6 * (size_t)m);
    double* Ji = malloc(sizeof(double) * 36 * (size_t)m);
    double* Jj = malloc(sizeof(double) * 36 * (size_t)m);
    double* g = calloc((size_t)N, sizeof(double));
    double* Hd = calloc((size_t)n * 36, sizeof(double));
    double* Minv = calloc((size_t)n * 36, sizeof(double));
    double* Pm = calloc((size_t)n * 36, sizeof(double));
    double* D = calloc((size_t)N, sizeof(double));
    double* scale = calloc((size_t)N, sizeof(double));
    double* dx = calloc((size_t)N, sizeof(double));
    double* cr = calloc((size_t)N, sizeof(double));
    double* cz = calloc((size_t)N, sizeof(double));
    double* cp = calloc((size_t)N, sizeof(double));
    double* cq = calloc((size_t)N, sizeof(double));
    double* xn = malloc(sizeof(double) * 7 * (size_t)n);
    double* Ac = malloc(sizeof(double) * (size_t)nc * nc);
    double* Lt = malloc(sizeof(double) * (size_t)nc * nc);
    double* Gi = malloc(sizeof(double) * 36 * (size_t)m);
    double* Gj = malloc(sizeof(double) * 36 * (size_t)m);
    double* rc_ = calloc((size_t)nc, sizeof(double));
    int* nstart = calloc((size_t)n + 1, sizeof(int));
    int* ncode = malloc(sizeof(int) * 2 * (size_t)m);
    double* tbuf = malloc(sizeof(double) * 6 * (size_t)m);
    memset(sum, 0, sizeof *sum);
    const double t_start = now_s();
    int rc = ORC_NO_CONVERGENCE, cg_total = 0;
    double worst = 0.0;
    sum->termination_reason = ORC_TERM_MAX_ITER;
    {   /* edge ends by node */
        for (int e = 0; e < m; ++e) { ++nstart[p->edge_i[e] + 1]; ++nstart[p->edge_j[e] + 1]; }
        for (int i = 0; i < n; ++i) nstart[i + 1] += nstart[i];
        int* fill = malloc(sizeof(int) * (size_t)n);
        memcpy(fill, nstart, sizeof(int) * (size_t)n);
        for (int e = 0; e < m; ++e) { ncode[fill[p->edge_i[e]]++] = 2 * e; ncode[fill[p->edge_j[e]]++] = 2 * e + 1; }
        free(fill);
    }
    pgs_op A = {n, m, p->edge_i, p->edge_j, Ji, Jj, D, nstart, ncode, tbuf};
    pgs_op A0 = A;
    A0.D = NULL;

#define PGS_LINEARIZE(costvar)                                                                      \
    do {                                                                                            \
        costvar = orc_pg_evaluate(p, r, Ji, Jj);                                                     \
        _Pragma("omp parallel for schedule(static) num_threads(pgs_threads(n))")                    \
        for (int i = 0; i < n; ++i) {                                                               \
            double* gi = &g[(size_t)i * 6];                                                         \
            double* hi = &Hd[(size_t)i * 36];                                                       \
            memset(gi, 0, sizeof(double) * 6); memset(hi, 0, sizeof(double) * 36);                  \
            for (int c = nstart[i]; c < nstart[i + 1]; ++c) {                                       \
                const int e = ncode[c] >> 1, side = ncode[c] & 1;                                   \
                const double* J = side ? &Jj[(size_t)e * 36] : &Ji[(size_t)e * 36];                 \
                for (int a = 0; a < 6; ++a)                                                         \
                    for (int k = 0; k < 6; ++k) {                                                   \
                        gi[k] += J[a * 6 + k] * r[(size_t)e * 6 + a];                               \
                        for (int l = 0; l < 6; ++l) hi[k * 6 + l] += J[a * 6 + k] * J[a * 6 + l];   \
                    }                                                                               \
            }                                                                                       \
        }                                                                                           \
        gmax = 0;                                                                                   \
        for (int a = 0; a < N; ++a) if (fabs(g[a]) > gmax) gmax = fabs(g[a]);                       \
        /* coarse space: P_k = Ad(T_k^-1 T_ref(group)), zero rows for constant nodes; G = J P */    \
        for (int k = 0; k < n; ++k) {                                                               \
            const int ref = (k / agg) * agg + agg / 2 < n ? (k / agg) * agg + agg / 2 : n - 1;      \
            double Tki[7], rel[7];                                                                  \
            orc_se3_inverse(&p->poses[(size_t)k * 7], Tki);                                         \
            orc_se3_compose(Tki, &p->poses[(size_t)ref * 7], rel);                                  \
            se3_Ad(rel, &Pm[(size_t)k * 36]);                                                       \
            if (p->node_fixed && p->node_fixed[k]) memset(&Pm[(size_t)k * 36], 0, sizeof(double) * 36); \
        }                                                                                           \
        _Pragma("omp parallel for schedule(static) num_threads(pgs_threads(m))")                    \
        for (int e = 0; e < m; ++e) {                                                               \
            mat6_mul(&Ji[(size_t)e * 36], &Pm[(size_t)p->edge_i[e] * 36], &Gi[(size_t)e * 36]);     \
            mat6_mul(&Jj[(size_t)e * 36], &Pm[(size_t)p->edge_j[e] * 36], &Gj[(size_t)e * 36]);     \
        }                                                                                           \
    } while (0)

    double cost, gmax;
    PGS_LINEARIZE(cost);
    sum->initial_cost = cost;
    for (int a = 0; a < N; ++a) scale[a] = opt->jacobi_scaling ? 1.0 / (1.0 + sqrt(Hd[(size_t)(a / 6) * 36 + (a % 6) * 7])) : 1.0;
    double radius = opt->initial_trust_region_radius, decrease_factor = 2.0;
    double x_norm = 0;
    for (int a = 0; a < 7 * n; ++a) x_norm += p->poses[a] * p->poses[a];
    x_norm = sqrt(x_norm);
    int iter = 0;
    if (trace) { memset(trace, 0, sizeof(double) * ORC_TRACE_COLS); trace[0] = cost; trace[2] = gmax; trace[5] = radius; trace[6] = 1; }
    if (!isfinite(cost)) { rc = ORC_FAILURE; sum->termination_reason = ORC_TERM_SOLVER_FAIL; goto fin; }      /* (Ceres: initial evaluation failed) */
    if (gmax <= opt->gradient_tolerance) { rc = ORC_CONVERGENCE; sum->termination_reason = ORC_TERM_GRADIENT; goto fin; }

    while (1) {
        if (iter >= opt->max_num_iterations) { rc = ORC_NO_CONVERGENCE; sum->termination_reason = ORC_TERM_MAX_ITER; break; }
        if (radius < opt->min_trust_region_radius) { rc = ORC_CONVERGENCE; sum->termination_reason = ORC_TERM_MIN_RADIUS; break; }
        ++iter;
        int ok = 1;
        for (int a = 0; a < N; ++a) {
            const double s2 = scale[a] * scale[a];
            const double h = Hd[(size_t)(a / 6) * 36 + (a % 6) * 7];
            D[a] = fmin(fmax(h * s2, opt->min_lm_diagonal), opt->max_lm_diagonal) / radius / s2;
        }
        /* preconditioner: block Jacobi */
        for (int i = 0; i < n && ok; ++i) {
            double B[36];
            memcpy(B, &Hd[(size_t)i * 36], sizeof B);
            for (int k = 0; k < 6; ++k) B[k * 7] += D[(size_t)i * 6 + k];
            if (p->node_fixed && p->node_fixed[i]) { memset(&Minv[(size_t)i * 36], 0, sizeof(double) * 36); continue; }
            if (inv6_spd(B, &Minv[(size_t)i * 36])) ok = 0;
        }
        /* ... + Galerkin coarse matrix Ac = P^T (J^T J + D) P, dense, Cholesky */
        if (ok) {
            memset(Ac, 0, sizeof(double) * (size_t)nc * nc);
            for (int e = 0; e < m; ++e) {
                const int ai = p->edge_i[e] / agg, aj = p->edge_j[e] / agg;
                const double *gi = &Gi[(size_t)e * 36], *gj = &Gj[(size_t)e * 36];
                for (int a = 0; a < 6; ++a)
                    for (int k = 0; k < 6; ++k)
                        for (int l = 0; l < 6; ++l) {
                            Ac[(size_t)(6 * ai + k) * nc + 6 * ai + l] += gi[a * 6 + k] * gi[a * 6 + l];
                            Ac[(size_t)(6 * aj + k) * nc + 6 * aj + l] += gj[a * 6 + k] * gj[a * 6 + l];
                            Ac[(size_t)(6 * ai + k) * nc + 6 * aj + l] += gi[a * 6 + k] * gj[a * 6 + l];
                            Ac[(size_t)(6 * aj + k) * nc + 6 * ai + l] += gj[a * 6 + k] * gi[a * 6 + l];
                        }
            }
            for (int k = 0; k < n; ++k) {
                const int a0 = 6 * (k / agg);
                const double* P = &Pm[(size_t)k * 36];
                for (int c = 0; c < 6; ++c)
                    for (int u = 0; u < 6; ++u)
                        for (int v = 0; v < 6; ++v) Ac[(size_t)(a0 + u) * nc + a0 + v] += P[c * 6 + u] * D[(size_t)k * 6 + c] * P[c * 6 + v];
            }
            for (int a = 0; a < nc; ++a) if (!(Ac[(size_t)a * nc + a] > 0.0)) Ac[(size_t)a * nc + a] = 1.0;   /* a group of constant nodes */
            if (orc_cholesky_lower(Ac, nc, pgs_threads(nc * 4)) != 0) ok = 0;
            if (ok)
                for (int a = 0; a < nc; ++a)
                    for (int b = 0; b <= a; ++b) Lt[(size_t)b * nc + a] = Ac[(size_t)a * nc + b];
        }
#define PGS_PRECOND(rv, zv)                                                                         \
        do {                                                                                        \
            memset(rc_, 0, sizeof(double) * (size_t)nc);                                            \
            for (int k = 0; k < n; ++k) {                                                           \
                const double* P = &Pm[(size_t)k * 36];                                              \
                for (int u = 0; u < 6; ++u) {                                                       \
                    double s = 0, z0 = 0;                                                           \
                    for (int c = 0; c < 6; ++c) { s += P[c * 6 + u] * rv[(size_t)k * 6 + c]; z0 += Minv[(size_t)k * 36 + u * 6 + c] * rv[(size_t)k * 6 + c]; } \
                    rc_[6 * (k / agg) + u] += s;                                                    \
                    zv[(size_t)k * 6 + u] = z0;                                                     \
                }                                                                                   \
            }                                                                                       \
            for (int a = 0; a < nc; ++a) rc_[a] = (rc_[a] - dotk(&Ac[(size_t)a * nc], rc_, a)) / Ac[(size_t)a * nc + a];          \
            for (int a = nc - 1; a >= 0; --a) rc_[a] = (rc_[a] - dotk(&Lt[(size_t)a * nc + a + 1], &rc_[a + 1], nc - 1 - a)) / Ac[(size_t)a * nc + a]; \
            for (int k = 0; k < n; ++k) {                                                           \
                const double* P = &Pm[(size_t)k * 36];                                              \
                for (int c = 0; c < 6; ++c) {                                                       \
                    double s = 0;                                                                   \
                    for (int u = 0; u < 6; ++u) s += P[c * 6 + u] * rc_[6 * (k / agg) + u];         \
                    zv[(size_t)k * 6 + c] += s;                                                     \
                }                                                                                   \
            }                                                                                       \
        } while (0)
        double model_change = 0, new_cost = 0, step_norm = 0, rho = 0;
        if (ok) {
            /* PCG on (J^T J + D) x = -g */
            double bb = 0;
            for (int a = 0; a < N; ++a) { dx[a] = 0; cr[a] = -g[a]; bb += g[a] * g[a]; }
            PGS_PRECOND(cr, cz);
            memcpy(cp, cz, sizeof(double) * (size_t)N);
            double rz = vdot(cr, cz, N), rr = bb;
            int k = 0;
            while (rr > 1e-26 * bb && k < 20000) {
                pgs_apply(&A, cp, cq);
                const double pq = vdot(cp, cq, N);
                if (!(pq > 0.0)) { ok = 0; break; }
                const double alpha = rz / pq;
                for (int a = 0; a < N; ++a) { dx[a] += alpha * cp[a]; cr[a] -= alpha * cq[a]; }
                PGS_PRECOND(cr, cz);
                const double rzn = vdot(cr, cz, N);
                rr = vdot(cr, cr, N);
                const double beta = rzn / rz;
                rz = rzn;
                for (int a = 0; a < N; ++a) cp[a] = cz[a] + beta * cp[a];
                ++k;
            }
            cg_total += k;
            /* certificate: the residual recomputed from scratch */
            pgs_apply(&A, dx, cq);
            double res = 0;
            for (int a = 0; a < N; ++a) res += (cq[a] + g[a]) * (cq[a] + g[a]);
            res = sqrt(res / (bb > 0 ? bb : 1.0));
            if (res > worst) worst = res;
            if (!(res <= 1e-10)) ok = 0;
        }
        if (ok) {
            pgs_apply(&A0, dx, cq);                         /* J^T J dx */
            model_change = -vdot(g, dx, N) - 0.5 * vdot(dx, cq, N);
            if (!(model_change > 0) || !isfinite(model_change)) ok = 0;
        }
        int accepted = 0;
        if (ok) {
            for (int i = 0; i < n; ++i) {
                if (p->node_fixed && p->node_fixed[i]) memcpy(&xn[(size_t)i * 7], &p->poses[(size_t)i * 7], sizeof(double) * 7);
                else orc_se3_retract(&p->poses[(size_t)i * 7], &dx[(size_t)i * 6], &xn[(size_t)i * 7]);
            }
            orc_pg_problem q = *p;
            q.poses = xn;
            new_cost = orc_pg_evaluate(&q, rn, NULL, NULL);
            for (int a = 0; a < 7 * n; ++a) step_norm += (xn[a] - p->poses[a]) * (xn[a] - p->poses[a]);
            step_norm = sqrt(step_norm);
            const double cost_change = cost - new_cost;
            rho = cost_change / model_change;
            if (trace) { double* tr = &trace[iter * ORC_TRACE_COLS]; tr[0] = new_cost; tr[1] = cost_change; tr[3] = step_norm; tr[4] = rho; }
            if (step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) {
                rc = ORC_CONVERGENCE; sum->termination_reason = ORC_TERM_PARAMETER;
                if (trace) { trace[iter * ORC_TRACE_COLS + 5] = radius; trace[iter * ORC_TRACE_COLS + 2] = gmax; }
                break;
            }
            if (fabs(cost_change) <= opt->function_tolerance * cost) {
                if (opt->function_tolerance_takes_step && rho > opt->min_relative_decrease) {
                    memcpy(p->poses, xn, sizeof(double) * 7 * (size_t)n); cost = new_cost; ++sum->num_successful_steps;
                    if (trace) trace[iter * ORC_TRACE_COLS + 6] = 1;
                }
                rc = ORC_CONVERGENCE; sum->termination_reason = ORC_TERM_FUNCTION;
                if (trace) { trace[iter * ORC_TRACE_COLS + 5] = radius; trace[iter * ORC_TRACE_COLS + 2] = gmax; }
                break;
            }
            accepted = rho > opt->min_relative_decrease;
        }
        if (accepted) {
            memcpy(p->poses, xn, sizeof(double) * 7 * (size_t)n);
            x_norm = 0;
            for (int a = 0; a < 7 * n; ++a) x_norm += p->poses[a] * p->poses[a];
            x_norm = sqrt(x_norm);
            ++sum->num_successful_steps;
            PGS_LINEARIZE(cost);
            const double t = 2.0 * rho - 1.0;
            radius = fmin(opt->max_trust_region_radius, radius / fmax(1.0 / 3.0, 1.0 - t * t * t));
            decrease_factor = 2.0;
        } else {
            ++sum->num_unsuccessful_steps;
            radius /= decrease_factor;
            decrease_factor *= 2.0;
        }
        if (trace) {
            double* tr = &trace[iter * ORC_TRACE_COLS];
            if (!ok) { tr[0] = cost; tr[1] = 0; tr[3] = 0; tr[4] = 0; }
            tr[2] = gmax; tr[5] = radius; tr[6] = accepted;
        }
        if (accepted && gmax <= opt->gradient_tolerance) { rc = ORC_CONVERGENCE; sum->termination_reason = ORC_TERM_GRADIENT; break; }
    }
fin:
    sum->num_iterations = iter;
    sum->final_cost = cost;
    sum->final_radius = radius;
    sum->final_gradient_max_norm = gmax;
    sum->termination_type = rc;
    sum->seconds_total = now_s() - t_start;
    if (cg_iterations_total) *cg_iterations_total = cg_total;
    if (worst_linear_residual) *worst_linear_residual = worst;
    free(r); free(rn); free(Ji); free(Jj); free(g); free(Hd); free(Minv); free(Pm); free(D); free(scale); free(dx); free(cr); free(cz);
    free(cp); free(cq); free(xn); free(Ac); free(Lt); free(Gi); free(Gj); free(rc_); free(nstart); free(ncode); free(tbuf);
    return rc;
}
#undef PGS_LINEARIZE
#undef PGS_PRECOND

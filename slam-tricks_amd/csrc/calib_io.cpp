// calib_io.cpp -- the two ends of the calibration path, host C++ behind the C ABI:
//   * chessboard corner files: st3-calibration/src/src/cbcorner.cpp:34-73 ("rows,cols" header, then
//     "i,j,x,y" per corner; the reference parses x and y with std::stof, i.e. through float)
//   * Zhang's closed-form initialisation, the start point of the refinement that runs on the device
//     (stba_calib_gauss_newton): DLT homographies (calib.cpp:55-93), intrinsics from the image of the
//     absolute conic (:95-140), extrinsics (:142-173).
// 9..20 views of <= 100 corners: a few hundred kFLOP, far below one kernel launch -- this stays on the
// host on purpose.  The null vectors are taken with a one-sided Jacobi SVD (same accuracy class as the
// reference's Eigen::JacobiSVD; forming A^T A would square the condition number of the unnormalised DLT).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "common.hpp"
#include "small_linalg.hpp"

namespace stba {
namespace {

void mat3_mul_vec(const double* M, const double* v, double* o) {
    for (int i = 0; i < 3; ++i) o[i] = M[i * 3] * v[0] + M[i * 3 + 1] * v[1] + M[i * 3 + 2] * v[2];
}
double norm3(const double* v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
void cross3(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

// nearest rotation (polar factor) by Newton iteration X <- (X + X^-T) / 2
void nearest_rotation(double* R) {
    for (int it = 0; it < 50; ++it) {
        const double a = R[0], b = R[1], c = R[2], d = R[3], e = R[4], f = R[5], g = R[6], h = R[7], i = R[8];
        const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
        const double inv[9] = {(e * i - f * h) / det, (c * h - b * i) / det, (b * f - c * e) / det,
                               (f * g - d * i) / det, (a * i - c * g) / det, (c * d - a * f) / det,
                               (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det};
        double delta = 0.0;
        for (int r = 0; r < 3; ++r)
            for (int cidx = 0; cidx < 3; ++cidx) {
                const double nv = 0.5 * (R[r * 3 + cidx] + inv[cidx * 3 + r]);     // X^-T = (X^-1)^T
                delta = std::max(delta, std::fabs(nv - R[r * 3 + cidx]));
                R[r * 3 + cidx] = nv;
            }
        if (delta < 1e-15) break;
    }
}

// SE3 log, tangent order [rho, theta] (the order of CalibSolver's pose update, calib.cpp:397-402)
void se3_log_rt(const double* R, const double* t, double* xi) {
    const double tr = R[0] + R[4] + R[8];
    const double cth = std::min(1.0, std::max(-1.0, 0.5 * (tr - 1.0)));
    const double a = std::acos(cth);
    double w[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    const double k = (a < 1e-10) ? 0.5 : a / (2.0 * std::sin(a));
    for (double& x : w) x *= k;
    const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double K2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) K2[i * 3 + j] = K[i * 3] * K[j] + K[i * 3 + 1] * K[3 + j] + K[i * 3 + 2] * K[6 + j];
    const double c1 = (a < 1e-10) ? 0.5 : (1.0 - std::cos(a)) / (a * a);
    const double c2 = (a < 1e-10) ? 1.0 / 6.0 : (a - std::sin(a)) / (a * a * a);
    double V[9];
    for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * K[i] + c2 * K2[i];
    // rho = V^-1 t (3x3 solve by Cramer)
    const double det = V[0] * (V[4] * V[8] - V[5] * V[7]) - V[1] * (V[3] * V[8] - V[5] * V[6]) + V[2] * (V[3] * V[7] - V[4] * V[6]);
    const double inv[9] = {(V[4] * V[8] - V[5] * V[7]) / det, (V[2] * V[7] - V[1] * V[8]) / det, (V[1] * V[5] - V[2] * V[4]) / det,
                           (V[5] * V[6] - V[3] * V[8]) / det, (V[0] * V[8] - V[2] * V[6]) / det, (V[2] * V[3] - V[0] * V[5]) / det,
                           (V[3] * V[7] - V[4] * V[6]) / det, (V[1] * V[6] - V[0] * V[7]) / det, (V[0] * V[4] - V[1] * V[3]) / det};
    mat3_mul_vec(inv, t, xi);
    xi[3] = w[0]; xi[4] = w[1]; xi[5] = w[2];
}

}  // namespace
}  // namespace stba

using namespace stba;

extern "C" {

int stba_corners_read(const char* path, int* rows, int* cols, double* xy, int capacity) {
    if (!path || !rows || !cols) return fail(STBA_ERR_INVALID_ARGUMENT, "null argument");
    std::ifstream f(path);
    if (!f) return fail(STBA_ERR_INVALID_ARGUMENT, std::string("cannot open ") + path);
    std::string line;
    if (!std::getline(f, line)) return fail(STBA_ERR_INVALID_ARGUMENT, "empty corner file");
    const size_t comma = line.find(',');
    if (comma == std::string::npos) return fail(STBA_ERR_INVALID_ARGUMENT, "corner file: bad header");
    const int r = std::atoi(line.substr(0, comma).c_str()), c = std::atoi(line.substr(comma + 1).c_str());
    if (r <= 0 || c <= 0) return fail(STBA_ERR_INVALID_ARGUMENT, "corner file: bad board size");
    *rows = r; *cols = c;
    if (!xy) return STBA_OK;                                   // size query
    if (capacity < r * c) return fail(STBA_ERR_INVALID_ARGUMENT, "corner buffer too small");
    std::vector<char> seen((size_t)r * c, 0);
    while (std::getline(f, line)) {
        if (line.empty()) continue;
        int i = 0, j = 0;
        float x = 0.f, y = 0.f;                                // std::stof in the reference: float precision
        if (std::sscanf(line.c_str(), "%d,%d,%f,%f", &i, &j, &x, &y) != 4 || i < 0 || i >= r || j < 0 || j >= c)
            return fail(STBA_ERR_INVALID_ARGUMENT, "corner file: bad line '" + line + "'");
        xy[2 * ((size_t)i * c + j)] = (double)x;
        xy[2 * ((size_t)i * c + j) + 1] = (double)y;
        seen[(size_t)i * c + j] = 1;
    }
    for (char s : seen)
        if (!s) return fail(STBA_ERR_INVALID_ARGUMENT, "corner file: missing corners");
    return STBA_OK;
}

int stba_corners_write(const char* path, int rows, int cols, const double* xy) {
    if (!path || !xy || rows <= 0 || cols <= 0) return fail(STBA_ERR_INVALID_ARGUMENT, "bad argument");
    FILE* f = std::fopen(path, "w");
    if (!f) return fail(STBA_ERR_INVALID_ARGUMENT, std::string("cannot open ") + path);
    std::fprintf(f, "%d,%d\n", rows, cols);
    for (int i = 0; i < rows; ++i)
        for (int j = 0; j < cols; ++j)
            std::fprintf(f, "%d,%d,%.3f,%.3f\n", i, j, xy[2 * ((size_t)i * cols + j)], xy[2 * ((size_t)i * cols + j) + 1]);
    std::fclose(f);
    return STBA_OK;
}

int stba_zhang_init(int n_views, int n_corners, const double* obj, const double* img, double* params, double* homographies) {
    if (n_views < 2 || n_corners < 4 || !obj || !img || !params) return fail(STBA_ERR_INVALID_ARGUMENT, "bad argument");
    std::vector<double> Hs((size_t)n_views * 9);
    // homographies, calib.cpp:55-93: rows [x y 1 0 0 0 -ux -uy -u], [0 0 0 x y 1 -vx -vy -v]
    for (int v = 0; v < n_views; ++v) {
        std::vector<double> A((size_t)2 * n_corners * 9, 0.0);
        for (int k = 0; k < n_corners; ++k) {
            const double x = obj[2 * ((size_t)v * n_corners + k)], y = obj[2 * ((size_t)v * n_corners + k) + 1];
            const double u = img[2 * ((size_t)v * n_corners + k)], w = img[2 * ((size_t)v * n_corners + k) + 1];
            double* r0 = &A[(size_t)(2 * k) * 9];
            double* r1 = &A[(size_t)(2 * k + 1) * 9];
            r0[0] = x; r0[1] = y; r0[2] = 1; r0[6] = -u * x; r0[7] = -u * y; r0[8] = -u;
            r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -w * x; r1[7] = -w * y; r1[8] = -w;
        }
        smallest_right_singular_vector(A, 2 * n_corners, 9, &Hs[(size_t)v * 9]);
    }
    if (homographies) std::memcpy(homographies, Hs.data(), Hs.size() * sizeof(double));
    // intrinsics, calib.cpp:95-140 (zero skew: b = [B11 B13 B22 B23 B33])
    auto cof = [](const double* H, int i, int j, double* o) {
        const double hi[3] = {H[i], H[3 + i], H[6 + i]}, hj[3] = {H[j], H[3 + j], H[6 + j]};
        o[0] = hi[0] * hj[0]; o[1] = hi[2] * hj[0] + hi[0] * hj[2]; o[2] = hi[1] * hj[1];
        o[3] = hi[2] * hj[1] + hi[1] * hj[2]; o[4] = hi[2] * hj[2];
    };
    std::vector<double> C((size_t)2 * n_views * 5);
    for (int v = 0; v < n_views; ++v) {
        double c01[5], c00[5], c11[5];
        cof(&Hs[(size_t)v * 9], 0, 1, c01); cof(&Hs[(size_t)v * 9], 0, 0, c00); cof(&Hs[(size_t)v * 9], 1, 1, c11);
        for (int k = 0; k < 5; ++k) { C[(size_t)(2 * v) * 5 + k] = c01[k]; C[(size_t)(2 * v + 1) * 5 + k] = c00[k] - c11[k]; }
    }
    double bv[5];
    smallest_right_singular_vector(C, 2 * n_views, 5, bv);
    const double b11 = bv[0], b13 = bv[1], b22 = bv[2], b23 = bv[3], b33 = bv[4];
    const double v0 = -b23 / b22;
    const double lam = b33 - (b13 * b13 - v0 * b11 * b23) / b11;
    if (!(lam / b11 > 0.0) || !(lam / b22 > 0.0)) return fail(STBA_ERR_INVALID_ARGUMENT, "zhang_init: degenerate views (no positive focal length)");
    const double alpha = std::sqrt(lam / b11), beta = std::sqrt(lam / b22);
    const double u0 = -b13 * alpha * alpha / lam;
    for (int k = 0; k < 9 + 6 * n_views; ++k) params[k] = 0.0;
    params[0] = alpha; params[1] = beta; params[2] = u0; params[3] = v0;
    // extrinsics, calib.cpp:142-173
    const double Ki[9] = {1.0 / alpha, 0, -u0 / alpha, 0, 1.0 / beta, -v0 / beta, 0, 0, 1};
    for (int v = 0; v < n_views; ++v) {
        const double* H = &Hs[(size_t)v * 9];
        const double h0[3] = {H[0], H[3], H[6]}, h1[3] = {H[1], H[4], H[7]}, h2[3] = {H[2], H[5], H[8]};
        double r1[3], r2[3], r3[3], t[3];
        mat3_mul_vec(Ki, h0, r1); mat3_mul_vec(Ki, h1, r2); mat3_mul_vec(Ki, h2, t);
        const double n1 = norm3(r1), n2 = norm3(r2);
        const double l = 1.0 / (2.0 * n1) + 1.0 / (2.0 * n2);
        for (int k = 0; k < 3; ++k) { r1[k] /= n1; r2[k] /= n2; t[k] *= l; }
        cross3(r1, r2, r3);
        cross3(r2, r3, r1);
        double R[9] = {r1[0], r2[0], r3[0], r1[1], r2[1], r3[1], r1[2], r2[2], r3[2]};
        nearest_rotation(R);
        if (t[2] < 0) {     // sign ambiguity of the homography's null vector: keep the board in front of the camera
            for (int r = 0; r < 3; ++r) { R[r * 3] = -R[r * 3]; R[r * 3 + 1] = -R[r * 3 + 1]; }
            for (double& x : t) x = -x;
        }
        se3_log_rt(R, t, &params[9 + 6 * v]);
    }
    return STBA_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------
// Trajectory files and the accuracy metric of the pose-graph path, host C++:
//   * odometry files, st16-pcl-viewer/src/src/scene.cpp:66-110 (ReadOdom): "format ascii 1.0" /
//     "element odometryInfo N" / property lines / "end_header", then N lines "timeStamp qx qy qz qw x y z";
//     the quaternion is normalised, the translation goes through float (std::stof) like in the reference
//   * absolute trajectory error, st4-kalman/src/src/pose_simulation.cpp:198-209:
//     sqrt(mean_i |log(T_truth_i^-1 T_est_i)|^2) over the 6-vector SE3 logarithm
// ------------------------------------------------------------------------------------------------------------
namespace stba {
namespace {
void quat_mul(const double* a, const double* b, double* o) {     // (x, y, z, w)
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
}  // namespace
}  // namespace stba

extern "C" {

int stba_odometry_read(const char* path, int* n_poses, double* stamps, double* poses, int capacity) {
    if (!path || !n_poses) return fail(STBA_ERR_INVALID_ARGUMENT, "null argument");
    std::ifstream f(path);
    if (!f) return fail(STBA_ERR_INVALID_ARGUMENT, std::string("cannot open ") + path);
    std::string line;
    if (!std::getline(f, line)) return fail(STBA_ERR_INVALID_ARGUMENT, "odometry file: empty");       // format line
    if (!std::getline(f, line)) return fail(STBA_ERR_INVALID_ARGUMENT, "odometry file: no element line");
    int n = -1;
    {
        char a[64], b[64];
        if (std::sscanf(line.c_str(), "%63s %63s %d", a, b, &n) != 3 || n < 0)
            return fail(STBA_ERR_INVALID_ARGUMENT, "odometry file: bad element line");
    }
    bool header_done = false;
    while (std::getline(f, line))
        if (line.size() >= 10 && line.compare(0, 10, "end_header") == 0) { header_done = true; break; }
    if (!header_done) return fail(STBA_ERR_INVALID_ARGUMENT, "odometry file: no end_header");
    *n_poses = n;
    if (!poses && !stamps) return STBA_OK;                      // size query
    if (capacity < n) return fail(STBA_ERR_INVALID_ARGUMENT, "odometry buffer too small");
    for (int i = 0; i < n; ++i) {
        if (!std::getline(f, line)) return fail(STBA_ERR_INVALID_ARGUMENT, "odometry file: fewer poses than announced");
        double ts, q[4];
        float t[3];
        if (std::sscanf(line.c_str(), "%lf %lf %lf %lf %lf %f %f %f", &ts, &q[0], &q[1], &q[2], &q[3], &t[0], &t[1], &t[2]) != 8)
            return fail(STBA_ERR_INVALID_ARGUMENT, "odometry file: bad pose line");
        const double nq = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        if (!(nq > 0.0)) return fail(STBA_ERR_INVALID_ARGUMENT, "odometry file: zero quaternion");
        if (stamps) stamps[i] = ts;
        if (poses) {
            for (int k = 0; k < 4; ++k) poses[7 * (size_t)i + k] = q[k] / nq;
            for (int k = 0; k < 3; ++k) poses[7 * (size_t)i + 4 + k] = (double)t[k];
        }
    }
    return STBA_OK;
}

int stba_odometry_write(const char* path, int n_poses, const double* stamps, const double* poses) {
    if (!path || n_poses < 0 || !poses) return fail(STBA_ERR_INVALID_ARGUMENT, "bad argument");
    FILE* f = std::fopen(path, "w");
    if (!f) return fail(STBA_ERR_INVALID_ARGUMENT, std::string("cannot open ") + path);
    std::fprintf(f, "format ascii 1.0\nelement odometryInfo %d\nproperty double timeStamp\nproperty double qx\nproperty double qy\n"
                    "property double qz\nproperty double qw\nproperty double x\nproperty double y\nproperty double z\nend_header\n", n_poses);
    for (int i = 0; i < n_poses; ++i) {
        const double* p = poses + 7 * (size_t)i;
        std::fprintf(f, "%.9f %.10f %.10f %.10f %.10f %.10f %.10f %.10f\n", stamps ? stamps[i] : (double)i, p[0], p[1], p[2], p[3],
                     p[4], p[5], p[6]);
    }
    std::fclose(f);
    return STBA_OK;
}

int stba_trajectory_ate(int n_poses, const double* truth, const double* estimate, double* ate) {
    if (n_poses <= 0 || !truth || !estimate || !ate) return fail(STBA_ERR_INVALID_ARGUMENT, "bad argument");
    double s = 0.0;
    for (int i = 0; i < n_poses; ++i) {
        const double* a = truth + 7 * (size_t)i;
        const double* b = estimate + 7 * (size_t)i;
        // D = A^-1 B:  q = conj(qa) qb,  t = Ra^T (tb - ta)
        const double qc[4] = {-a[0], -a[1], -a[2], a[3]};
        double q[4], Ra[9], R[9];
        quat_mul(qc, b, q);
        if (q[3] < 0) for (double& x : q) x = -x;
        quat_to_rot(a, Ra);
        quat_to_rot(q, R);
        const double d[3] = {b[4] - a[4], b[5] - a[5], b[6] - a[6]};
        const double t[3] = {Ra[0] * d[0] + Ra[3] * d[1] + Ra[6] * d[2], Ra[1] * d[0] + Ra[4] * d[1] + Ra[7] * d[2],
                             Ra[2] * d[0] + Ra[5] * d[1] + Ra[8] * d[2]};
        double xi[6];
        se3_log_rt(R, t, xi);
        for (int k = 0; k < 6; ++k) s += xi[k] * xi[k];
    }
    *ate = std::sqrt(s / n_poses);
    return STBA_OK;
}

}  // extern "C"

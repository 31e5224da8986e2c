// small_linalg.hpp -- host-side dense helpers for the closed-form initialisers (calib_io.cpp, two_view.hip)
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

namespace stba {

// One-sided (Hestenes) Jacobi SVD of A (m x n, row-major, m >= n): on return the columns of A are
// U * diag(sigma) (mutually orthogonal), V (n x n, row-major) holds the right singular vectors.
inline void jacobi_svd_onesided(std::vector<double>& A, int m, int n, std::vector<double>& V) {
    V.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double app = 0, aqq = 0, apq = 0;
                for (int r = 0; r < m; ++r) {
                    const double x = A[(size_t)r * n + p], y = A[(size_t)r * n + q];
                    app += x * x; aqq += y * y; apq += x * y;
                }
                if (apq == 0.0) continue;
                off = std::max(off, std::fabs(apq) / std::sqrt(std::max(app * aqq, 1e-300)));
                const double zeta = (aqq - app) / (2.0 * apq);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < m; ++r) {
                    const double x = A[(size_t)r * n + p], y = A[(size_t)r * n + q];
                    A[(size_t)r * n + p] = c * x - s * y;
                    A[(size_t)r * n + q] = s * x + c * y;
                }
                for (int r = 0; r < n; ++r) {
                    const double x = V[(size_t)r * n + p], y = V[(size_t)r * n + q];
                    V[(size_t)r * n + p] = c * x - s * y;
                    V[(size_t)r * n + q] = s * x + c * y;
                }
            }
        if (off < 1e-15) break;
    }
}

// right singular vector of the smallest singular value (same accuracy class as Eigen::JacobiSVD, which the
// reference uses; forming A^T A instead would square the condition number of the unnormalised DLT systems)
inline void smallest_right_singular_vector(std::vector<double> A, int m, int n, double* v_out) {
    std::vector<double> V;
    jacobi_svd_onesided(A, m, n, V);
    int best = 0;
    double bn = 1e300;
    for (int j = 0; j < n; ++j) {
        double nn = 0;
        for (int r = 0; r < m; ++r) nn += A[(size_t)r * n + j] * A[(size_t)r * n + j];
        if (nn < bn) { bn = nn; best = j; }
    }
    for (int r = 0; r < n; ++r) v_out[r] = V[(size_t)r * n + best];
}

// full SVD of a 3x3 (row-major): M = U diag(s) V^T, s descending, U and V orthogonal (a zero singular
// value gets the cross product of the other two left vectors)
inline void svd3(const double* M, double* U, double* s, double* V) {
    std::vector<double> A(M, M + 9), Vv;
    jacobi_svd_onesided(A, 3, 3, Vv);
    double nrm[3];
    int ord[3] = {0, 1, 2};
    for (int j = 0; j < 3; ++j) nrm[j] = std::sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]);
    std::sort(ord, ord + 3, [&](int a, int b) { return nrm[a] > nrm[b]; });
    for (int k = 0; k < 3; ++k) {
        const int j = ord[k];
        s[k] = nrm[j];
        for (int r = 0; r < 3; ++r) { V[r * 3 + k] = Vv[(size_t)r * 3 + j]; U[r * 3 + k] = (nrm[j] > 0) ? A[(size_t)r * 3 + j] / nrm[j] : 0.0; }
    }
    if (!(s[2] > 1e-12 * s[0])) {            // rank 2: complete U with the cross product
        U[2] = U[3] * U[7] - U[6] * U[4];
        U[5] = U[6] * U[1] - U[0] * U[7];
        U[8] = U[0] * U[4] - U[3] * U[1];
    }
}

}  // namespace stba

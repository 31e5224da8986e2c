// Opportunistic REAL-Ceres baseline row of bench.py: compiled and run only when a Ceres
// installation is found on the box (the build image ships none, so this file has never been compiled there --
// bench.py reports {"found": false} in that case and never assumes a number).
// Semantics of st20-g2o/src/include/test_ceres.h:98-152: one residual block per observation with parameter
// blocks {SO3 quaternion (x,y,z,w) 4, POS 3, landmark 3}, the quaternion updated as q (x) exp(delta)
// (LieLocalParameterization<Sophus::SO3d>, test_ceres.h:14-45), first and last camera constant,
// linear_solver_type = SPARSE_SCHUR; analytic cost function (SURVEY.md 8d asks for the analytic form so that the
// baseline is not penalised by autodiff), fixed iteration count.
// usage: ceres_baseline <scene.bin> <iterations> <threads>    -> one JSON line
#include <ceres/ceres.h>
#include <ceres/version.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <vector>

namespace {

void QuatMul(const double* a, const double* b, double* o) {
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
void RightPlus(const double* q, const double* d, double* out) {
    const double th = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const double im = th < 1e-10 ? 0.5 : std::sin(0.5 * th) / th;
    const double e[4] = {im * d[0], im * d[1], im * d[2], std::cos(0.5 * th)};
    QuatMul(q, e, out);
}
void RightPlusJacobian(const double* q, double* J) {      // 4 x 3 row-major, d (q (x) exp(delta)) / d delta at 0
    const double x = 0.5 * q[0], y = 0.5 * q[1], z = 0.5 * q[2], w = 0.5 * q[3];
    const double M[12] = {w, -z, y, z, w, -x, -y, x, w, -x, -y, -z};
    std::memcpy(J, M, sizeof M);
}

#if CERES_VERSION_MAJOR > 2 || (CERES_VERSION_MAJOR == 2 && CERES_VERSION_MINOR >= 1)
class So3RightManifold : public ceres::Manifold {
public:
    int AmbientSize() const override { return 4; }
    int TangentSize() const override { return 3; }
    bool Plus(const double* x, const double* delta, double* xpd) const override { RightPlus(x, delta, xpd); return true; }
    bool PlusJacobian(const double* x, double* J) const override { RightPlusJacobian(x, J); return true; }
    bool Minus(const double* y, const double* x, double* d) const override {      // log(x^-1 y)
        const double xi[4] = {-x[0], -x[1], -x[2], x[3]};
        double r[4]; QuatMul(xi, y, r);
        const double n = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        const double k = n < 1e-10 ? 2.0 / r[3] : 2.0 * std::atan2(n, r[3]) / n;
        d[0] = k * r[0]; d[1] = k * r[1]; d[2] = k * r[2];
        return true;
    }
    bool MinusJacobian(const double* x, double* J) const override {
        double P[12]; RightPlusJacobian(x, P);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) J[i * 4 + j] = 4.0 * P[j * 3 + i];   // pseudo-inverse of P (P^T P = I/4)
        return true;
    }
};
#define SET_MANIFOLD(problem, ptr, m) (problem).SetManifold((ptr), (m))
#else
class So3RightManifold : public ceres::LocalParameterization {
public:
    int GlobalSize() const override { return 4; }
    int LocalSize() const override { return 3; }
    bool Plus(const double* x, const double* delta, double* xpd) const override { RightPlus(x, delta, xpd); return true; }
    bool ComputeJacobian(const double* x, double* J) const override { RightPlusJacobian(x, J); return true; }
};
#define SET_MANIFOLD(problem, ptr, m) (problem).SetParameterization((ptr), (m))
#endif

// r = proj(conj(q) (L - t)) - feature, ambient Jacobians (2x4 | 2x3 | 2x3)
class Reprojection : public ceres::SizedCostFunction<2, 4, 3, 3> {
public:
    Reprojection(double fx, double fy) : fx_(fx), fy_(fy) {}
    bool Evaluate(double const* const* p, double* r, double** J) const override {
        const double* q = p[0]; const double* t = p[1]; const double* L = p[2];
        const double x = q[0], y = q[1], z = q[2], w = q[3];
        const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z),
                             2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
        const double d[3] = {L[0] - t[0], L[1] - t[1], L[2] - t[2]};
        const double pc[3] = {R[0] * d[0] + R[3] * d[1] + R[6] * d[2], R[1] * d[0] + R[4] * d[1] + R[7] * d[2], R[2] * d[0] + R[5] * d[1] + R[8] * d[2]};
        const double zi = 1.0 / pc[2];
        r[0] = pc[0] * zi - fx_; r[1] = pc[1] * zi - fy_;
        if (!J) return true;
        const double A[6] = {zi, 0, -pc[0] * zi * zi, 0, zi, -pc[1] * zi * zi};
        if (J[1] || J[2])
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 3; ++j) {
                    double s = 0;
                    for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * R[j * 3 + k];      // (A R^T)_ij
                    if (J[1]) J[1][i * 3 + j] = -s;
                    if (J[2]) J[2][i * 3 + j] = s;
                }
        if (J[0]) {
            // d pc / d q for pc = R(q)^T d with the (non-normalised) polynomial form of R; rows of 3x4
            const double dx = d[0], dy = d[1], dz = d[2];
            const double D[12] = {
                2 * (y * dy + z * dz),           2 * (-2 * y * dx + x * dy - w * dz), 2 * (-2 * z * dx + w * dy + x * dz), 2 * (z * dy - y * dz),
                2 * (y * dx - 2 * x * dy + w * dz), 2 * (x * dx + z * dz),            2 * (-w * dx - 2 * z * dy + y * dz), 2 * (-z * dx + x * dz),
                2 * (z * dx - w * dy - 2 * x * dz), 2 * (w * dx + z * dy - 2 * y * dz), 2 * (x * dx + y * dy),             2 * (y * dx - x * dy)};
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 4; ++j) {
                    double s = 0;
                    for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * D[k * 4 + j];
                    J[0][i * 4 + j] = s;
                }
        }
        return true;
    }
private:
    double fx_, fy_;
};

}  // namespace

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: %s scene.bin iterations threads\n", argv[0]); return 2; }
    std::ifstream f(argv[1], std::ios::binary);
    int h[3];
    if (!f.read(reinterpret_cast<char*>(h), sizeof h)) return 2;
    const int nc = h[0], np = h[1], no = h[2];
    std::vector<double> cams((size_t)nc * 7), pts((size_t)np * 3), feat((size_t)no * 2);
    std::vector<int> oc(no), op(no);
    std::vector<unsigned char> fixed(nc);
    f.read(reinterpret_cast<char*>(cams.data()), cams.size() * 8); f.read(reinterpret_cast<char*>(pts.data()), pts.size() * 8);
    f.read(reinterpret_cast<char*>(oc.data()), (size_t)no * 4); f.read(reinterpret_cast<char*>(op.data()), (size_t)no * 4);
    f.read(reinterpret_cast<char*>(feat.data()), feat.size() * 8); f.read(reinterpret_cast<char*>(fixed.data()), nc);
    if (!f) return 2;
    const int iters = std::atoi(argv[2]), threads = std::atoi(argv[3]);
    ceres::Problem problem;
    for (int i = 0; i < no; ++i) {
        double* q = &cams[(size_t)oc[i] * 7];
        problem.AddResidualBlock(new Reprojection(feat[2 * (size_t)i], feat[2 * (size_t)i + 1]), nullptr, q, q + 4, &pts[(size_t)op[i] * 3]);
    }
    for (int c = 0; c < nc; ++c) {
        double* q = &cams[(size_t)c * 7];
        if (!problem.HasParameterBlock(q)) continue;
        SET_MANIFOLD(problem, q, new So3RightManifold());
        if (fixed[c]) { problem.SetParameterBlockConstant(q); problem.SetParameterBlockConstant(q + 4); }
    }
    ceres::Solver::Options options;
    options.num_threads = threads;
    options.linear_solver_type = ceres::SPARSE_SCHUR;
    options.max_num_iterations = iters;
    options.function_tolerance = 0; options.gradient_tolerance = 0; options.parameter_tolerance = 0;     // fixed work
    ceres::Solver::Summary summary;
    const auto t0 = std::chrono::steady_clock::now();
    ceres::Solve(options, &problem, &summary);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const int done = (int)summary.iterations.size() - 1;
    std::printf("{\"ceres_version\": \"%s\", \"threads\": %d, \"iterations\": %d, \"seconds\": %.6f, \"iterations_per_sec\": %.6f, "
                "\"initial_cost\": %.17g, \"final_cost\": %.17g}\n", CERES_VERSION_STRING, threads, done, dt, done > 0 ? done / dt : 0.0,
                summary.initial_cost, summary.final_cost);
    return 0;
}

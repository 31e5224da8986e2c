// stba/ceres.h -- the reference's operator API (the slice of the Ceres Solver C++ API that
// Unsigned-Long/slam-tricks uses) as a header-only C++17 layer over the C ABI in stba.h.
//
// A maintainer of the reference switches  #include "ceres/ceres.h"  to
//     #include "stba/ceres.h"
//     namespace ceres = stba_ceres;
// and links libstba.so; call sites keep their shape (INTEGRATION.md shows the diff).
//
// Entry points mirrored (paths relative to /root/reference):
//   CostFunction::Evaluate, SizedCostFunction<2,3,3>        st17-ceres/src/include/solver.hpp:157-212
//   AutoDiffCostFunction<F,2,4,3>, <F,2,3>                  solver.hpp:135, st20-g2o/src/include/sim_data.h:175
//   DynamicAutoDiffCostFunction<F> (+AddParameterBlock/SetNumResiduals)
//                                                           solver.hpp:104,263-265, test_ceres.h:56,112-115,
//                                                           st17-ceres/src/ceres_bound.cpp:14,29-30
//   LocalParameterization {Plus, ComputeJacobian, GlobalSize, LocalSize}
//                                                           solver.hpp:30-94, test_ceres.h:14-45
//   Problem::{AddResidualBlock, AddParameterBlock, SetParameterBlockConstant,
//             SetParameterLowerBound, SetParameterUpperBound}
//                                                           solver.hpp:267-270, test_ceres.h:119-130,
//                                                           ceres_bound.cpp:32,51-53
//   Solver::Options / Summary::BriefReport / Solve / IterationCallback
//                                                           solver.hpp:215-290, test_ceres.h:83-151
//
// Where the work runs.  User CostFunction::Evaluate / autodiff functors are host code and are
// evaluated on the host (as in the reference); everything else -- normal equations, Schur
// complement, Cholesky, LM step -- runs in libstba's HIP kernels:
//   * residual blocks that ARE the reprojection factor of test_ceres.h:47-81 -- either the built-in
//     stba_ceres::ReprojectionFactor or ANY user cost function with blocks {4, 3, 3} -> 2 residuals
//     (the reference's own ns_st20::ProjectFactor behind DynamicAutoDiffCostFunction, test_ceres.h:111-121)
//     that is recognised numerically (internal::ProbeReprojection: its value equals
//     proj(R^T (L - t)) - feature at generic points, its Jet Jacobian equals the closed form)
//     -> fully device-resident BA engine (stba_ba_*): residuals and Jacobians are computed by the
//     HIP kernel too, the user functor is never called during the solve;
//   * a BA-SHAPED problem (every block {4, 3, 3} -> 2, quaternion chart) with any OTHER factor -> the device engine with the
//     user's cost functions evaluated on the host in bulk ("gpu-ba-hostjac", stba_ba_set_host_linearizer);
//   * any other problem -> the callback path of stba_dense_solve (<= 4096 local parameters).
// There is no CPU solver behind this header: without a HIP device Solve() reports FAILURE.
#ifndef STBA_CERES_H
#define STBA_CERES_H

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <thread>
#include <typeindex>
#include <typeinfo>
#include <cstdint>
#include <unordered_map>
#include <vector>

#include "../stba.h"

namespace stba_ceres {

// ------------------------------------------------------------------------------------------
// forward-mode dual numbers (ceres::Jet)
// ------------------------------------------------------------------------------------------
template <typename T, int N>
struct Jet {
    T a;
    T v[N];
    Jet() : a(T(0)) { for (int i = 0; i < N; ++i) v[i] = T(0); }
    Jet(const T& value) : a(value) { for (int i = 0; i < N; ++i) v[i] = T(0); }   // NOLINT
    Jet(const T& value, int k) : a(value) { for (int i = 0; i < N; ++i) v[i] = T(0); v[k] = T(1); }
    Jet& operator+=(const Jet& y) { *this = *this + y; return *this; }
    Jet& operator-=(const Jet& y) { *this = *this - y; return *this; }
    Jet& operator*=(const Jet& y) { *this = *this * y; return *this; }
    Jet& operator/=(const Jet& y) { *this = *this / y; return *this; }
};
#define STBA_JET_BIN(op, expr_a, expr_v)                                                         \
    template <typename T, int N> inline Jet<T, N> operator op(const Jet<T, N>& f, const Jet<T, N>& g) { \
        Jet<T, N> h; h.a = expr_a; for (int i = 0; i < N; ++i) h.v[i] = expr_v; return h; }
STBA_JET_BIN(+, f.a + g.a, f.v[i] + g.v[i])
STBA_JET_BIN(-, f.a - g.a, f.v[i] - g.v[i])
STBA_JET_BIN(*, f.a * g.a, f.a * g.v[i] + f.v[i] * g.a)
STBA_JET_BIN(/, f.a / g.a, (f.v[i] - (f.a / g.a) * g.v[i]) / g.a)
#undef STBA_JET_BIN
template <typename T, int N> inline Jet<T, N> operator-(const Jet<T, N>& f) { Jet<T, N> h; h.a = -f.a; for (int i = 0; i < N; ++i) h.v[i] = -f.v[i]; return h; }
template <typename T, int N> inline Jet<T, N> operator+(const Jet<T, N>& f, T s) { Jet<T, N> h = f; h.a += s; return h; }
template <typename T, int N> inline Jet<T, N> operator+(T s, const Jet<T, N>& f) { return f + s; }
template <typename T, int N> inline Jet<T, N> operator-(const Jet<T, N>& f, T s) { Jet<T, N> h = f; h.a -= s; return h; }
template <typename T, int N> inline Jet<T, N> operator-(T s, const Jet<T, N>& f) { return (-f) + s; }
template <typename T, int N> inline Jet<T, N> operator*(const Jet<T, N>& f, T s) { Jet<T, N> h; h.a = f.a * s; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * s; return h; }
template <typename T, int N> inline Jet<T, N> operator*(T s, const Jet<T, N>& f) { return f * s; }
template <typename T, int N> inline Jet<T, N> operator/(const Jet<T, N>& f, T s) { return f * (T(1) / s); }
template <typename T, int N> inline Jet<T, N> operator/(T s, const Jet<T, N>& g) { return Jet<T, N>(s) / g; }
template <typename T, int N> inline bool operator<(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a < g.a; }
template <typename T, int N> inline bool operator>(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a > g.a; }
template <typename T, int N> inline bool operator<(const Jet<T, N>& f, T g) { return f.a < g; }
template <typename T, int N> inline bool operator>(const Jet<T, N>& f, T g) { return f.a > g; }
template <typename T, int N> inline Jet<T, N> sqrt(const Jet<T, N>& f) { Jet<T, N> h; h.a = std::sqrt(f.a); const T d = T(0.5) / h.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * d; return h; }
template <typename T, int N> inline Jet<T, N> sin(const Jet<T, N>& f) { Jet<T, N> h; h.a = std::sin(f.a); const T d = std::cos(f.a); for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * d; return h; }
template <typename T, int N> inline Jet<T, N> cos(const Jet<T, N>& f) { Jet<T, N> h; h.a = std::cos(f.a); const T d = -std::sin(f.a); for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * d; return h; }
template <typename T, int N> inline Jet<T, N> atan2(const Jet<T, N>& y, const Jet<T, N>& x) { Jet<T, N> h; h.a = std::atan2(y.a, x.a); const T d = T(1) / (x.a * x.a + y.a * y.a); for (int i = 0; i < N; ++i) h.v[i] = (x.a * y.v[i] - y.a * x.v[i]) * d; return h; }
template <typename T, int N> inline Jet<T, N> abs(const Jet<T, N>& f) { return f.a < T(0) ? -f : f; }
using std::sqrt; using std::sin; using std::cos; using std::atan2; using std::abs;

// ------------------------------------------------------------------------------------------
// enums / small types
// ------------------------------------------------------------------------------------------
enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR,
                        ITERATIVE_SCHUR, CGNR };
enum CallbackReturnType { SOLVER_CONTINUE, SOLVER_ABORT, SOLVER_TERMINATE_SUCCESSFULLY };
enum TerminationType { CONVERGENCE, NO_CONVERGENCE, FAILURE, USER_SUCCESS, USER_FAILURE };
enum Ownership { DO_NOT_TAKE_OWNERSHIP, TAKE_OWNERSHIP };
constexpr int DYNAMIC = -1;

// The reference only passes nullptr (test_ceres.h:120, solver.hpp:267).  Robust losses are NOT implemented: a Problem that holds a
// residual block with a non-null LossFunction is REFUSED by Solve() (FAILURE, parameters untouched, the reason in Summary::message and
// on stderr) instead of being solved unweighted.
class LossFunction { public: virtual ~LossFunction() = default; };

struct IterationSummary {
    int iteration = 0;
    bool step_is_valid = false, step_is_successful = false;
    double cost = 0, cost_change = 0, gradient_max_norm = 0, step_norm = 0, relative_decrease = 0,
           trust_region_radius = 0;
};

class IterationCallback {
public:
    virtual ~IterationCallback() = default;
    virtual CallbackReturnType operator()(const IterationSummary& summary) = 0;
};

// ------------------------------------------------------------------------------------------
// CostFunction family
// ------------------------------------------------------------------------------------------
class CostFunction {
public:
    virtual ~CostFunction() = default;
    virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const = 0;
    const std::vector<int>& parameter_block_sizes() const { return parameter_block_sizes_; }
    int num_residuals() const { return num_residuals_; }
protected:
    std::vector<int>* mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
    void set_num_residuals(int n) { num_residuals_ = n; }
private:
    friend class Problem;
    std::vector<int> parameter_block_sizes_;
    int num_residuals_ = 0;
    bool owned_by_problem_ = false;      // set by the Problem that will delete it: a cost function shared by residual blocks is freed once
};

template <int kNumResiduals, int... Ns>
class SizedCostFunction : public CostFunction {
public:
    SizedCostFunction() {
        set_num_residuals(kNumResiduals);
        *mutable_parameter_block_sizes() = std::vector<int>{Ns...};
    }
};

namespace internal {
template <int... Ns> struct Sum;
template <> struct Sum<> { static constexpr int value = 0; };
template <int N, int... Ns> struct Sum<N, Ns...> { static constexpr int value = N + Sum<Ns...>::value; };

// calls functor(p0, p1, ..., residuals) with the parameter pointers unpacked
template <typename F, typename T, size_t... I>
inline bool CallVariadic(const F& f, T const* const* params, T* residuals, std::index_sequence<I...>) {
    return f(params[I]..., residuals);
}
}  // namespace internal

template <typename CostFunctor, int kNumResiduals, int... Ns>
class AutoDiffCostFunction : public SizedCostFunction<kNumResiduals, Ns...> {
public:
    explicit AutoDiffCostFunction(CostFunctor* functor, Ownership own = TAKE_OWNERSHIP) : functor_(functor), own_(own) {}
    ~AutoDiffCostFunction() override { if (own_ == TAKE_OWNERSHIP) delete functor_; }
    bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
        constexpr int kBlocks = sizeof...(Ns);
        constexpr int kParams = internal::Sum<Ns...>::value;
        if (!jacobians) return internal::CallVariadic(*functor_, parameters, residuals, std::make_index_sequence<kBlocks>{});
        using JetT = Jet<double, kParams>;
        const int sizes[kBlocks] = {Ns...};
        JetT x[kParams];
        JetT const* ptrs[kBlocks];
        int off = 0;
        for (int b = 0; b < kBlocks; ++b) {
            ptrs[b] = x + off;
            for (int k = 0; k < sizes[b]; ++k) x[off + k] = JetT(parameters[b][k], off + k);
            off += sizes[b];
        }
        JetT out[kNumResiduals];
        if (!internal::CallVariadic(*functor_, ptrs, out, std::make_index_sequence<kBlocks>{})) return false;
        off = 0;
        for (int b = 0; b < kBlocks; ++b) {
            for (int r = 0; r < kNumResiduals; ++r) {
                if (b == 0) residuals[r] = out[r].a;
                if (jacobians[b])
                    for (int k = 0; k < sizes[b]; ++k) jacobians[b][r * sizes[b] + k] = out[r].v[off + k];
            }
            off += sizes[b];
        }
        return true;
    }
private:
    CostFunctor* functor_;
    Ownership own_;
};

template <typename CostFunctor, int Stride = 4>
class DynamicAutoDiffCostFunction : public CostFunction {
public:
    explicit DynamicAutoDiffCostFunction(CostFunctor* functor, Ownership own = TAKE_OWNERSHIP) : functor_(functor), own_(own) {}
    ~DynamicAutoDiffCostFunction() override { if (own_ == TAKE_OWNERSHIP) delete functor_; }
    void AddParameterBlock(int size) {
        auto* v = mutable_parameter_block_sizes();
        if (v->capacity() == 0) v->reserve(4);       // (one allocation for the usual two to four blocks instead of one per doubling: 10^6 cost functions at C5)
        v->push_back(size);
    }
    void SetNumResiduals(int n) { set_num_residuals(n); }
    bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
        const auto& sizes = parameter_block_sizes();
        const int nb = (int)sizes.size(), nr = num_residuals();
        if (!jacobians) return (*functor_)(parameters, residuals);
        using JetT = Jet<double, Stride>;
        int total = 0;
        for (int s : sizes) total += s;
        // small problems (the reference's: 7-10 parameters, 2 residuals) work on the stack: no allocation per evaluation
        constexpr int kStackParams = 32, kStackRes = 8, kStackBlocks = 8;
        JetT x_stack[kStackParams], out_stack[kStackRes];
        JetT const* ptrs_stack[kStackBlocks];
        int blk_stack[kStackParams], loc_stack[kStackParams];
        std::vector<JetT> x_heap, out_heap;
        std::vector<JetT const*> ptrs_heap;
        std::vector<int> blk_heap, loc_heap;
        const bool on_stack = total <= kStackParams && nr <= kStackRes && nb <= kStackBlocks;
        if (!on_stack) { x_heap.resize(total); out_heap.resize(nr); ptrs_heap.resize(nb); blk_heap.resize(total); loc_heap.resize(total); }
        JetT* x = on_stack ? x_stack : x_heap.data();
        JetT* out = on_stack ? out_stack : out_heap.data();
        JetT const** ptrs = on_stack ? ptrs_stack : ptrs_heap.data();
        int* blk = on_stack ? blk_stack : blk_heap.data();
        int* loc = on_stack ? loc_stack : loc_heap.data();
        for (int b = 0, o = 0; b < nb; ++b) {
            ptrs[b] = x + o;
            for (int k = 0; k < sizes[b]; ++k, ++o) { blk[o] = b; loc[o] = k; }
        }
        // ceil(total / Stride) passes, Stride partial derivatives per pass (as Ceres does)
        for (int start = 0; start < total || start == 0; start += Stride) {
            for (int b = 0, o = 0; b < nb; ++b)
                for (int k = 0; k < sizes[b]; ++k, ++o) {
                    x[o] = JetT(parameters[b][k]);
                    if (o >= start && o < start + Stride) x[o].v[o - start] = 1.0;
                }
            if (!(*functor_)(ptrs, out)) return false;
            for (int r = 0; r < nr; ++r) {
                residuals[r] = out[r].a;
                for (int o = start; o < std::min(total, start + Stride); ++o)
                    if (jacobians[blk[o]]) jacobians[blk[o]][r * sizes[blk[o]] + loc[o]] = out[r].v[o - start];
            }
            if (total == 0) break;
        }
        return true;
    }
private:
    CostFunctor* functor_;
    Ownership own_;
};

// ------------------------------------------------------------------------------------------
// LocalParameterization (Ceres <= 2.1 API, as the reference uses it)
// ------------------------------------------------------------------------------------------
class LocalParameterization {
public:
    virtual ~LocalParameterization() = default;
    virtual bool Plus(const double* x, const double* delta, double* x_plus_delta) const = 0;
    virtual bool ComputeJacobian(const double* x, double* jacobian) const = 0;   // GlobalSize x LocalSize, row-major
    virtual int GlobalSize() const = 0;
    virtual int LocalSize() const = 0;
};

// ------------------------------------------------------------------------------------------
// built-in factor / manifold kinds the HIP kernels implement natively
// ------------------------------------------------------------------------------------------
// q (x,y,z,w) <- q (x) exp(delta): LieLocalParameterization<Sophus::SO3d> (solver.hpp:30-61)
class QuaternionRightPlus : public LocalParameterization {
public:
    bool Plus(const double* q, const double* d, double* out) const override {
        const double th2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        double im, re;
        if (th2 < 1e-20) { im = 0.5 - th2 / 48.0; re = 1.0 - th2 / 8.0; }
        else { const double th = std::sqrt(th2); im = std::sin(0.5 * th) / th; re = std::cos(0.5 * th); }
        const double bx = im * d[0], by = im * d[1], bz = im * d[2], bw = re;
        const double ax = q[0], ay = q[1], az = q[2], aw = q[3];
        double o[4] = {aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                       aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz};
        const double n = std::sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
        for (int i = 0; i < 4; ++i) out[i] = o[i] / n;
        return true;
    }
    bool ComputeJacobian(const double* q, double* J) const override {   // notes.tex:131-144
        const double x = 0.5 * q[0], y = 0.5 * q[1], z = 0.5 * q[2], w = 0.5 * q[3];
        const double M[12] = {w, -z, y, z, w, -x, -y, x, w, -x, -y, -z};
        std::memcpy(J, M, sizeof M);
        return true;
    }
    int GlobalSize() const override { return 4; }
    int LocalSize() const override { return 3; }
};

// ns_st20::ProjectFactor (test_ceres.h:47-81) as a built-in: blocks {SO3 quaternion 4, POS 3,
// landmark 3}, residual = proj(R^T (L - t)) - feature.  Evaluate() is provided for completeness
// (ambient 2x4 quaternion Jacobian), but a Problem made only of these runs fully on the GPU.
class ReprojectionFactor : public SizedCostFunction<2, 4, 3, 3> {
public:
    ReprojectionFactor(double fx, double fy) : fx_(fx), fy_(fy) {}
    static ReprojectionFactor* Create(const double* feature) { return new ReprojectionFactor(feature[0], feature[1]); }
    double fx() const { return fx_; }
    double fy() const { return fy_; }
    template <typename T>
    bool operator()(const T* q, const T* t, const T* L, T* r) const {
        // conj(q) * (L - t) with Eigen's v + 2w(u x v) + 2 u x (u x v), u = -q.xyz
        const T u0 = -q[0], u1 = -q[1], u2 = -q[2], w = q[3];
        const T v0 = L[0] - t[0], v1 = L[1] - t[1], v2 = L[2] - t[2];
        const T a0 = T(2.0) * (u1 * v2 - u2 * v1), a1 = T(2.0) * (u2 * v0 - u0 * v2), a2 = T(2.0) * (u0 * v1 - u1 * v0);
        const T x = v0 + w * a0 + (u1 * a2 - u2 * a1);
        const T y = v1 + w * a1 + (u2 * a0 - u0 * a2);
        const T z = v2 + w * a2 + (u0 * a1 - u1 * a0);
        r[0] = x / z - T(fx_);
        r[1] = y / z - T(fy_);
        return true;
    }
    bool Evaluate(double const* const* p, double* residuals, double** jacobians) const override {
        if (!jacobians) return (*this)(p[0], p[1], p[2], residuals);
        using J10 = Jet<double, 10>;
        J10 x[10], out[2];
        for (int k = 0; k < 4; ++k) x[k] = J10(p[0][k], k);
        for (int k = 0; k < 3; ++k) { x[4 + k] = J10(p[1][k], 4 + k); x[7 + k] = J10(p[2][k], 7 + k); }
        (*this)(x, x + 4, x + 7, out);
        for (int r = 0; r < 2; ++r) {
            residuals[r] = out[r].a;
            if (jacobians[0]) for (int k = 0; k < 4; ++k) jacobians[0][r * 4 + k] = out[r].v[k];
            if (jacobians[1]) for (int k = 0; k < 3; ++k) jacobians[1][r * 3 + k] = out[r].v[4 + k];
            if (jacobians[2]) for (int k = 0; k < 3; ++k) jacobians[2][r * 3 + k] = out[r].v[7 + k];
        }
        return true;
    }
private:
    double fx_, fy_;
};

// ------------------------------------------------------------------------------------------
// Pose graph through the same operator API (BASELINE config C4; build-defined: the reference has no pose-graph code, the
// conventions are its Lie-group notes', st23-lie-group-v2/doc.tex:862-996).  A pose is ONE parameter block of 7 doubles
// (qx qy qz qw tx ty tz); SE3RightPlus is its chart T <- T exp(delta), delta = [rho, theta] (Sophus' SE3 tangent order);
// RelativePoseFactor is the edge r = log(Z^-1 T_i^-1 T_j) in R^6.  A Problem made only of these runs on the device pose-graph
// engine (stba_pg_*: "gpu-pg"); Evaluate / Plus / ComputeJacobian below are what the generic host path uses.
// ------------------------------------------------------------------------------------------
namespace se3 {
template <typename T> inline void QuatMul(const T* a, const T* b, T* o) {
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
template <typename T> inline void QuatRotate(const T* q, const T* v, T* o) {     // R(q) v = v + 2w (u x v) + 2 u x (u x v)
    const T u0 = q[0], u1 = q[1], u2 = q[2], w = q[3];
    const T a0 = T(2.0) * (u1 * v[2] - u2 * v[1]), a1 = T(2.0) * (u2 * v[0] - u0 * v[2]), a2 = T(2.0) * (u0 * v[1] - u1 * v[0]);
    o[0] = v[0] + w * a0 + (u1 * a2 - u2 * a1);
    o[1] = v[1] + w * a1 + (u2 * a0 - u0 * a2);
    o[2] = v[2] + w * a2 + (u0 * a1 - u1 * a0);
}
template <typename T> inline void Inverse(const T* a, T* o) {
    const T qc[4] = {-a[0], -a[1], -a[2], a[3]};
    T t[3];
    QuatRotate(qc, a + 4, t);
    for (int i = 0; i < 4; ++i) o[i] = qc[i];
    for (int i = 0; i < 3; ++i) o[4 + i] = -t[i];
}
template <typename T> inline void Compose(const T* a, const T* b, T* o) {
    T q[4], t[3];
    QuatMul(a, b, q);
    QuatRotate(a, b + 4, t);
    for (int i = 0; i < 4; ++i) o[i] = q[i];
    for (int i = 0; i < 3; ++i) o[4 + i] = t[i] + a[4 + i];
}
// Sophus SE3::log of (q, t) -> [rho, theta]
template <typename T> inline void Log(const T* P, T* xi) {
    const T n2 = P[0] * P[0] + P[1] * P[1] + P[2] * P[2], qw = P[3];
    T k;
    if (n2 < T(1e-20)) k = T(2.0) / qw - T(2.0 / 3.0) * n2 / (qw * qw * qw);
    else { const T n = sqrt(n2); k = T(2.0) * ((qw < T(0.0)) ? atan2(-n, -qw) : atan2(n, qw)) / n; }
    const T w[3] = {k * P[0], k * P[1], k * P[2]};
    const T th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    T a, b;
    if (th2 < T(1e-20)) { a = T(0.5) - th2 / T(24.0); b = T(1.0 / 6.0) - th2 / T(120.0); }
    else { const T th = sqrt(th2); a = (T(1.0) - cos(th)) / th2; b = (th - sin(th)) / (th2 * th); }
    const T K[9] = {T(0.0), -w[2], w[1], w[2], T(0.0), -w[0], -w[1], w[0], T(0.0)};
    T V[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const T k2 = K[i * 3] * K[j] + K[i * 3 + 1] * K[3 + j] + K[i * 3 + 2] * K[6 + j];
            V[i * 3 + j] = (i == j ? T(1.0) : T(0.0)) + a * K[i * 3 + j] + b * k2;
        }
    const T aa = V[0], bb = V[1], cc = V[2], dd = V[3], ee = V[4], ff = V[5], gg = V[6], hh = V[7], ii = V[8];
    const T C0 = ee * ii - ff * hh, C1 = ff * gg - dd * ii, C2 = dd * hh - ee * gg;
    const T inv = T(1.0) / (aa * C0 + bb * C1 + cc * C2);
    const T* t = P + 4;
    xi[0] = inv * (C0 * t[0] + (cc * hh - bb * ii) * t[1] + (bb * ff - cc * ee) * t[2]);
    xi[1] = inv * (C1 * t[0] + (aa * ii - cc * gg) * t[1] + (cc * dd - aa * ff) * t[2]);
    xi[2] = inv * (C2 * t[0] + (bb * gg - aa * hh) * t[1] + (aa * ee - bb * dd) * t[2]);
    xi[3] = w[0]; xi[4] = w[1]; xi[5] = w[2];
}
// Sophus SE3::exp of [rho, theta] -> (q, t)
inline void Exp(const double* d, double* P) {
    const double* th = d + 3;
    const double th2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
    double im, re, a, b;
    if (th2 < 1e-20) { im = 0.5 - th2 / 48.0; re = 1.0 - th2 / 8.0; a = 0.5 - th2 / 24.0; b = 1.0 / 6.0 - th2 / 120.0; }
    else { const double t = std::sqrt(th2); im = std::sin(0.5 * t) / t; re = std::cos(0.5 * t); a = (1.0 - std::cos(t)) / th2; b = (t - std::sin(t)) / (th2 * t); }
    P[0] = im * th[0]; P[1] = im * th[1]; P[2] = im * th[2]; P[3] = re;
    const double K[9] = {0, -th[2], th[1], th[2], 0, -th[0], -th[1], th[0], 0};
    for (int i = 0; i < 3; ++i) {
        double s = 0.0;
        for (int j = 0; j < 3; ++j) {
            const double k2 = K[i * 3] * K[j] + K[i * 3 + 1] * K[3 + j] + K[i * 3 + 2] * K[6 + j];
            s += ((i == j ? 1.0 : 0.0) + a * K[i * 3 + j] + b * k2) * d[j];
        }
        P[4 + i] = s;
    }
}
}  // namespace se3

class SE3RightPlus : public LocalParameterization {
public:
    bool Plus(const double* x, const double* d, double* out) const override {
        double e[7], o[7];
        se3::Exp(d, e);
        se3::Compose(x, e, o);
        const double n = std::sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
        for (int i = 0; i < 4; ++i) out[i] = o[i] / n;
        for (int i = 0; i < 3; ++i) out[4 + i] = o[4 + i];
        return true;
    }
    // d (x (+) delta) / d delta at 0, 7 x 6 row-major: the quaternion moves with theta only (the 4 x 3 block of
    // QuaternionRightPlus), the translation with rho only (R(q))
    bool ComputeJacobian(const double* x, double* J) const override {
        std::fill(J, J + 42, 0.0);
        const double qx = 0.5 * x[0], qy = 0.5 * x[1], qz = 0.5 * x[2], qw = 0.5 * x[3];
        const double M[12] = {qw, -qz, qy, qz, qw, -qx, -qy, qx, qw, -qx, -qy, -qz};
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 3; ++c) J[r * 6 + 3 + c] = M[r * 3 + c];
        for (int c = 0; c < 3; ++c) {
            double e[3] = {0, 0, 0}, col[3];
            e[c] = 1.0;
            se3::QuatRotate(x, e, col);
            for (int r = 0; r < 3; ++r) J[(4 + r) * 6 + c] = col[r];
        }
        return true;
    }
    int GlobalSize() const override { return 7; }
    int LocalSize() const override { return 6; }
};

class RelativePoseFactor : public SizedCostFunction<6, 7, 7> {
public:
    explicit RelativePoseFactor(const double* measurement) { std::memcpy(z_, measurement, sizeof z_); }
    static RelativePoseFactor* Create(const double* measurement) { return new RelativePoseFactor(measurement); }
    const double* measurement() const { return z_; }
    template <typename T>
    bool operator()(const T* Ti, const T* Tj, T* r) const {
        T Z[7], Zi[7], Tii[7], A[7], E[7];
        for (int k = 0; k < 7; ++k) Z[k] = T(z_[k]);
        se3::Inverse(Z, Zi);
        se3::Inverse(Ti, Tii);
        se3::Compose(Tii, Tj, A);
        se3::Compose(Zi, A, E);
        if (E[3] < T(0.0)) for (int k = 0; k < 4; ++k) E[k] = -E[k];       // shortest rotation
        se3::Log(E, r);
        return true;
    }
    bool Evaluate(double const* const* p, double* residuals, double** jacobians) const override {
        if (!jacobians) return (*this)(p[0], p[1], residuals);
        using J14 = Jet<double, 14>;
        J14 x[14], out[6];
        for (int k = 0; k < 7; ++k) { x[k] = J14(p[0][k], k); x[7 + k] = J14(p[1][k], 7 + k); }
        (*this)(x, x + 7, out);
        for (int r = 0; r < 6; ++r) {
            residuals[r] = out[r].a;
            if (jacobians[0]) for (int k = 0; k < 7; ++k) jacobians[0][r * 7 + k] = out[r].v[k];
            if (jacobians[1]) for (int k = 0; k < 7; ++k) jacobians[1][r * 7 + k] = out[r].v[7 + k];
        }
        return true;
    }
private:
    double z_[7];
};

// ------------------------------------------------------------------------------------------
// Problem
// ------------------------------------------------------------------------------------------
class Problem {
public:
    struct Options {
        Ownership cost_function_ownership = TAKE_OWNERSHIP, loss_function_ownership = TAKE_OWNERSHIP,
                  local_parameterization_ownership = TAKE_OWNERSHIP;
    };
    Problem() = default;
    explicit Problem(const Options& o) : options_(o) {}
    Problem(const Problem&) = delete;
    Problem& operator=(const Problem&) = delete;
    ~Problem() {
        // the reference never frees what it news (solver.hpp:104,258; test_ceres.h:56,106): the problem
        // owns cost functions and parameterisations, shared pointers are freed once.
        if (options_.cost_function_ownership == TAKE_OWNERSHIP) for (auto* c : owned_costs_) delete c;
        if (options_.local_parameterization_ownership == TAKE_OWNERSHIP) for (auto* l : owned_params_) delete l;
        if (options_.loss_function_ownership == TAKE_OWNERSHIP) for (auto* l : owned_losses_) delete l;
    }

    void AddParameterBlock(double* values, int size, LocalParameterization* local = nullptr) {
        Block& b = block(values, size);
        if (local) { b.local = local; owned_params_.insert(local); }
    }
    template <typename... Ts>
    void AddResidualBlock(CostFunction* cost, LossFunction* loss, double* x0, Ts*... xs) {
        double* const b[] = {x0, xs...};
        AddResidualBlock(cost, loss, b, 1 + sizeof...(xs));
    }
    void AddResidualBlock(CostFunction* cost, LossFunction* loss, std::initializer_list<double*> blocks) {
        AddResidualBlock(cost, loss, blocks.begin(), blocks.size());
    }
    void AddResidualBlock(CostFunction* cost, LossFunction* loss, const std::vector<double*>& blocks) {
        AddResidualBlock(cost, loss, blocks.data(), blocks.size());
    }
    void AddResidualBlock(CostFunction* cost, LossFunction* loss, double* const* blocks, size_t n_blocks) {
        Residual r;
        r.cost = cost;
        const auto& sizes = cost->parameter_block_sizes();
        if (sizes.size() != n_blocks) { std::fprintf(stderr, "stba_ceres: block count mismatch\n"); std::abort(); }
        r.blocks.pool = &block_pool_; r.blocks.off = (int)block_pool_.size(); r.blocks.n = (int)n_blocks;
        for (size_t i = 0; i < n_blocks; ++i) block_pool_.push_back(block(blocks[i], sizes[i]).index);
        residuals_.push_back(r);
        // (a cost function shared by several residual blocks is listed once: the mark lives in the cost function -- sorting 10^6
        // pointers at destruction was a fifth of the destructor's time)
        if (options_.cost_function_ownership == TAKE_OWNERSHIP && !cost->owned_by_problem_) { cost->owned_by_problem_ = true; owned_costs_.push_back(cost); }
        if (loss) { owned_losses_.insert(loss); ++num_loss_functions_; }
    }
    int NumLossFunctions() const { return num_loss_functions_; }   // residual blocks added with a non-null LossFunction (Solve refuses them)
    void SetParameterBlockConstant(double* values) { find(values).constant = true; }
    void SetParameterBlockVariable(double* values) { find(values).constant = false; }
    void SetParameterLowerBound(double* values, int index, double lower) { Block& b = find(values); b.ensure_bounds(); b.lower[index] = lower; }
    void SetParameterUpperBound(double* values, int index, double upper) { Block& b = find(values); b.ensure_bounds(); b.upper[index] = upper; }
    int NumParameterBlocks() const { return (int)blocks_.size(); }
    int NumResidualBlocks() const { return (int)residuals_.size(); }
    int NumResiduals() const { int n = 0; for (auto& r : residuals_) n += r.cost->num_residuals(); return n; }

    // ---- internals used by Solve ----
    struct Block {
        double* ptr = nullptr; int size = 0, index = 0; bool constant = false;
        LocalParameterization* local = nullptr;
        std::vector<double> lower, upper;
        int local_size() const { return local ? local->LocalSize() : size; }
        void ensure_bounds() { if (lower.empty()) { lower.assign(size, -1e300); upper.assign(size, 1e300); } }
    };
    // the parameter blocks of one residual block: a view into the problem's one index pool (a std::vector per residual block was
    // a heap allocation per observation: 10^6 of them at config C5)
    struct BlockList {
        const std::vector<int>* pool = nullptr; int off = 0, n = 0;
        size_t size() const { return (size_t)n; }
        int operator[](size_t i) const { return (*pool)[(size_t)off + i]; }
        const int* begin() const { return pool->data() + off; }
        const int* end() const { return pool->data() + off + n; }
    };
    struct Residual { CostFunction* cost = nullptr; BlockList blocks; };
    std::vector<Block>& blocks() { return blocks_; }
    std::vector<Residual>& residuals() { return residuals_; }

private:
    // parameter block pointer -> index: an open-addressing table (four look-ups per residual block at the reference's BA call site,
    // 4 x 10^6 at C5: std::unordered_map's node per key and its modulo were a third of the construction time)
    struct PtrIndex {
        std::vector<double*> keys; std::vector<int> vals; size_t used = 0; int shift = 64;
        static size_t mix(const double* p) { return (size_t)((reinterpret_cast<std::uintptr_t>(p) >> 3) * 0x9E3779B97F4A7C15ull); }
        void grow() {
            const size_t cap = keys.empty() ? 64 : keys.size() * 2;       // (small first: a PnP problem has two blocks)
            std::vector<double*> k2(cap, nullptr); std::vector<int> v2(cap, 0);
            int sh = 64; for (size_t c = cap; c > 1; c >>= 1) --sh;
            for (size_t q = 0; q < keys.size(); ++q) if (keys[q]) {
                size_t h = mix(keys[q]) >> sh;
                while (k2[h]) h = (h + 1) & (cap - 1);
                k2[h] = keys[q]; v2[h] = vals[q];
            }
            keys.swap(k2); vals.swap(v2); shift = sh;
        }
        int find(const double* p) const {
            if (keys.empty()) return -1;
            const size_t mask = keys.size() - 1;
            for (size_t h = mix(p) >> shift;; h = (h + 1) & mask) { if (keys[h] == p) return vals[h]; if (!keys[h]) return -1; }
        }
        void insert(double* p, int v) {
            if (2 * (used + 1) > keys.size()) grow();
            const size_t mask = keys.size() - 1;
            size_t h = mix(p) >> shift;
            while (keys[h]) h = (h + 1) & mask;
            keys[h] = p; vals[h] = v; ++used;
        }
    };
    Block& block(double* p, int size) {
        const int at = index_.find(p);
        if (at >= 0) {
            if (blocks_[(size_t)at].size != size) { std::fprintf(stderr, "stba_ceres: block re-added with another size\n"); std::abort(); }
            return blocks_[(size_t)at];
        }
        Block b; b.ptr = p; b.size = size; b.index = (int)blocks_.size();
        index_.insert(p, b.index);
        blocks_.push_back(b);
        return blocks_.back();
    }
    Block& find(double* p) {
        const int at = index_.find(p);
        if (at < 0) { std::fprintf(stderr, "stba_ceres: unknown parameter block\n"); std::abort(); }
        return blocks_[(size_t)at];
    }
    Options options_;
    std::vector<Block> blocks_;
    std::vector<Residual> residuals_;
    std::vector<int> block_pool_;
    PtrIndex index_;
    std::vector<CostFunction*> owned_costs_;       // (every cost function once: CostFunction::owned_by_problem_)
    int num_loss_functions_ = 0;
    std::set<LocalParameterization*> owned_params_;
    std::set<LossFunction*> owned_losses_;
};

// ------------------------------------------------------------------------------------------
// Solver
// ------------------------------------------------------------------------------------------
class Solver {
public:
    struct Options {
        int max_num_iterations = 50;
        int num_threads = 1;
        LinearSolverType linear_solver_type = SPARSE_NORMAL_CHOLESKY;
        bool minimizer_progress_to_stdout = false;
        bool update_state_every_iteration = false;
        std::vector<IterationCallback*> callbacks;
        double initial_trust_region_radius = 1e4, max_trust_region_radius = 1e16, min_trust_region_radius = 1e-32,
               min_relative_decrease = 1e-3, min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32,
               function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
        bool jacobi_scaling = true;
        bool force_callback_path = false;   // (not in Ceres) never replace user cost functions by the built-in device factor: see Solve()
    };
    struct Summary {
        TerminationType termination_type = FAILURE;
        std::string message;
        double initial_cost = 0, final_cost = 0;
        // Ceres' own timing fields: everything inside Solve() / in front of the minimiser / the minimiser / behind it
        double total_time_in_seconds = 0, preprocessor_time_in_seconds = 0, minimizer_time_in_seconds = 0, postprocessor_time_in_seconds = 0;
        // (not in Ceres) where the host side of the drop-in spends its time, seconds: recognise = DetectBa (structure, probes, every
        // user block at its start point); pack = parameters and masks into the engine's arrays; engine_create = stba_*_create (regrouping,
        // Schur plan, uploads); device_solve = stba_*_solve; write_back = parameters back into the user's blocks; verify = every
        // recognised user block at the end point; resolve = a second solve with the user's own code, if the verification failed
        struct Phases { double recognise = 0, pack = 0, engine_create = 0, device_solve = 0, write_back = 0, verify = 0, resolve = 0; } phases;
        int num_successful_steps = 0, num_unsuccessful_steps = 0;
        std::vector<IterationSummary> iterations;
        std::string execution_path;   // "gpu-ba" | "gpu-ba-hostjac" | "gpu-pg" | "gpu-dense-callback"
        std::string BriefReport() const {
            char buf[512];
            const char* t = termination_type == CONVERGENCE ? "CONVERGENCE" : termination_type == NO_CONVERGENCE ? "NO_CONVERGENCE"
                            : termination_type == FAILURE ? "FAILURE" : "USER";
            std::snprintf(buf, sizeof buf, "stba Solver Summary: Iterations: %d, Initial cost: %e, Final cost: %e, Termination: %s",
                          (int)iterations.size(), initial_cost, final_cost, t);
            return buf;
        }
        std::string FullReport() const { return BriefReport() + " [" + execution_path + "] " + message; }
    };
};

namespace internal {

inline stba_lm_options ToC(const Solver::Options& o) {
    stba_lm_options c;
    stba_lm_default_options(&c);
    c.max_num_iterations = o.max_num_iterations;
    c.initial_trust_region_radius = o.initial_trust_region_radius;
    c.max_trust_region_radius = o.max_trust_region_radius;
    c.min_trust_region_radius = o.min_trust_region_radius;
    c.min_relative_decrease = o.min_relative_decrease;
    c.min_lm_diagonal = o.min_lm_diagonal;
    c.max_lm_diagonal = o.max_lm_diagonal;
    c.function_tolerance = o.function_tolerance;
    c.gradient_tolerance = o.gradient_tolerance;
    c.parameter_tolerance = o.parameter_tolerance;
    c.jacobi_scaling = o.jacobi_scaling ? 1 : 0;
    c.num_threads = o.num_threads;
    c.minimizer_progress_to_stdout = o.minimizer_progress_to_stdout ? 1 : 0;
    c.update_state_every_iteration = o.update_state_every_iteration ? 1 : 0;
    return c;
}

struct CallbackCtx {
    const Solver::Options* options;
    Solver::Summary* summary;
    void (*sync_state)(void*);   // copies the current iterate into the user's parameter memory
    void* sync_user;
    bool user_abort = false, user_success = false;
};

inline int IterationTrampoline(void* user, int iteration, double cost, double cost_change, double gradient_max_norm,
                               double step_norm, double radius, int ok) {
    auto* ctx = static_cast<CallbackCtx*>(user);
    IterationSummary it;
    it.iteration = iteration; it.cost = cost; it.cost_change = cost_change; it.gradient_max_norm = gradient_max_norm;
    it.step_norm = step_norm; it.trust_region_radius = radius; it.step_is_valid = true; it.step_is_successful = ok != 0;
    if (!ctx->options->callbacks.empty() && ctx->options->update_state_every_iteration && ctx->sync_state)
        ctx->sync_state(ctx->sync_user);   // solver.hpp:233,238: the callback dereferences the live state
    for (auto* cb : ctx->options->callbacks) {
        const CallbackReturnType r = (*cb)(it);
        if (r == SOLVER_ABORT) { ctx->user_abort = true; return 1; }
        if (r == SOLVER_TERMINATE_SUCCESSFULLY) { ctx->user_success = true; return 1; }
    }
    return 0;
}

inline void FillSummary(const stba_lm_summary& s, const std::vector<double>& trace, Solver::Summary* out) {
    out->initial_cost = s.initial_cost; out->final_cost = s.final_cost; out->minimizer_time_in_seconds = s.seconds_total;
    out->num_successful_steps = s.num_successful_steps; out->num_unsuccessful_steps = s.num_unsuccessful_steps;
    out->termination_type = s.termination_type == STBA_CONVERGENCE ? CONVERGENCE
                            : s.termination_type == STBA_NO_CONVERGENCE ? NO_CONVERGENCE : FAILURE;
    if (s.termination_type == STBA_FAILURE && !std::isfinite(s.initial_cost))
        out->message = "Initial residual and Jacobian evaluation failed.";      // (Ceres' wording for a non-finite start point)
    out->iterations.clear();
    for (int i = 0; i <= s.num_iterations; ++i) {
        const double* t = &trace[(size_t)i * STBA_TRACE_COLS];
        IterationSummary it;
        it.iteration = i; it.cost = t[0]; it.cost_change = t[1]; it.gradient_max_norm = t[2]; it.step_norm = t[3];
        it.relative_decrease = t[4]; it.trust_region_radius = t[5]; it.step_is_successful = t[6] != 0; it.step_is_valid = true;
        out->iterations.push_back(it);
    }
}

// ---- recognising the reprojection factor behind an arbitrary CostFunction ---------------------
// The reference's BA call site (st20-g2o/src/include/test_ceres.h:109-121) builds every residual block
// from its OWN functor (ns_st20::ProjectFactor behind DynamicAutoDiffCostFunction, blocks 4/3/3 -> 2).
// Such a block runs on the device if it IS the built-in factor: r = proj(conj(q) (L - t)) - feature.
inline void ReprojectionAt(const double* q, const double* t, const double* L, double* proj, double* pc_out = nullptr) {
    const double u0 = -q[0], u1 = -q[1], u2 = -q[2], w = q[3];
    const double v0 = L[0] - t[0], v1 = L[1] - t[1], v2 = L[2] - t[2];
    const double a0 = 2.0 * (u1 * v2 - u2 * v1), a1 = 2.0 * (u2 * v0 - u0 * v2), a2 = 2.0 * (u0 * v1 - u1 * v0);
    const double x = v0 + w * a0 + (u1 * a2 - u2 * a1), y = v1 + w * a1 + (u2 * a0 - u0 * a2), z = v2 + w * a2 + (u0 * a1 - u1 * a0);
    proj[0] = x / z; proj[1] = y / z;
    if (pc_out) { pc_out[0] = x; pc_out[1] = y; pc_out[2] = z; }
}

// generic probe points (unit quaternions, landmark well in front of the camera)
struct ProbePoint { double q[4], t[3], L[3]; };
inline const ProbePoint* ProbePoints() {
    // (the third point has a different geometry: a rotation of more than 90 degrees and a landmark that projects far off
    // the optical axis, |x/z| ~ 1.9: a cost that clamps or guards large image coordinates differs here)
    // (the fourth point lies BEHIND the camera, z = -1.6 in the camera frame: the built-in factor divides by z whatever its sign,
    // as the reference's functor does (test_ceres.h:72-78); a cost with a cheirality guard differs here and keeps its own code --
    // on the host-linearised device path if the problem is BA-shaped)
    static const ProbePoint pts[4] = {
        {{0.18257418583505536, 0.3651483716701107, 0.5477225575051661, 0.7302967433402214}, {0.3, -0.2, 0.1}, {1.1, 0.7, 2.9}},
        {{-0.2721655269759087, 0.1360827634879543, 0.4082482904638630, 0.8606629658238704}, {-0.4, 0.25, -0.6}, {0.2, -0.9, 3.3}},
        {{0.0, 0.8, 0.0, 0.6}, {0.5, 0.1, -0.2}, {1.035, -0.775, -2.83}},
        {{0.0, 0.0, 0.0, 1.0}, {0.2, -0.1, 0.4}, {0.9, 0.3, -1.2}}};
    return pts;
}

// The recognition's value checks, per residual block.  `feature` is recovered at the canonical point (identity rotation, camera at
// the origin, landmark on the optical axis: proj = 0, so feature = -residual); the block must then equal proj - feature
//   * at four generic probe points -- the FIRST block of every C++ type (ProbeReprojectionValue; with the Jacobian check below),
//   * AT ITS OWN DATA -- the parameter values it holds when Solve is called and the values it holds when the solve ends -- EVERY
//     block (ReprojectionValueAtData): a block's own data is a generic point of its own (some rotation, some camera position, some
//     landmark), so a per-instance weight, intrinsic or clamp shows there.
// Doubles only, no Jacobians: two evaluations per block before the solve and one after it (round 6; five + one + one until round 5,
// which was 0.22 s of a C5 Solve()).
inline bool IsReprojectionShape(const CostFunction* cost) {
    const auto& sz = cost->parameter_block_sizes();
    return cost->num_residuals() == 2 && sz.size() == 3 && sz[0] == 4 && sz[1] == 3 && sz[2] == 3;
}
inline bool CanonicalFeature(const CostFunction* cost, double* feature) {
    const double q0[4] = {0, 0, 0, 1}, t0[3] = {0, 0, 0}, L0[3] = {0, 0, 1};
    const double* p0[3] = {q0, t0, L0};
    double r[2] = {0, 0};
    if (!cost->Evaluate(p0, r, nullptr) || !std::isfinite(r[0]) || !std::isfinite(r[1])) return false;
    feature[0] = -r[0]; feature[1] = -r[1];
    return true;
}
inline bool ProbeReprojectionValue(const CostFunction* cost, double* feature) {
    if (!IsReprojectionShape(cost) || !CanonicalFeature(cost, feature)) return false;
    const ProbePoint* pp = ProbePoints();
    double r[2];
    for (int k = 0; k < 4; ++k) {
        const double* p[3] = {pp[k].q, pp[k].t, pp[k].L};
        double proj[2];
        ReprojectionAt(pp[k].q, pp[k].t, pp[k].L, proj);
        if (!cost->Evaluate(p, r, nullptr)) return false;
        for (int i = 0; i < 2; ++i)
            if (!(std::fabs(r[i] - (proj[i] - feature[i])) <= 1e-12 * (1.0 + std::fabs(proj[i]) + std::fabs(feature[i])))) return false;
    }
    return true;
}

// value check of one residual block AT ITS OWN DATA.  The tolerance grows with |L - t| / |z|: near z = 0 the projection is
// ill-conditioned and two correct ways of computing it differ by more than 1e-11 (advisor, round 5).
inline bool ReprojectionValueAtData(const CostFunction* cost, const double* q, const double* t, const double* L, const double* feature) {
    const double* p[3] = {q, t, L};
    double r[2] = {0, 0}, proj[2], pc[3];
    if (!cost->Evaluate(p, r, nullptr)) return false;
    ReprojectionAt(q, t, L, proj, pc);
    const double amp = std::max(1.0, (std::fabs(pc[0]) + std::fabs(pc[1]) + std::fabs(pc[2])) / std::fabs(pc[2]));
    for (int i = 0; i < 2; ++i)
        if (!(std::fabs(r[i] - (proj[i] - feature[i])) <= 1e-11 * amp * (1.0 + std::fabs(proj[i]) + std::fabs(feature[i])))) return false;
    return true;
}

// derivative check, once per cost-function TYPE: the ambient Jacobians the user's Evaluate returns, composed
// with the quaternion right-plus chart, must equal the closed form the HIP kernel evaluates
// (A hat(pInC) | -A R^T | A R^T, A = d proj / d pInC; SURVEY.md header fact 2).
inline bool ProbeReprojectionJacobian(const CostFunction* cost) {
    const ProbePoint& P = ProbePoints()[0];
    const double* p[3] = {P.q, P.t, P.L};
    double r[2], Jq[8], Jt[6], JL[6];
    double* J[3] = {Jq, Jt, JL};
    if (!cost->Evaluate(p, r, J)) return false;
    double plus[12];
    QuaternionRightPlus().ComputeJacobian(P.q, plus);
    double proj[2], pc[3];
    ReprojectionAt(P.q, P.t, P.L, proj, pc);
    const double zi = 1.0 / pc[2];
    const double A[6] = {zi, 0, -pc[0] * zi * zi, 0, zi, -pc[1] * zi * zi};
    const double H[9] = {0, -pc[2], pc[1], pc[2], 0, -pc[0], -pc[1], pc[0], 0};
    const double x = P.q[0], y = P.q[1], z = P.q[2], w = P.q[3];
    const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z),
                         2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) {
            double rot = 0, exp_rot = 0, exp_L = 0;
            for (int g = 0; g < 4; ++g) rot += Jq[i * 4 + g] * plus[g * 3 + j];
            for (int k = 0; k < 3; ++k) { exp_rot += A[i * 3 + k] * H[k * 3 + j]; exp_L += A[i * 3 + k] * R[j * 3 + k]; }   // (A R^T)_ij
            if (!(std::fabs(rot - exp_rot) <= 1e-9) || !(std::fabs(JL[i * 3 + j] - exp_L) <= 1e-9) ||
                !(std::fabs(Jt[i * 3 + j] + exp_L) <= 1e-9)) return false;
        }
    return true;
}

// joins a helper thread when the scope is left, however it is left
struct JoinOnExit {
    std::thread* t;
    ~JoinOnExit() { if (t && t->joinable()) t->join(); }
};

// [lo, hi) in `threads` contiguous ranges, one std::thread each (threads <= 1: inline).  Used for the per-block evaluations of the
// recognition only when Solver::Options::num_threads > 1 -- the caller's promise, as in Ceres, that Evaluate may run concurrently.
template <class F> inline void ParallelRanges(size_t n, int threads, F fn) {
    if (threads <= 1 || n < 4096) { fn((size_t)0, n); return; }
    const size_t nt = std::min<size_t>((size_t)threads, std::max<size_t>(1, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    for (size_t k = 0; k < nt; ++k) th.emplace_back(fn, n * k / nt, n * (k + 1) / nt);
    for (auto& t : th) t.join();
}

// ---- path 1: every residual block is the reprojection factor (built-in or recognised) --------
struct BaLayout {
    std::vector<int> rot_block, pos_block;   // per camera: Problem block indices
    std::vector<int> pt_block;               // per landmark
    std::vector<int> obs_cam, obs_pt;
    std::vector<double> feat;
    std::vector<unsigned char> user;         // per residual block: 1 = a user cost function taken over (0: the built-in factor)
    size_t n_user = 0;
};

// every recognised USER block evaluated at the parameter values its blocks hold right now (before the solve: DetectBa; after it: Solve)
inline bool VerifyRecognisedBlocks(Problem& p, const BaLayout& L, int threads) {
    if (!L.n_user) return true;
    auto& res = p.residuals();
    auto& blk = p.blocks();
    std::atomic<int> bad{0};
    ParallelRanges(res.size(), threads, [&](size_t lo, size_t hi) {
        for (size_t k = lo; k < hi; ++k) {
            if (!L.user[k]) continue;
            const auto& r = res[k];
            if (!ReprojectionValueAtData(r.cost, blk[r.blocks[0]].ptr, blk[r.blocks[1]].ptr, blk[r.blocks[2]].ptr, &L.feat[2 * k])) { bad.store(1); return; }
        }
    });
    return bad.load() == 0;
}

// the PER-BLOCK half of the recognition: every user block's feature (its residual at the canonical point) and its value at its own
// data.  10^6 blocks = 2 x 10^6 calls of the user's Evaluate, on the calling thread (options.num_threads of them): the one part of the
// recognition that costs time -- Solve runs it while a helper thread creates the device engine (SolveBa).
inline bool DetectBaBlocks(Problem& p, BaLayout* L, int threads) {
    auto& res = p.residuals();
    auto& blk = p.blocks();
    std::atomic<int> bad{0};
    ParallelRanges(res.size(), threads, [&](size_t lo, size_t hi) {
        for (size_t k = lo; k < hi; ++k) {
            if (!L->user[k]) continue;
            const auto& r = res[k];
            if (!CanonicalFeature(r.cost, &L->feat[2 * k]) ||
                !ReprojectionValueAtData(r.cost, blk[r.blocks[0]].ptr, blk[r.blocks[1]].ptr, blk[r.blocks[2]].ptr, &L->feat[2 * k])) { bad.store(1); return; }
        }
    });
    return bad.load() == 0;
}

// probe = false: only the SHAPE is required (blocks 4 / 3 / 3 -> 2 residuals, quaternion chart on the first block, no bounds);
// the factor itself stays the user's (host-linearised path, see Solve) and L->feat is left at zero.
// blocks_later = true: the structure and the per-TYPE probes only; the caller runs DetectBaBlocks itself.
inline bool DetectBa(Problem& p, BaLayout* L, bool probe = true, int threads = 1, bool blocks_later = false) {
    auto& res = p.residuals();
    auto& blk = p.blocks();
    if (res.empty()) return false;
    const size_t nr = res.size();
    // ---- structure: one pass, flat tables (block index -> camera / landmark)
    std::vector<int> cam_of_rot(blk.size(), -1), cam_of_pos(blk.size(), -1), pt_of(blk.size(), -1);
    L->obs_cam.resize(nr); L->obs_pt.resize(nr); L->feat.assign(2 * nr, 0.0); L->user.assign(nr, 0); L->n_user = 0;
    const std::type_info* last_type = nullptr;
    bool last_builtin = false;
    std::vector<std::pair<const std::type_info*, const CostFunction*>> first_of_type;
    for (size_t k = 0; k < nr; ++k) {
        const auto& r = res[k];
        if (r.blocks.size() != 3 || !IsReprojectionShape(r.cost)) return false;
        if (probe) {
            const std::type_info& ti = typeid(*r.cost);
            if (!last_type || !(ti == *last_type)) {
                last_type = &ti;
                last_builtin = dynamic_cast<const ReprojectionFactor*>(r.cost) != nullptr;
                bool seen = false;
                for (auto& ft : first_of_type) seen = seen || (*ft.first == ti);
                if (!seen) first_of_type.emplace_back(&ti, last_builtin ? nullptr : r.cost);
            }
            if (last_builtin) { auto* f = static_cast<const ReprojectionFactor*>(r.cost); L->feat[2 * k] = f->fx(); L->feat[2 * k + 1] = f->fy(); }
            else { L->user[k] = 1; ++L->n_user; }
        }
        const int b0 = r.blocks[0], b1 = r.blocks[1], b2 = r.blocks[2];
        const auto& rb = blk[b0];
        // a user LocalParameterization with the 4 -> 3 signature is accepted only if it IS the quaternion right-plus (checked
        // numerically by the caller through UsesQuaternionRightPlus)
        if (!rb.local || rb.local->GlobalSize() != 4 || rb.local->LocalSize() != 3) return false;
        if (blk[b1].local || blk[b2].local) return false;
        if (!rb.lower.empty() || !blk[b1].lower.empty() || !blk[b2].lower.empty()) return false;
        int c = cam_of_rot[b0];
        if (c < 0) {
            if (cam_of_pos[b1] >= 0) return false;                 // a position block shared by two rotation blocks
            c = (int)L->rot_block.size(); cam_of_rot[b0] = c; cam_of_pos[b1] = c; L->rot_block.push_back(b0); L->pos_block.push_back(b1);
        } else if (L->pos_block[(size_t)c] != b1) return false;    // a rotation block shared by two cameras with different position blocks
        int j = pt_of[b2];
        if (j < 0) { j = (int)L->pt_block.size(); pt_of[b2] = j; L->pt_block.push_back(b2); }
        L->obs_cam[k] = c; L->obs_pt[k] = j;
    }
    for (size_t c = 0; c < L->rot_block.size(); ++c)                // a block cannot be a rotation AND a landmark / position
        if (pt_of[(size_t)L->rot_block[c]] >= 0 || pt_of[(size_t)L->pos_block[c]] >= 0 || cam_of_pos[(size_t)L->rot_block[c]] >= 0) return false;
    if (!probe || !L->n_user) return true;
    // ---- the user's own cost functions (test_ceres.h:111-121): accepted iff they ARE the reprojection factor
    for (auto& ft : first_of_type) {
        double f[2];
        if (ft.second && (!ProbeReprojectionValue(ft.second, f) || !ProbeReprojectionJacobian(ft.second))) return false;
    }
    if (blocks_later) return true;
    return DetectBaBlocks(p, L, threads);
}

// numerically confirms that a user-supplied 4->3 parameterisation is q (x) exp(delta)
inline bool UsesQuaternionRightPlus(const LocalParameterization* lp) {
    if (dynamic_cast<const QuaternionRightPlus*>(lp)) return true;
    const double q[4] = {0.18257418583505536, 0.3651483716701107, 0.5477225575051661, 0.7302967433402214};
    const double d[3] = {0.013, -0.021, 0.008};
    double a[4], b[4];
    QuaternionRightPlus ref;
    if (!lp->Plus(q, d, a)) return false;
    ref.Plus(q, d, b);
    for (int i = 0; i < 4; ++i) if (std::fabs(a[i] - b[i]) > 1e-13) return false;
    return true;
}

struct BaSync { stba_ba* ba; Problem* p; const BaLayout* L; std::vector<double> cams, pts; };
inline void BaCopyOut(void* user) {
    auto* s = static_cast<BaSync*>(user);
    if (stba_ba_get_params(s->ba, s->cams.data(), s->pts.data()) != STBA_OK) return;
    for (size_t c = 0; c < s->L->rot_block.size(); ++c) {
        std::memcpy(s->p->blocks()[s->L->rot_block[c]].ptr, &s->cams[c * 7], 4 * sizeof(double));
        std::memcpy(s->p->blocks()[s->L->pos_block[c]].ptr, &s->cams[c * 7 + 4], 3 * sizeof(double));
    }
    for (size_t j = 0; j < s->L->pt_block.size(); ++j)
        std::memcpy(s->p->blocks()[s->L->pt_block[j]].ptr, &s->pts[j * 3], 3 * sizeof(double));
}

// host lineariser of the "gpu-ba-hostjac" path (stba_ba_set_host_linearizer): the user's cost functions, evaluated in bulk at
// the engine's current parameters, in residual-block order; the camera Jacobian is composed with the rotation block's chart
// (2x4 . 4x3) next to the 2x3 position block.  options.num_threads > 1 (and an OpenMP build) evaluates blocks in parallel,
// as Ceres does -- the reference pins num_threads = 1 (test_ceres.h:143).
struct BaHostCtx { Problem* p; const BaLayout* L; int threads; };
inline int BaHostLinearize(void* user, const double* cams, const double* pts, double* r, double* Jc, double* Jp) {
    auto* c = static_cast<BaHostCtx*>(user);
    auto& res = c->p->residuals();
    const int no = (int)res.size();
    int bad = 0;
#if defined(_OPENMP)
#pragma omp parallel for schedule(static) num_threads(c->threads > 1 ? c->threads : 1) reduction(+ : bad)
#endif
    for (int i = 0; i < no; ++i) {
        const double* q = cams + (size_t)c->L->obs_cam[i] * 7;
        const double* prm[3] = {q, q + 4, pts + (size_t)c->L->obs_pt[i] * 3};
        double Jq[8], Jt[6], JL[6];
        double* J[3] = {Jq, Jt, JL};
        if (!res[i].cost->Evaluate(prm, r + (size_t)i * 2, Jc ? J : nullptr)) { ++bad; continue; }
        if (!Jc) continue;
        double plus[12];
        const LocalParameterization* lp = c->p->blocks()[res[i].blocks[0]].local;
        if (!lp->ComputeJacobian(q, plus)) { ++bad; continue; }
        for (int row = 0; row < 2; ++row) {
            for (int l = 0; l < 3; ++l) {
                double sacc = 0.0;
                for (int g = 0; g < 4; ++g) sacc += Jq[row * 4 + g] * plus[g * 3 + l];
                Jc[(size_t)i * 12 + row * 6 + l] = sacc;
                Jc[(size_t)i * 12 + row * 6 + 3 + l] = Jt[row * 3 + l];
                Jp[(size_t)i * 6 + row * 3 + l] = JL[row * 3 + l];
            }
        }
    }
    return bad ? 1 : 0;
}

inline double WallSeconds() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// blocks_layout / blocks_ok (round 6): the per-block half of the recognition is still to be done (DetectBa(..., blocks_later)): it runs
// HERE, on the calling thread, while a helper thread creates the device engine (regrouping, Schur plan, uploads: ~50 ms at 10^6
// observations, next to ~35 ms of user Evaluate calls); the features it finds go to the engine afterwards (stba_ba_set_features).
// *blocks_ok = false: a block is not the reprojection factor -- nothing was solved, the engine is gone, the caller goes on.
inline bool SolveBa(const Solver::Options& o, Problem* p, const BaLayout& L, Solver::Summary* sum, bool host_jacobians = false,
                    BaLayout* blocks_layout = nullptr, bool* blocks_ok = nullptr, std::thread* destroyer = nullptr) {
    double t0 = WallSeconds();
    auto lap = [&](double* acc) { const double t1 = WallSeconds(); *acc += t1 - t0; t0 = t1; };
    const int nc = (int)L.rot_block.size(), np = (int)L.pt_block.size(), no = (int)L.obs_cam.size();
    BaSync sync{nullptr, p, &L, std::vector<double>((size_t)nc * 7), std::vector<double>((size_t)np * 3)};
    std::vector<unsigned char> cam_fixed((size_t)nc * 6, 0), pt_fixed((size_t)np, 0);
    for (int c = 0; c < nc; ++c) {
        const auto& rb = p->blocks()[L.rot_block[c]]; const auto& pb = p->blocks()[L.pos_block[c]];
        std::memcpy(&sync.cams[(size_t)c * 7], rb.ptr, 4 * sizeof(double));
        std::memcpy(&sync.cams[(size_t)c * 7 + 4], pb.ptr, 3 * sizeof(double));
        if (rb.constant) cam_fixed[c * 6] = cam_fixed[c * 6 + 1] = cam_fixed[c * 6 + 2] = 1;
        if (pb.constant) cam_fixed[c * 6 + 3] = cam_fixed[c * 6 + 4] = cam_fixed[c * 6 + 5] = 1;
    }
    for (int j = 0; j < np; ++j) {
        std::memcpy(&sync.pts[(size_t)j * 3], p->blocks()[L.pt_block[j]].ptr, 3 * sizeof(double));
        pt_fixed[j] = p->blocks()[L.pt_block[j]].constant ? 1 : 0;
    }
    lap(&sum->phases.pack);
    int rc = STBA_OK;
    std::string create_error;
    auto create = [&]() {
        rc = stba_ba_create(&sync.ba, nc, np, no, sync.cams.data(), sync.pts.data(), L.obs_cam.data(), L.obs_pt.data(),
                            L.feat.data(), cam_fixed.data(), pt_fixed.data(), nullptr);
        if (rc != STBA_OK) create_error = stba_last_error();      // (the error text is thread-local: taken where it was set)
    };
    int device = 0;
    if (blocks_layout && !std::getenv("STBA_CERES_NO_OVERLAP") && stba_get_device(&device) == STBA_OK) {
        // (the helper thread creates the engine from the layout as it stands -- the user blocks' features still zero: nothing reads
        // them before the first linearisation; it starts on device 0 whatever this thread has selected, hence stba_set_device)
        std::thread helper([&]() { if ((rc = stba_set_device(device)) != STBA_OK) create_error = stba_last_error(); else create(); });
        JoinOnExit helper_guard{&helper};                           // (a user Evaluate that throws must not leave the thread running)
        const double tb0 = WallSeconds();
        const bool ok = DetectBaBlocks(*p, blocks_layout, o.num_threads);
        const double t_blocks = WallSeconds() - tb0;
        helper.join();
        sum->phases.recognise += t_blocks;
        t0 += t_blocks;                                             // (what is left of the creation's wall time goes to engine_create)
        if (!ok) {
            if (rc == STBA_OK) stba_ba_destroy(sync.ba);
            lap(&sum->phases.engine_create);
            if (blocks_ok) *blocks_ok = false;
            return false;
        }
        if (rc == STBA_OK && (rc = stba_ba_set_features(sync.ba, L.feat.data())) != STBA_OK) { create_error = stba_last_error(); stba_ba_destroy(sync.ba); }
    } else {
        if (blocks_layout && !DetectBaBlocks(*p, blocks_layout, o.num_threads)) { if (blocks_ok) *blocks_ok = false; lap(&sum->phases.recognise); return false; }
        if (blocks_layout) lap(&sum->phases.recognise);
        create();
    }
    lap(&sum->phases.engine_create);
    if (rc != STBA_OK) { sum->termination_type = FAILURE; sum->message = std::string("stba_ba_create: ") + create_error; return false; }
    BaHostCtx hctx{p, &L, o.num_threads};
    if (host_jacobians && (rc = stba_ba_set_host_linearizer(sync.ba, &BaHostLinearize, &hctx)) != STBA_OK) {
        sum->termination_type = FAILURE; sum->message = std::string("stba_ba_set_host_linearizer: ") + stba_last_error();
        stba_ba_destroy(sync.ba);
        return false;
    }
    stba_lm_options co = ToC(o);
    stba_lm_summary cs;
    std::vector<double> trace((size_t)(o.max_num_iterations + 1) * STBA_TRACE_COLS, 0.0);
    CallbackCtx ctx{&o, sum, &BaCopyOut, &sync};
    rc = stba_ba_solve(sync.ba, &co, &cs, trace.data(), o.callbacks.empty() ? nullptr : &IterationTrampoline, &ctx);
    lap(&sum->phases.device_solve);
    if (rc == STBA_OK) {
        BaCopyOut(&sync);   // parameters are updated in place, like ceres::Solve
        FillSummary(cs, trace, sum);
        if (ctx.user_abort) sum->termination_type = USER_FAILURE;
        if (ctx.user_success) sum->termination_type = USER_SUCCESS;
    } else {
        sum->termination_type = FAILURE;
        sum->message = std::string("stba_ba_solve: ") + stba_last_error();
    }
    // (the engine's ~40 device buffers take a few ms to free: with `destroyer` that happens on a helper thread, next to the caller's
    // end-point check of the recognised blocks; the caller joins it)
    int dev_now = 0;
    if (destroyer && stba_get_device(&dev_now) == STBA_OK) {
        stba_ba* engine = sync.ba;
        *destroyer = std::thread([engine, dev_now]() { if (stba_set_device(dev_now) == STBA_OK) stba_ba_destroy(engine); });
    } else stba_ba_destroy(sync.ba);
    lap(&sum->phases.write_back);
    return rc == STBA_OK;
}

// ---- path 2: generic residual blocks through the host-callback dense path -------------------
struct DenseCtx {
    Problem* p;
    std::vector<int> var_blocks;         // non-constant block indices
    std::vector<int> amb_off, loc_off;   // per var block
    int n_amb = 0, n_loc = 0, n_res = 0;
    // scratch of DenseResidual, sized once per solve (it runs a dozen times per solve of a 40-residual problem whose published
    // wall time is 0.12 ms: no allocation per call, none per residual block)
    std::vector<const double*> cur, params;
    std::vector<int> var_index;
    std::vector<double*> jacs;
    std::vector<double> jac_store, plusJ;
    void Prepare() {
        auto& blocks = p->blocks();
        cur.resize(blocks.size()); var_index.assign(blocks.size(), -1);
        for (size_t v = 0; v < var_blocks.size(); ++v) var_index[var_blocks[v]] = (int)v;
        size_t max_nb = 0, max_jac = 0, max_plus = 0;
        for (auto& res : p->residuals()) {
            size_t need = 0;
            for (int k : res.blocks) { need += (size_t)res.cost->num_residuals() * blocks[k].size; max_plus = std::max(max_plus, (size_t)blocks[k].size * blocks[k].local_size()); }
            max_nb = std::max(max_nb, res.blocks.size()); max_jac = std::max(max_jac, need);
        }
        params.resize(max_nb); jacs.resize(max_nb); jac_store.resize(max_jac); plusJ.resize(max_plus);
    }
};

inline int DenseResidual(void* user, const double* x, double* r, double* J) {
    auto* c = static_cast<DenseCtx*>(user);
    auto& blocks = c->p->blocks();
    for (size_t k = 0; k < blocks.size(); ++k) c->cur[k] = blocks[k].ptr;
    for (size_t v = 0; v < c->var_blocks.size(); ++v) c->cur[c->var_blocks[v]] = x + c->amb_off[v];
    if (J) std::fill(J, J + (size_t)c->n_res * c->n_loc, 0.0);
    int row = 0;
    for (auto& res : c->p->residuals()) {
        const int nr = res.cost->num_residuals();
        const size_t nb = res.blocks.size();
        const double** params = c->params.data();
        double** jacs = c->jacs.data();
        size_t o = 0;
        for (size_t b = 0; b < nb; ++b) {
            params[b] = c->cur[res.blocks[b]];
            // constant blocks get no Jacobian request, like Ceres (NB solver.hpp:183: the reference's
            // PnPSizedCostFunction then skips ALL its Jacobians -- only hit when a block is constant)
            jacs[b] = (J && c->var_index[res.blocks[b]] >= 0) ? c->jac_store.data() + o : nullptr;
            o += (size_t)nr * blocks[res.blocks[b]].size;
        }
        if (J) std::fill(c->jac_store.begin(), c->jac_store.begin() + (long)o, 0.0);
        if (!res.cost->Evaluate(params, r + row, J ? jacs : nullptr)) return 1;
        if (J) {
            for (size_t b = 0; b < nb; ++b) {
                const int v = c->var_index[res.blocks[b]];
                if (v < 0 || !jacs[b]) continue;
                const auto& blk = blocks[res.blocks[b]];
                const int gs = blk.size, ls = blk.local_size();
                if (blk.local) {
                    double* plusJ = c->plusJ.data();
                    std::fill(plusJ, plusJ + (size_t)gs * ls, 0.0);
                    if (!blk.local->ComputeJacobian(params[b], plusJ)) return 1;
                    for (int rr = 0; rr < nr; ++rr)
                        for (int l = 0; l < ls; ++l) {
                            double s = 0;
                            for (int g = 0; g < gs; ++g) s += jacs[b][rr * gs + g] * plusJ[(size_t)g * ls + l];
                            J[(size_t)(row + rr) * c->n_loc + c->loc_off[v] + l] += s;
                        }
                } else {
                    for (int rr = 0; rr < nr; ++rr)
                        for (int g = 0; g < gs; ++g) J[(size_t)(row + rr) * c->n_loc + c->loc_off[v] + g] += jacs[b][rr * gs + g];
                }
            }
        }
        row += nr;
    }
    return 0;
}

inline void DensePlus(void* user, const double* x, const double* d, double* out) {
    auto* c = static_cast<DenseCtx*>(user);
    for (size_t v = 0; v < c->var_blocks.size(); ++v) {
        const auto& blk = c->p->blocks()[c->var_blocks[v]];
        if (blk.local) blk.local->Plus(x + c->amb_off[v], d + c->loc_off[v], out + c->amb_off[v]);
        else for (int k = 0; k < blk.size; ++k) out[c->amb_off[v] + k] = x[c->amb_off[v] + k] + d[c->loc_off[v] + k];
    }
}

struct DenseSync { DenseCtx* c; const double* x; };

inline bool SolveDense(const Solver::Options& o, Problem* p, Solver::Summary* sum) {
    DenseCtx c;
    c.p = p;
    bool any_bounds = false, any_local = false;
    std::vector<unsigned char> used(p->blocks().size(), 0);
    for (auto& r : p->residuals()) for (int k : r.blocks) used[(size_t)k] = 1;
    for (auto& b : p->blocks()) {
        if (!used[(size_t)b.index] || b.constant) continue;
        c.var_blocks.push_back(b.index); c.amb_off.push_back(c.n_amb); c.loc_off.push_back(c.n_loc);
        c.n_amb += b.size; c.n_loc += b.local_size();
        any_bounds |= !b.lower.empty(); any_local |= (b.local != nullptr);
    }
    c.n_res = p->NumResiduals();
    c.Prepare();
    if (c.n_loc == 0 || c.n_res == 0) { sum->termination_type = CONVERGENCE; sum->message = "nothing to optimise"; return true; }
    if (c.n_loc > 4096 || (double)c.n_loc * c.n_res > 2.7e8) { sum->termination_type = FAILURE; sum->message = "generic (callback) problems are limited to 4096 local parameters and 2.7e8 Jacobian entries; use ReprojectionFactor for large bundle adjustment"; return false; }
    std::vector<double> x(c.n_amb), lo, up;
    for (size_t v = 0; v < c.var_blocks.size(); ++v) std::memcpy(&x[c.amb_off[v]], p->blocks()[c.var_blocks[v]].ptr, sizeof(double) * p->blocks()[c.var_blocks[v]].size);
    if (any_bounds) {
        lo.assign(c.n_amb, -1e300); up.assign(c.n_amb, 1e300);
        for (size_t v = 0; v < c.var_blocks.size(); ++v) {
            const auto& b = p->blocks()[c.var_blocks[v]];
            if (!b.lower.empty()) for (int k = 0; k < b.size; ++k) { lo[c.amb_off[v] + k] = b.lower[k]; up[c.amb_off[v] + k] = b.upper[k]; }
        }
    }
    stba_lm_options co = ToC(o);
    stba_lm_summary cs;
    std::vector<double> trace((size_t)(o.max_num_iterations + 1) * STBA_TRACE_COLS, 0.0);
    struct Sync { DenseCtx* c; double* x; } sync{&c, x.data()};
    auto copy_out = [](void* u) {
        auto* s = static_cast<Sync*>(u);
        for (size_t v = 0; v < s->c->var_blocks.size(); ++v) {
            auto& b = s->c->p->blocks()[s->c->var_blocks[v]];
            std::memcpy(b.ptr, s->x + s->c->amb_off[v], sizeof(double) * b.size);
        }
    };
    CallbackCtx ctx{&o, sum, copy_out, &sync};
    const int rc = stba_dense_solve(&DenseResidual, any_local ? &DensePlus : nullptr, &c, c.n_amb, c.n_loc, c.n_res, x.data(),
                                    any_bounds ? lo.data() : nullptr, any_bounds ? up.data() : nullptr, &co, &cs, trace.data(),
                                    o.callbacks.empty() ? nullptr : &IterationTrampoline, &ctx);
    if (rc != STBA_OK) { sum->termination_type = FAILURE; sum->message = std::string("stba_dense_solve: ") + stba_last_error(); return false; }
    copy_out(&sync);
    FillSummary(cs, trace, sum);
    if (ctx.user_abort) sum->termination_type = USER_FAILURE;
    if (ctx.user_success) sum->termination_type = USER_SUCCESS;
    return true;
}

// ---- path 3: pose graph on the device engine (stba_pg_*) -------------------------------------
// returns false WITHOUT touching the summary if the problem is not a pose graph of built-in factors
inline bool SolvePoseGraph(const Solver::Options& o, Problem* p, Solver::Summary* sum) {
    if (p->residuals().empty()) return false;
    std::map<int, int> node_of;
    std::vector<int> node_block, ei, ej;
    std::vector<double> meas;
    for (auto& r : p->residuals()) {
        auto* f = dynamic_cast<RelativePoseFactor*>(r.cost);
        if (!f || r.blocks.size() != 2 || r.blocks[0] == r.blocks[1]) return false;
        int ends[2];
        for (int k = 0; k < 2; ++k) {
            const auto& b = p->blocks()[r.blocks[k]];
            if (b.size != 7 || !b.local || !dynamic_cast<SE3RightPlus*>(b.local) || !b.lower.empty()) return false;
            auto it = node_of.find(r.blocks[k]);
            if (it == node_of.end()) { ends[k] = (int)node_block.size(); node_of[r.blocks[k]] = ends[k]; node_block.push_back(r.blocks[k]); }
            else ends[k] = it->second;
        }
        ei.push_back(ends[0]); ej.push_back(ends[1]);
        meas.insert(meas.end(), f->measurement(), f->measurement() + 7);
    }
    const int n = (int)node_block.size(), m = (int)ei.size();
    std::vector<double> poses((size_t)n * 7);
    std::vector<unsigned char> fixed((size_t)n, 0);
    for (int k = 0; k < n; ++k) {
        std::memcpy(&poses[(size_t)k * 7], p->blocks()[node_block[k]].ptr, 7 * sizeof(double));
        fixed[k] = p->blocks()[node_block[k]].constant ? 1 : 0;
    }
    sum->execution_path = "gpu-pg";
    stba_pg* pg = nullptr;
    int rc = stba_pg_create(&pg, n, m, poses.data(), ei.data(), ej.data(), meas.data(), fixed.data(), nullptr);
    if (rc != STBA_OK) { sum->termination_type = FAILURE; sum->message = std::string("stba_pg_create: ") + stba_last_error(); return true; }
    stba_lm_options co = ToC(o);
    stba_lm_summary cs;
    std::vector<double> trace((size_t)(o.max_num_iterations + 1) * STBA_TRACE_COLS, 0.0);
    int pcg_total = 0;
    rc = stba_pg_solve(pg, &co, nullptr, &cs, trace.data(), &pcg_total);
    if (rc == STBA_OK) rc = stba_pg_get_poses(pg, poses.data());
    if (rc == STBA_OK) {
        for (int k = 0; k < n; ++k) std::memcpy(p->blocks()[node_block[k]].ptr, &poses[(size_t)k * 7], 7 * sizeof(double));
        FillSummary(cs, trace, sum);
    } else {
        sum->termination_type = FAILURE;
        sum->message = std::string("stba_pg_solve: ") + stba_last_error();
    }
    stba_pg_destroy(pg);
    return true;
}

}  // namespace internal

namespace internal {
inline void SolveDispatch(const Solver::Options& options, Problem* problem, Solver::Summary* summary) {
    if (problem->NumLossFunctions() > 0) {
        // (Ceres would apply the loss; this layer has none to apply -- an unweighted solve would be a silently different problem)
        summary->termination_type = FAILURE;
        summary->message = "stba_ceres: " + std::to_string(problem->NumLossFunctions()) + " residual block(s) carry a LossFunction; robust losses are "
                           "not implemented by this layer (the reference passes nullptr: test_ceres.h:120) -- nothing was solved.";
        std::fprintf(stderr, "%s\n", summary->message.c_str());
        return;
    }
    BaLayout L;
    // CONTRACT of the recognition (DetectBa): a user cost function is replaced by the built-in device factor if
    //   * the first block of its C++ type equals proj(conj(q)(L - t)) - feature at four generic probe points, one of them behind the
    //     camera (1e-12), and its Jacobian, composed with the chart, equals the closed form at a probe point (1e-9);
    //   * EVERY block equals the factor AT ITS OWN DATA -- the parameter values it holds when Solve is called (1e-11) -- with the
    //     feature it shows at the canonical point, and once more at the point the solve ENDED at: a cost that departed from the
    //     factor on the way is solved again with its own code (below).
    // What remains unchecked is a cost that differs from the factor only at iterates in between; Solver::Options::force_callback_path
    // = true (or STBA_CERES_FORCE_CALLBACK=1 in the environment) keeps every block on the generic path, where the user's Evaluate is
    // what runs.  The summary's message names the number of blocks taken over.
    // NOTE for callers with IterationCallbacks: if the end-point check fails, the callbacks have already fired for the discarded
    // device solve and fire again for the second one (the summary's message says that a second solve ran; its iterations are the
    // ones reported, the time of both is in total_time_in_seconds, the first one's under phases.resolve's complement).
    const char* fe = std::getenv("STBA_CERES_FORCE_CALLBACK");
    const bool force_cb = options.force_callback_path || (fe && *fe && *fe != '0');
    double t0 = WallSeconds();
    bool ba = !force_cb && DetectBa(*problem, &L, true, options.num_threads, true);      // (the per-block half: inside SolveBa)
    if (ba)
        for (int rb : L.rot_block) ba = ba && UsesQuaternionRightPlus(problem->blocks()[rb].local);
    summary->phases.recognise = WallSeconds() - t0;
    std::string carried;
    std::vector<double> saved;
    bool still_factor = true;
    if (ba) {
        if (L.n_user) for (auto& b : problem->blocks()) saved.insert(saved.end(), b.ptr, b.ptr + b.size);
        summary->execution_path = "gpu-ba";
        bool blocks_ok = true;
        std::thread destroyer;
        JoinOnExit destroyer_guard{&destroyer};
        SolveBa(options, problem, L, summary, false, L.n_user ? &L : nullptr, &blocks_ok, L.n_user ? &destroyer : nullptr);
        if (!blocks_ok) { ba = false; summary->execution_path.clear(); }       // a block is not the factor: on to the other paths, nothing was touched
        t0 = WallSeconds();
        still_factor = !ba || summary->termination_type == FAILURE || VerifyRecognisedBlocks(*problem, L, options.num_threads);
        if (destroyer.joinable()) destroyer.join();
        if (ba) summary->phases.verify = WallSeconds() - t0;
    }
    if (ba) {
        const bool still = still_factor;
        if (still) {
            if (L.n_user) summary->message += (summary->message.empty() ? "" : " ") + std::to_string(L.n_user) +
                                              " user cost functions recognised as the reprojection factor (probe points, start point, end point) and evaluated by the device kernel.";
            return;
        }
        size_t o = 0;
        for (auto& b : problem->blocks()) { std::copy(saved.begin() + o, saved.begin() + o + b.size, b.ptr); o += (size_t)b.size; }
        const Solver::Summary::Phases first = summary->phases;
        *summary = Solver::Summary();
        summary->phases = first;
        carried = "a user cost function taken for the reprojection factor differs from it at the solution: solved again with the user's Evaluate.";
    }
    const double t_resolve = WallSeconds();
    // a pose graph: every residual block a RelativePoseFactor between two 7-double pose blocks with the SE3 right-plus chart
    if (!force_cb && options.callbacks.empty() && SolvePoseGraph(options, problem, summary)) return;
    // BA-SHAPED, but not (or not to be taken for) the built-in factor: every residual block is {quaternion 4, position 3,
    // landmark 3} -> 2 with the quaternion right-plus chart.  The user's cost functions are evaluated on the host, in bulk, into
    // the device engine's residual / Jacobian buffers, and the Schur complement, the factorisation, the back-substitution and the
    // LM loop run on the device as for the built-in factor ("gpu-ba-hostjac"): any size the engine takes, where the dense
    // callback path below stops at 4096 local parameters.  (The reference's BA cost IS a generic functor: test_ceres.h:56.)
    BaLayout L2;
    bool shape = DetectBa(*problem, &L2, false);
    if (shape)
        for (int rb : L2.rot_block) shape = shape && UsesQuaternionRightPlus(problem->blocks()[rb].local);
    if (shape) {
        summary->execution_path = "gpu-ba-hostjac";
        Solver::Summary::Phases ph = summary->phases;
        SolveBa(options, problem, L2, summary, true);
        if (!carried.empty()) { summary->phases = ph; summary->phases.resolve = WallSeconds() - t_resolve; }
    }
    else { summary->execution_path = "gpu-dense-callback"; SolveDense(options, problem, summary); }
    if (!carried.empty()) summary->message = carried + (summary->message.empty() ? "" : " ") + summary->message;
}
}  // namespace internal

inline void Solve(const Solver::Options& options, Problem* problem, Solver::Summary* summary) {
    *summary = Solver::Summary();
    const double t0 = internal::WallSeconds();
    internal::SolveDispatch(options, problem, summary);
    // Ceres' timing fields: total = everything inside Solve(); minimizer = the engine's own solve time (set by FillSummary)
    summary->total_time_in_seconds = internal::WallSeconds() - t0;
    const auto& ph = summary->phases;
    summary->preprocessor_time_in_seconds = ph.recognise + ph.pack + ph.engine_create;
    summary->postprocessor_time_in_seconds = ph.write_back + ph.verify;
}

}  // namespace stba_ceres
#endif

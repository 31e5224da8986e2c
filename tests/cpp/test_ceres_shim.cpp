// Restates the reference's L2 "problem construction" call sites on top of include/stba/ceres.h:
//   st17-ceres/src/ceres_bound.cpp:25-68                    bounds demo
//   st17-ceres/src/include/solver.hpp:247-385               SolvePnPWith{DynamicAutoDiff,AutoDiff,SizedCostFunction}
//   st20-g2o/src/include/test_ceres.h:98-152                SolveWithCeresDynamicAutoDiff
// with the user-side classes (LieLocalParameterization, functors, callbacks) written against the
// mirrored API exactly as the reference writes them against Ceres.  Sophus/Eigen are replaced by
// a few templated helpers.  Prints "key value..." lines that tests/test_cpp_shim.py checks.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#include "stba/ceres.h"
namespace ceres = stba_ceres;

// ---------------------------------------------------------------- tiny SO3 helpers (Sophus stand-ins)
template <typename T> static void QuatConjRotate(const T* q, const T* v, T* out) {   // conj(q) * v
    const T u0 = -q[0], u1 = -q[1], u2 = -q[2], w = q[3];
    const T a0 = T(2.0) * (u1 * v[2] - u2 * v[1]), a1 = T(2.0) * (u2 * v[0] - u0 * v[2]), a2 = T(2.0) * (u0 * v[1] - u1 * v[0]);
    out[0] = v[0] + w * a0 + (u1 * a2 - u2 * a1);
    out[1] = v[1] + w * a1 + (u2 * a0 - u0 * a2);
    out[2] = v[2] + w * a2 + (u0 * a1 - u1 * a0);
}
static void So3Exp(const double* w, double* q) {
    const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double im = th < 1e-10 ? 0.5 : std::sin(0.5 * th) / th;
    q[0] = im * w[0]; q[1] = im * w[1]; q[2] = im * w[2]; q[3] = std::cos(0.5 * th);
}
static void So3Log(const double* q, double* w) {
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    const double k = n < 1e-10 ? 2.0 / q[3] : 2.0 * std::atan2(n, q[3]) / n;
    w[0] = k * q[0]; w[1] = k * q[1]; w[2] = k * q[2];
}
static void QuatMul(const double* a, const double* b, double* o) {
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}

// ---------------------------------------------------------------- user classes, as in the reference
// solver.hpp:30-61 / test_ceres.h:14-45
class LieLocalParameterization : public ceres::LocalParameterization {
public:
    bool Plus(double const* T_raw, double const* delta_raw, double* T_plus_delta_raw) const override {
        double e[4];
        So3Exp(delta_raw, e);
        QuatMul(T_raw, e, T_plus_delta_raw);
        return true;
    }
    bool ComputeJacobian(double const* q, double* J) const override {   // Dx_this_mul_exp_x_at_0
        const double c0 = q[3] / 2, c1 = q[2] / 2, c2 = -c1, c3 = q[1] / 2, c4 = q[0] / 2, c5 = -c4, c6 = -c3;
        const double M[12] = {c0, c2, c3, c1, c0, c5, c6, c4, c0, c5, c6, c2};
        std::memcpy(J, M, sizeof M);
        return true;
    }
    int GlobalSize() const override { return 4; }
    int LocalSize() const override { return 3; }
};

// solver.hpp:63-94
class LieR3LocalParameterization : public ceres::LocalParameterization {
public:
    bool Plus(const double* x, const double* delta, double* x_plus_delta) const override {
        double a[4], b[4], c[4];
        So3Exp(x, a); So3Exp(delta, b); QuatMul(a, b, c);
        So3Log(c, x_plus_delta);
        return true;
    }
    bool ComputeJacobian(const double*, double* j) const override {
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        std::memcpy(j, I, sizeof I);
        return true;
    }
    int GlobalSize() const override { return 3; }
    int LocalSize() const override { return 3; }
};

struct CorrPair { double point[3]; double feature[2]; };   // solver.hpp:18-26

// solver.hpp:96-125
struct PnPDynamicAutoDiffFunctor {
    CorrPair c;
    explicit PnPDynamicAutoDiffFunctor(const CorrPair& cp) : c(cp) {}
    static auto Create(const CorrPair& cp) { return new ceres::DynamicAutoDiffCostFunction<PnPDynamicAutoDiffFunctor>(new PnPDynamicAutoDiffFunctor(cp)); }
    template <typename T> bool operator()(T const* const* parameters, T* residuals) const {
        const T* q = parameters[0]; const T* t = parameters[1];
        T d[3] = {T(c.point[0]) - t[0], T(c.point[1]) - t[1], T(c.point[2]) - t[2]}, pc[3];
        QuatConjRotate(q, d, pc);
        residuals[0] = pc[0] / pc[2] - T(c.feature[0]);
        residuals[1] = pc[1] / pc[2] - T(c.feature[1]);
        return true;
    }
};

// solver.hpp:127-155
struct PnPAutoDiffFunctor {
    CorrPair c;
    explicit PnPAutoDiffFunctor(const CorrPair& cp) : c(cp) {}
    static auto Create(const CorrPair& cp) { return new ceres::AutoDiffCostFunction<PnPAutoDiffFunctor, 2, 4, 3>(new PnPAutoDiffFunctor(cp)); }
    template <typename T> bool operator()(const T* const q, const T* const t, T* residuals) const {
        T d[3] = {T(c.point[0]) - t[0], T(c.point[1]) - t[1], T(c.point[2]) - t[2]}, pc[3];
        QuatConjRotate(q, d, pc);
        residuals[0] = pc[0] / pc[2] - T(c.feature[0]);
        residuals[1] = pc[1] / pc[2] - T(c.feature[1]);
        return true;
    }
};

// solver.hpp:157-213; reference_rot_formula = true reproduces solver.hpp:195 (hat(R^-1 Pw)),
// false is the correct right-perturbation derivative hat(pInC) (SURVEY.md header fact 2)
struct PnPSizedCostFunction : public ceres::SizedCostFunction<2, 3, 3> {
    CorrPair c; bool reference_rot_formula;
    PnPSizedCostFunction(const CorrPair& cp, bool ref) : c(cp), reference_rot_formula(ref) {}
    bool Evaluate(const double* const* parameters, double* residuals, double** jacobians) const override {
        double q[4]; So3Exp(parameters[0], q);
        const double* t = parameters[1];
        double d[3] = {c.point[0] - t[0], c.point[1] - t[1], c.point[2] - t[2]}, pc[3];
        QuatConjRotate(q, d, pc);
        residuals[0] = pc[0] / pc[2] - c.feature[0];
        residuals[1] = pc[1] / pc[2] - c.feature[1];
        if (jacobians != nullptr && jacobians[0] != nullptr && jacobians[1] != nullptr) {   // solver.hpp:183
            const double X = pc[0], Y = pc[1], Z = pc[2], Zi = 1.0 / Z;
            const double A[6] = {Zi, 0, -X * Zi * Zi, 0, Zi, -Y * Zi * Zi};
            double h[3];
            if (reference_rot_formula) QuatConjRotate(q, c.point, h); else { h[0] = X; h[1] = Y; h[2] = Z; }
            const double H[9] = {0, -h[2], h[1], h[2], 0, -h[0], -h[1], h[0], 0};
            double e1[3] = {1, 0, 0}, e2[3] = {0, 1, 0}, e3[3] = {0, 0, 1}, r1[3], r2[3], r3[3];
            QuatConjRotate(q, e1, r1); QuatConjRotate(q, e2, r2); QuatConjRotate(q, e3, r3);   // columns of R^T
            const double Rt[9] = {r1[0], r2[0], r3[0], r1[1], r2[1], r3[1], r1[2], r2[2], r3[2]};
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 3; ++j) {
                    double s = 0, sp = 0;
                    for (int k = 0; k < 3; ++k) { s += A[i * 3 + k] * H[k * 3 + j]; sp += A[i * 3 + k] * Rt[k * 3 + j]; }
                    jacobians[0][i * 3 + j] = s;
                    jacobians[1][i * 3 + j] = -sp;
                }
        }
        return true;
    }
};

// solver.hpp:215-245: reads the LIVE parameter memory from inside the callback
struct VisualCallBack : public ceres::IterationCallback {
    const double* so3; const double* pos; std::vector<double> seen;
    VisualCallBack(const double* s, const double* p) : so3(s), pos(p) {}
    ceres::CallbackReturnType operator()(const ceres::IterationSummary&) override {
        seen.push_back(pos[0]); seen.push_back(so3[3]);
        return ceres::SOLVER_CONTINUE;
    }
};

// test_ceres.h:47-81: the user's own functor.  Solve() recognises it numerically as the reprojection factor
// and runs the problem on the device-resident BA engine (execution_path "gpu-ba") -- no source edit.
struct ProjectFactor {
    double feature[2];
    explicit ProjectFactor(const double* f) { feature[0] = f[0]; feature[1] = f[1]; }
    static auto Create(const double* f) { return new ceres::DynamicAutoDiffCostFunction<ProjectFactor>(new ProjectFactor(f)); }
    template <typename T> bool operator()(T const* const* parameters, T* residuals) const {
        const T* q = parameters[0]; const T* t = parameters[1]; const T* L = parameters[2];
        T d[3] = {L[0] - t[0], L[1] - t[1], L[2] - t[2]}, pc[3];
        QuatConjRotate(q, d, pc);
        residuals[0] = pc[0] / pc[2] - T(feature[0]);
        residuals[1] = pc[1] / pc[2] - T(feature[1]);
        return true;
    }
};

// NOT the reprojection factor (residual scaled by 2: same minimiser, different function): must be rejected
// by the probe; BA-shaped, so the device engine runs it with this functor evaluated on the host ("gpu-ba-hostjac").
struct ScaledProjectFactor {
    double feature[2];
    explicit ScaledProjectFactor(const double* f) { feature[0] = f[0]; feature[1] = f[1]; }
    static auto Create(const double* f) { return new ceres::DynamicAutoDiffCostFunction<ScaledProjectFactor>(new ScaledProjectFactor(f)); }
    template <typename T> bool operator()(T const* const* parameters, T* residuals) const {
        const T* q = parameters[0]; const T* t = parameters[1]; const T* L = parameters[2];
        T d[3] = {L[0] - t[0], L[1] - t[1], L[2] - t[2]}, pc[3];
        QuatConjRotate(q, d, pc);
        residuals[0] = T(2.0) * (pc[0] / pc[2] - T(feature[0]));
        residuals[1] = T(2.0) * (pc[1] / pc[2] - T(feature[1]));
        return true;
    }
};

// The reprojection factor EXCEPT where the landmark lies further than `far` from the origin: there the residual is doubled.  With `far`
// beyond every generic probe point this functor passes the probes; it is caught at its own DATA (DetectBa evaluates every recognised
// block at the parameter values it holds) and keeps its own code ("gpu-ba-hostjac").
struct FarScaledProjectFactor {
    double feature[2], far2;
    FarScaledProjectFactor(const double* f, double far) { feature[0] = f[0]; feature[1] = f[1]; far2 = far * far; }
    static auto Create(const double* f, double far) { return new ceres::DynamicAutoDiffCostFunction<FarScaledProjectFactor>(new FarScaledProjectFactor(f, far)); }
    template <typename T> bool operator()(T const* const* parameters, T* residuals) const {
        const T* q = parameters[0]; const T* t = parameters[1]; const T* L = parameters[2];
        T d[3] = {L[0] - t[0], L[1] - t[1], L[2] - t[2]}, pc[3];
        QuatConjRotate(q, d, pc);
        const T n2 = L[0] * L[0] + L[1] * L[1] + L[2] * L[2];
        const T w = (n2 > T(far2)) ? T(2.0) : T(1.0);
        residuals[0] = w * (pc[0] / pc[2] - T(feature[0]));
        residuals[1] = w * (pc[1] / pc[2] - T(feature[1]));
        return true;
    }
};

// The reprojection factor at the probe points AND at the start point; it departs from it once its landmark has MOVED (between 1e-7 and
// 0.5 away from where it started): only the check at the END of the solve can see that.  Solve() must notice, put the parameters
// back and solve again with this functor's own code.
struct MovedScaledProjectFactor {
    double feature[2], L0[3];
    MovedScaledProjectFactor(const double* f, const double* l0) { feature[0] = f[0]; feature[1] = f[1]; L0[0] = l0[0]; L0[1] = l0[1]; L0[2] = l0[2]; }
    static auto Create(const double* f, const double* l0) { return new ceres::DynamicAutoDiffCostFunction<MovedScaledProjectFactor>(new MovedScaledProjectFactor(f, l0)); }
    template <typename T> bool operator()(T const* const* parameters, T* residuals) const {
        const T* q = parameters[0]; const T* t = parameters[1]; const T* L = parameters[2];
        T d[3] = {L[0] - t[0], L[1] - t[1], L[2] - t[2]}, pc[3];
        QuatConjRotate(q, d, pc);
        const T m[3] = {L[0] - T(L0[0]), L[1] - T(L0[1]), L[2] - T(L0[2])};
        const T m2 = m[0] * m[0] + m[1] * m[1] + m[2] * m[2];
        const T w = (m2 > T(1e-14) && m2 < T(0.25)) ? T(1.5) : T(1.0);
        residuals[0] = w * (pc[0] / pc[2] - T(feature[0]));
        residuals[1] = w * (pc[1] / pc[2] - T(feature[1]));
        return true;
    }
};

// sim_data.h:165-194: the per-landmark triangulation factor of the scene generator.  NOTE the residual's sign,
// feature - proj (sim_data.h:191), the opposite of ProjectFactor's; WtoC is the INVERSE camera pose (sim_data.cpp:303).
struct Triangulation {
    double q_wc[4], p_wc[3], feature[2];       // WtoC.SO3 (as a quaternion), WtoC.POS
    Triangulation(const double* cam_q, const double* cam_t, const double* f) {
        q_wc[0] = -cam_q[0]; q_wc[1] = -cam_q[1]; q_wc[2] = -cam_q[2]; q_wc[3] = cam_q[3];        // inverse rotation
        double mt[3] = {-cam_t[0], -cam_t[1], -cam_t[2]};
        QuatConjRotate(cam_q, mt, p_wc);                                                          // -R^T t
        feature[0] = f[0]; feature[1] = f[1];
    }
    static auto Create(const double* cam_q, const double* cam_t, const double* f) {
        return new ceres::AutoDiffCostFunction<Triangulation, 2, 3>(new Triangulation(cam_q, cam_t, f));
    }
    template <typename T> bool operator()(const T* const pInW, T* residuals) const {
        // pInC = WtoC.SO3 * p + WtoC.POS: rotate by q_wc = conj-rotate by its conjugate
        const T qc[4] = {T(-q_wc[0]), T(-q_wc[1]), T(-q_wc[2]), T(q_wc[3])};
        T pc[3];
        QuatConjRotate(qc, pInW, pc);
        pc[0] = pc[0] + T(p_wc[0]); pc[1] = pc[1] + T(p_wc[1]); pc[2] = pc[2] + T(p_wc[2]);
        residuals[0] = T(feature[0]) - pc[0] / pc[2];
        residuals[1] = T(feature[1]) - pc[1] / pc[2];
        return true;
    }
};

// ceres_bound.cpp:8-23
struct DemoFunctor {
    static auto Create() { return new ceres::DynamicAutoDiffCostFunction<DemoFunctor>(new DemoFunctor()); }
    template <typename T> bool operator()(T const* const* parameters, T* residuals) const {
        residuals[0] = parameters[0][0] - T(3.0);
        return true;
    }
};

// ---------------------------------------------------------------- scene file written by the python test
struct Scene {
    int nc = 0, np = 0, no = 0;
    std::vector<double> cams, pts, feat; std::vector<int> oc, op; std::vector<unsigned char> fixed;
    bool load(const char* path) {
        std::ifstream f(path, std::ios::binary);
        if (!f) return false;
        int h[3]; f.read((char*)h, sizeof h); nc = h[0]; np = h[1]; no = h[2];
        cams.resize(nc * 7); pts.resize(np * 3); feat.resize(no * 2); oc.resize(no); op.resize(no); fixed.resize(nc);
        f.read((char*)cams.data(), cams.size() * 8); f.read((char*)pts.data(), pts.size() * 8);
        f.read((char*)oc.data(), no * 4); f.read((char*)op.data(), no * 4); f.read((char*)feat.data(), feat.size() * 8);
        f.read((char*)fixed.data(), nc);
        return (bool)f;
    }
};

static void print_vec(const char* key, const double* v, int n) {
    std::printf("%s", key);
    for (int i = 0; i < n; ++i) std::printf(" %.17g", v[i]);
    std::printf("\n");
}

// test_ceres.h:98-152.  kind 0: the built-in ReprojectionFactor; 1: the user's ProjectFactor exactly as the
// reference constructs it (test_ceres.h:109-130); 2: a user functor that is NOT the reprojection factor.
static double g_far = 1e30;                 // FarScaledProjectFactor's radius
static std::vector<double> g_pts0;          // the landmarks' start positions (MovedScaledProjectFactor)
static void SolveBA(Scene& s, int kind, const char* tag, int max_iterations = 50, int print_cams = -1) {
    ceres::LocalParameterization* localParameterization = new LieLocalParameterization();
    ceres::Problem problem;
    for (int i = 0; i < s.no; ++i) {
        double* so3 = &s.cams[s.oc[i] * 7]; double* pos = so3 + 4; double* lm = &s.pts[s.op[i] * 3];
        if (kind == 0) {
            problem.AddResidualBlock(ceres::ReprojectionFactor::Create(&s.feat[i * 2]), nullptr, {so3, pos, lm});
        } else if (kind == 1) {
            auto costFunc = ProjectFactor::Create(&s.feat[i * 2]);
            costFunc->AddParameterBlock(4); costFunc->AddParameterBlock(3); costFunc->AddParameterBlock(3);
            costFunc->SetNumResiduals(2);
            problem.AddResidualBlock(costFunc, nullptr, {so3, pos, lm});
        } else if (kind == 2) {
            auto costFunc = ScaledProjectFactor::Create(&s.feat[i * 2]);
            costFunc->AddParameterBlock(4); costFunc->AddParameterBlock(3); costFunc->AddParameterBlock(3);
            costFunc->SetNumResiduals(2);
            problem.AddResidualBlock(costFunc, nullptr, {so3, pos, lm});
        } else if (kind == 3) {
            auto costFunc = FarScaledProjectFactor::Create(&s.feat[i * 2], g_far);
            costFunc->AddParameterBlock(4); costFunc->AddParameterBlock(3); costFunc->AddParameterBlock(3);
            costFunc->SetNumResiduals(2);
            problem.AddResidualBlock(costFunc, nullptr, {so3, pos, lm});
        } else {
            auto costFunc = MovedScaledProjectFactor::Create(&s.feat[i * 2], &g_pts0[s.op[i] * 3]);
            costFunc->AddParameterBlock(4); costFunc->AddParameterBlock(3); costFunc->AddParameterBlock(3);
            costFunc->SetNumResiduals(2);
            problem.AddResidualBlock(costFunc, nullptr, {so3, pos, lm});
        }
        problem.AddParameterBlock(so3, 4, localParameterization);
        if (s.fixed[s.oc[i]]) { problem.SetParameterBlockConstant(so3); problem.SetParameterBlockConstant(pos); }
    }
    ceres::Solver::Options options;
    options.num_threads = 1;
    options.linear_solver_type = ceres::SPARSE_SCHUR;
    options.max_num_iterations = max_iterations;
    ceres::Solver::Summary summary;
    ceres::Solve(options, &problem, &summary);
    std::printf("%s_path %s\n%s_report %s\n", tag, summary.execution_path.c_str(), tag, summary.FullReport().c_str());
    std::printf("%s_term %d iters %d initial %.17g final %.17g\n", tag, (int)summary.termination_type,
                (int)summary.iterations.size() - 1, summary.initial_cost, summary.final_cost);
    std::string k = std::string(tag) + "_costs";
    std::vector<double> costs;
    for (auto& it : summary.iterations) costs.push_back(it.cost);
    print_vec(k.c_str(), costs.data(), (int)costs.size());
    k = std::string(tag) + "_cams"; print_vec(k.c_str(), s.cams.data(), (print_cams < 0 ? s.nc : std::min(s.nc, print_cams)) * 7);
    k = std::string(tag) + "_pts"; print_vec(k.c_str(), s.pts.data(), std::min(s.np, 50) * 3);
}

// host-only: what Solve() would dispatch to for each kind of problem (no device needed)
static void ProbeOnly(Scene& s, int kind, const char* tag) {
    ceres::LocalParameterization* lp = new LieLocalParameterization();
    ceres::Problem problem;
    for (int i = 0; i < s.no; ++i) {
        double* so3 = &s.cams[s.oc[i] * 7]; double* pos = so3 + 4; double* lm = &s.pts[s.op[i] * 3];
        ceres::CostFunction* cf;
        if (kind == 1) { auto c = ProjectFactor::Create(&s.feat[i * 2]); c->AddParameterBlock(4); c->AddParameterBlock(3); c->AddParameterBlock(3); c->SetNumResiduals(2); cf = c; }
        else if (kind == 2) { auto c = ScaledProjectFactor::Create(&s.feat[i * 2]); c->AddParameterBlock(4); c->AddParameterBlock(3); c->AddParameterBlock(3); c->SetNumResiduals(2); cf = c; }
        else { auto c = FarScaledProjectFactor::Create(&s.feat[i * 2], g_far); c->AddParameterBlock(4); c->AddParameterBlock(3); c->AddParameterBlock(3); c->SetNumResiduals(2); cf = c; }
        problem.AddResidualBlock(cf, nullptr, {so3, pos, lm});
        problem.AddParameterBlock(so3, 4, lp);
    }
    ceres::internal::BaLayout L;
    bool ba = ceres::internal::DetectBa(problem, &L);
    if (ba) for (int rb : L.rot_block) ba = ba && ceres::internal::UsesQuaternionRightPlus(problem.blocks()[rb].local);
    double ferr = 0;
    if (ba) for (int i = 0; i < s.no; ++i) for (int k = 0; k < 2; ++k) ferr = std::fmax(ferr, std::fabs(L.feat[i * 2 + k] - s.feat[i * 2 + k]));
    std::printf("%s detected %d cams %d pts %d obs %d feature_err %.3g\n", tag, (int)ba, (int)L.rot_block.size(), (int)L.pt_block.size(),
                (int)L.obs_cam.size(), ferr);
}

int main(int argc, char** argv) {
    if (argc >= 3 && std::strcmp(argv[1], "probe") == 0) {
        Scene s;
        if (!s.load(argv[2])) { std::printf("scene_load_failed\n"); return 2; }
        { Scene a = s; ProbeOnly(a, 1, "probe_user"); }
        { Scene a = s; ProbeOnly(a, 2, "probe_scaled"); }
        if (argc >= 4) {      // a functor that IS the factor at every probe point and is not at (some of) its own data
            g_far = std::atof(argv[3]);
            { Scene a = s; ProbeOnly(a, 3, "probe_far"); }
            g_far = 1e30;
            { Scene a = s; ProbeOnly(a, 3, "probe_far_never"); }
        }
        return 0;
    }
    // ---- "verify <scene> <far>": the recognition's checks at the data -- before the solve (kind 3) and after it (kind 4)
    if (argc >= 4 && std::strcmp(argv[1], "verify") == 0) {
        Scene s;
        if (!s.load(argv[2])) { std::printf("scene_load_failed\n"); return 2; }
        g_far = std::atof(argv[3]);
        { Scene a = s; SolveBA(a, 3, "ba_far"); }
        g_pts0 = s.pts;
        { Scene a = s; SolveBA(a, 4, "ba_moved"); }
        { Scene a = s; SolveBA(a, 1, "ba_user"); }
        return 0;
    }
    // ---- "tri <scene>": sim_data.cpp:298-311 -- one ceres::Problem PER LANDMARK (cameras fixed: they are not parameter
    // blocks at all), default options, AutoDiffCostFunction<Triangulation, 2, 3> per observation
    if (argc >= 3 && std::strcmp(argv[1], "tri") == 0) {
        Scene s;
        if (!s.load(argv[2])) { std::printf("scene_load_failed\n"); return 2; }
        std::vector<std::vector<int>> obs_of(s.np);
        for (int k = 0; k < s.no; ++k) obs_of[s.op[k]].push_back(k);
        const auto t0 = std::chrono::steady_clock::now();
        int n_solved = 0, n_conv = 0, iters = 0;
        std::string path;
        for (int j = 0; j < s.np; ++j) {
            if (obs_of[j].empty()) continue;
            ceres::Problem problem;
            for (int k : obs_of[j]) {
                const double* cam = &s.cams[s.oc[k] * 7];
                auto costFunc = Triangulation::Create(cam, cam + 4, &s.feat[k * 2]);
                problem.AddResidualBlock(costFunc, nullptr, &s.pts[j * 3]);
            }
            ceres::Solver::Options options;
            ceres::Solver::Summary summary;
            ceres::Solve(options, &problem, &summary);
            ++n_solved;
            n_conv += summary.termination_type == ceres::CONVERGENCE ? 1 : 0;
            iters += (int)summary.iterations.size() - 1;
            path = summary.execution_path;
        }
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("tri_summary problems %d converged %d iterations %d seconds %.6f path %s\n", n_solved, n_conv, iters, secs, path.c_str());
        print_vec("tri_pts", s.pts.data(), s.np * 3);
        return 0;
    }
    // ---- "big <scene> <iterations>": ONLY the reference's unchanged BA call site, at any size (config C5)
    if (argc >= 4 && std::strcmp(argv[1], "big") == 0) {
        Scene s;
        if (!s.load(argv[2])) { std::printf("scene_load_failed\n"); return 2; }
        SolveBA(s, 1, "ba_user", std::atoi(argv[3]), 1000000);
        return 0;
    }
    // ---- "loss": a residual block with a non-null LossFunction must be REFUSED (robust losses are not implemented; the reference
    // passes nullptr, test_ceres.h:120): FAILURE, parameters untouched, the reason in the message -- no device needed
    if (argc >= 2 && std::strcmp(argv[1], "loss") == 0) {
        struct Huber : ceres::LossFunction {};
        ceres::Problem problem;
        auto costFunc = DemoFunctor::Create();
        costFunc->AddParameterBlock(1);
        costFunc->SetNumResiduals(1);
        double x = 0.5;
        problem.AddResidualBlock(costFunc, new Huber(), &x);
        ceres::Solver::Options options;
        ceres::Solver::Summary summary;
        ceres::Solve(options, &problem, &summary);
        std::printf("loss x %.17g term %d losses %d msg %s\n", x, (int)summary.termination_type, problem.NumLossFunctions(), summary.message.c_str());
        // ownership: a cost function added to two residual blocks is deleted ONCE with the problem; one added to a problem that
        // does not take ownership is left alone
        static int deleted = 0;
        struct Counted : ceres::SizedCostFunction<1, 1> {
            ~Counted() override { ++deleted; }
            bool Evaluate(double const* const* p, double* r, double** J) const override { r[0] = p[0][0]; if (J && J[0]) J[0][0] = 1.0; return true; }
        };
        double a = 1.0, b = 2.0;
        {
            ceres::Problem owner;
            auto* shared = new Counted();
            owner.AddResidualBlock(shared, nullptr, &a);
            owner.AddResidualBlock(shared, nullptr, &b);
            owner.AddResidualBlock(new Counted(), nullptr, &a);
        }
        const int after_owner = deleted;
        Counted* kept = new Counted();
        {
            ceres::Problem::Options po;
            po.cost_function_ownership = ceres::DO_NOT_TAKE_OWNERSHIP;
            ceres::Problem borrower(po);
            borrower.AddResidualBlock(kept, nullptr, &a);
        }
        const int after_borrower = deleted;
        delete kept;
        std::printf("ownership deleted_by_owner %d deleted_by_borrower %d\n", after_owner, after_borrower - after_owner);
        return 0;
    }
    // ---- "time_ba <scene> <max_iterations> <reps> [threads]": wall-clock of the reference's BA call site as the reference times it
    // (test_ceres.h:103-104,149: the timer starts in front of the problem construction) -- construction and Solve() separately,
    // Solve()'s own phases from Summary::phases.  One line per repetition; the caller takes the median.
    if (argc >= 5 && std::strcmp(argv[1], "time_ba") == 0) {
        Scene s0;
        if (!s0.load(argv[2])) { std::printf("scene_load_failed\n"); return 2; }
        const int max_it = std::atoi(argv[3]), reps = std::atoi(argv[4]), threads = argc > 5 ? std::atoi(argv[5]) : 1;
        for (int rep = 0; rep < reps; ++rep) {
            Scene s = s0;
            const auto t0 = std::chrono::steady_clock::now();
            double t_build, t_solve, t_destroy;
            ceres::Solver::Summary summary;
            {
                ceres::LocalParameterization* localParameterization = new LieLocalParameterization();
                ceres::Problem problem;
                for (int i = 0; i < s.no; ++i) {
                    double* so3 = &s.cams[s.oc[i] * 7]; double* pos = so3 + 4; double* lm = &s.pts[s.op[i] * 3];
                    auto costFunc = ProjectFactor::Create(&s.feat[i * 2]);
                    costFunc->AddParameterBlock(4); costFunc->AddParameterBlock(3); costFunc->AddParameterBlock(3);
                    costFunc->SetNumResiduals(2);
                    problem.AddResidualBlock(costFunc, nullptr, {so3, pos, lm});
                    problem.AddParameterBlock(so3, 4, localParameterization);
                    if (s.fixed[s.oc[i]]) { problem.SetParameterBlockConstant(so3); problem.SetParameterBlockConstant(pos); }
                }
                ceres::Solver::Options options;
                options.num_threads = threads;
                options.linear_solver_type = ceres::SPARSE_SCHUR;
                options.max_num_iterations = max_it;
                const auto t1 = std::chrono::steady_clock::now();
                ceres::Solve(options, &problem, &summary);
                const auto t2 = std::chrono::steady_clock::now();
                t_build = std::chrono::duration<double>(t1 - t0).count();
                t_solve = std::chrono::duration<double>(t2 - t1).count();
            }
            t_destroy = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() - t_build - t_solve;
            const auto& ph = summary.phases;
            std::printf("time_ba_%d path %s term %d iters %d initial %.17g final %.17g build %.6f solve %.6f destroy %.6f recognise %.6f pack %.6f "
                        "engine_create %.6f device_solve %.6f write_back %.6f verify %.6f resolve %.6f minimizer %.6f\n",
                        rep, summary.execution_path.c_str(), (int)summary.termination_type, (int)summary.iterations.size() - 1, summary.initial_cost,
                        summary.final_cost, t_build, t_solve, t_destroy, ph.recognise, ph.pack, ph.engine_create, ph.device_solve, ph.write_back,
                        ph.verify, ph.resolve, summary.minimizer_time_in_seconds);
            if (rep == reps - 1) print_vec("time_ba_cams", s.cams.data(), std::min(s.nc, 1000000) * 7);
        }
        return 0;
    }
    // ---- "time_pnp <pnp file> <reps>": the reference's PUBLISHED workload (st17-ceres/img/release.png: 0.223 / 0.138 / 0.124 ms for
    // SolvePnPWith{DynamicAutoDiff, AutoDiff, SizedCostFunction}, solver.hpp:247-385), timed as the reference times it (solver.hpp:253-288:
    // the timer spans problem construction + Solve, countTime = true: no callback).  Prints every repetition's wall time in ms.
    if (argc >= 4 && std::strcmp(argv[1], "time_pnp") == 0) {
        std::ifstream f(argv[2], std::ios::binary);
        int n; f.read((char*)&n, 4);
        double truth[7], init[7];
        f.read((char*)truth, 56); f.read((char*)init, 56);
        std::vector<CorrPair> data(n);
        for (auto& c : data) { f.read((char*)c.point, 24); f.read((char*)c.feature, 16); }
        if (!f) { std::printf("pnp_load_failed\n"); return 2; }
        const int reps = std::atoi(argv[3]);
        const char* names[3] = {"pnp_dyn", "pnp_auto", "pnp_sized"};
        for (int variant = 0; variant < 3; ++variant) {
            std::vector<double> ms;
            double pose[7] = {0}; int iters = 0, term = -1; double final_cost = 0; std::string path;
            for (int rep = 0; rep < reps; ++rep) {
                const auto t0 = std::chrono::steady_clock::now();
                ceres::Solver::Summary summary;
                if (variant < 2) {
                    double SO3[4], POS[3]; std::memcpy(SO3, init, 32); std::memcpy(POS, init + 4, 24);
                    ceres::LocalParameterization* lp = new LieLocalParameterization();
                    ceres::Problem problem;
                    for (const auto& item : data) {
                        if (variant == 0) {
                            auto costFunc = PnPDynamicAutoDiffFunctor::Create(item);
                            costFunc->AddParameterBlock(4); costFunc->AddParameterBlock(3); costFunc->SetNumResiduals(2);
                            problem.AddResidualBlock(costFunc, nullptr, {SO3, POS});
                        } else problem.AddResidualBlock(PnPAutoDiffFunctor::Create(item), nullptr, {SO3, POS});
                        problem.AddParameterBlock(SO3, 4, lp);
                    }
                    ceres::Solver::Options options; options.num_threads = 1; options.linear_solver_type = ceres::DENSE_QR;
                    ceres::Solve(options, &problem, &summary);
                    std::memcpy(pose, SO3, 32); std::memcpy(pose + 4, POS, 24);
                } else {
                    double so3[3], POS[3]; So3Log(init, so3); std::memcpy(POS, init + 4, 24);
                    ceres::LocalParameterization* lp = new LieR3LocalParameterization();
                    ceres::Problem problem;
                    for (const auto& item : data) {
                        problem.AddResidualBlock(new PnPSizedCostFunction(item, true), nullptr, {so3, POS});
                        problem.AddParameterBlock(so3, 3, lp);
                    }
                    ceres::Solver::Options options; options.num_threads = 1; options.linear_solver_type = ceres::DENSE_QR;
                    ceres::Solve(options, &problem, &summary);
                    So3Exp(so3, pose); std::memcpy(pose + 4, POS, 24);
                }
                ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
                iters = (int)summary.iterations.size() - 1; term = (int)summary.termination_type; final_cost = summary.final_cost; path = summary.execution_path;
            }
            std::printf("%s path %s term %d iters %d final %.17g\n", names[variant], path.c_str(), term, iters, final_cost);
            print_vec((std::string(names[variant]) + "_ms").c_str(), ms.data(), (int)ms.size());
            print_vec((std::string(names[variant]) + "_pose").c_str(), pose, 7);
        }
        return 0;
    }
    if (argc >= 3 && std::strcmp(argv[1], "pg") == 0) {
        // a pose graph through the operator API: one 7-double block per pose with the SE3 right-plus chart, one RelativePoseFactor
        // per edge (build-defined, BASELINE config C4).  argv[3] = "dense": force the generic host path (checks Evaluate / the chart)
        std::ifstream f(argv[2], std::ios::binary);
        int n = 0, m = 0;
        f.read((char*)&n, 4); f.read((char*)&m, 4);
        std::vector<double> poses((size_t)n * 7), meas((size_t)m * 7);
        std::vector<int> ei(m), ej(m);
        std::vector<unsigned char> fixed(n);
        f.read((char*)poses.data(), poses.size() * 8); f.read((char*)ei.data(), m * 4); f.read((char*)ej.data(), m * 4);
        f.read((char*)meas.data(), meas.size() * 8); f.read((char*)fixed.data(), n);
        if (!f) { std::printf("pg_load_failed\n"); return 2; }
        ceres::LocalParameterization* chart = new ceres::SE3RightPlus();
        ceres::Problem problem;
        for (int e = 0; e < m; ++e)
            problem.AddResidualBlock(ceres::RelativePoseFactor::Create(&meas[(size_t)e * 7]), nullptr, {&poses[(size_t)ei[e] * 7], &poses[(size_t)ej[e] * 7]});
        for (int k = 0; k < n; ++k) {
            problem.AddParameterBlock(&poses[(size_t)k * 7], 7, chart);
            if (fixed[k]) problem.SetParameterBlockConstant(&poses[(size_t)k * 7]);
        }
        ceres::Solver::Options options;
        options.num_threads = 1;
        options.force_callback_path = (argc > 3 && std::strcmp(argv[3], "dense") == 0);
        ceres::Solver::Summary summary;
        ceres::Solve(options, &problem, &summary);
        std::printf("pg_path %s\npg_msg %s\n", summary.execution_path.c_str(), summary.message.c_str());
        std::printf("pg_term %d iters %d initial %.17g final %.17g\n", (int)summary.termination_type, (int)summary.iterations.size() - 1,
                    summary.initial_cost, summary.final_cost);
        std::vector<double> costs;
        for (auto& it : summary.iterations) costs.push_back(it.cost);
        print_vec("pg_costs", costs.data(), (int)costs.size());
        print_vec("pg_poses", poses.data(), n * 7);
        return 0;
    }
    if (argc >= 4 && std::strcmp(argv[1], "generic_big") == 0) {      // a BA-shaped problem with a factor the probe rejects, any size
        Scene s;
        if (!s.load(argv[2])) { std::printf("scene_load_failed\n"); return 2; }
        SolveBA(s, 2, "ba_generic", std::atoi(argv[3]), 1000000);
        return 0;
    }
    // ---- ceres_bound.cpp:25-68
    for (int bounded = 0; bounded < 2; ++bounded) {
        ceres::Problem problem;
        auto costFunc = DemoFunctor::Create();
        costFunc->AddParameterBlock(1);
        costFunc->SetNumResiduals(1);
        double x = 0.0;
        problem.AddResidualBlock(costFunc, nullptr, &x);
        if (bounded) { problem.SetParameterLowerBound(&x, 0, -2.0); problem.SetParameterUpperBound(&x, 0, 2.0); }
        ceres::Solver::Options options;
        options.num_threads = 1;
        options.linear_solver_type = ceres::DENSE_QR;
        ceres::Solver::Summary summary;
        ceres::Solve(options, &problem, &summary);
        std::printf("bound_%d x %.17g term %d path %s msg %s\n", bounded, x, (int)summary.termination_type,
                    summary.execution_path.c_str(), summary.message.c_str());
    }
    if (argc < 3) return 0;
    // ---- PnP: solver.hpp:247-385.  file: n, true pose(7), init pose(7), n*(point3, feature2)
    {
        std::ifstream f(argv[1], std::ios::binary);
        int n; f.read((char*)&n, 4);
        double truth[7], init[7];
        f.read((char*)truth, 56); f.read((char*)init, 56);
        std::vector<CorrPair> data(n);
        for (auto& c : data) { f.read((char*)c.point, 24); f.read((char*)c.feature, 16); }
        print_vec("pnp_truth", truth, 7);
        // SolvePnPWithDynamicAutoDiff (with the visual callback + update_state_every_iteration branch)
        {
            double SO3[4], POS[3]; std::memcpy(SO3, init, 32); std::memcpy(POS, init + 4, 24);
            ceres::LocalParameterization* lp = new LieLocalParameterization();
            ceres::Problem problem;
            for (const auto& item : data) {
                auto costFunc = PnPDynamicAutoDiffFunctor::Create(item);
                costFunc->AddParameterBlock(4); costFunc->AddParameterBlock(3); costFunc->SetNumResiduals(2);
                problem.AddResidualBlock(costFunc, nullptr, {SO3, POS});
                problem.AddParameterBlock(SO3, 4, lp);
            }
            ceres::Solver::Options options;
            auto* cb = new VisualCallBack(SO3, POS);
            options.callbacks.push_back(cb);
            options.update_state_every_iteration = true;
            options.num_threads = 1; options.linear_solver_type = ceres::DENSE_QR;
            ceres::Solver::Summary summary;
            ceres::Solve(options, &problem, &summary);
            std::printf("pnp_dyn iters %d initial %.17g final %.17g term %d callbacks %d first_cb_x %.17g last_cb_x %.17g\n",
                        (int)summary.iterations.size() - 1, summary.initial_cost, summary.final_cost, (int)summary.termination_type,
                        (int)cb->seen.size() / 2, cb->seen.empty() ? 0.0 : cb->seen[0], cb->seen.empty() ? 0.0 : cb->seen[cb->seen.size() - 2]);
            double out[7]; std::memcpy(out, SO3, 32); std::memcpy(out + 4, POS, 24); print_vec("pnp_dyn_pose", out, 7);
            delete cb;
        }
        // SolvePnPWithAutoDiff
        {
            double SO3[4], POS[3]; std::memcpy(SO3, init, 32); std::memcpy(POS, init + 4, 24);
            ceres::LocalParameterization* lp = new LieLocalParameterization();
            ceres::Problem problem;
            for (const auto& item : data) {
                problem.AddResidualBlock(PnPAutoDiffFunctor::Create(item), nullptr, {SO3, POS});
                problem.AddParameterBlock(SO3, 4, lp);
            }
            ceres::Solver::Options options; options.num_threads = 1; options.linear_solver_type = ceres::DENSE_QR;
            ceres::Solver::Summary summary;
            ceres::Solve(options, &problem, &summary);
            std::printf("pnp_auto iters %d final %.17g term %d\n", (int)summary.iterations.size() - 1, summary.final_cost, (int)summary.termination_type);
            double out[7]; std::memcpy(out, SO3, 32); std::memcpy(out + 4, POS, 24); print_vec("pnp_auto_pose", out, 7);
        }
        // SolvePnPWithSizedCostFunction, correct and reference rotation Jacobian
        for (int ref = 0; ref < 2; ++ref) {
            double so3[3], POS[3]; So3Log(init, so3); std::memcpy(POS, init + 4, 24);
            ceres::LocalParameterization* lp = new LieR3LocalParameterization();
            ceres::Problem problem;
            for (const auto& item : data) {
                problem.AddResidualBlock(new PnPSizedCostFunction(item, ref != 0), nullptr, {so3, POS});
                problem.AddParameterBlock(so3, 3, lp);
            }
            ceres::Solver::Options options; options.num_threads = 1; options.linear_solver_type = ceres::DENSE_QR;
            ceres::Solver::Summary summary;
            ceres::Solve(options, &problem, &summary);
            double out[7]; So3Exp(so3, out); std::memcpy(out + 4, POS, 24);
            std::printf("pnp_sized_%d iters %d final %.17g term %d\n", ref, (int)summary.iterations.size() - 1, summary.final_cost, (int)summary.termination_type);
            print_vec(ref ? "pnp_sized_1_pose" : "pnp_sized_0_pose", out, 7);
        }
    }
    // ---- BA: test_ceres.h:98-152
    Scene s;
    if (!s.load(argv[2])) { std::printf("scene_load_failed\n"); return 2; }
    { Scene a = s; SolveBA(a, 0, "ba_builtin"); }
    { Scene a = s; SolveBA(a, 1, "ba_user"); }            // the reference's unchanged call site
    if (argc > 3) { Scene s2; if (s2.load(argv[3])) SolveBA(s2, 2, "ba_generic"); }
    return 0;
}

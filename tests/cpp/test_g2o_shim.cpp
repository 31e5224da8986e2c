// Restates st20-g2o/src/include/test_g2o.h:19-147 (VertexCamera, VertexLandmark, EdgeProject,
// SolveWithG2O) against include/stba/g2o.h.  Sophus/Eigen types are replaced by plain structs plus
// the three Traits specialisations a maintainer would add.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>

#include "stba/g2o.h"
namespace g2o = stba_g2o;
using std::istream;
using std::ostream;

struct OptPose { double q[4] = {0, 0, 0, 1}; double t[3] = {0, 0, 0}; };   // sim_data.h:22-36
struct Vec3 { double v[3] = {0, 0, 0}; };
struct Vec2 { double v[2] = {0, 0}; };
namespace stba_g2o {
template <> struct Traits<OptPose> { static void get(const OptPose& p, double* o) { std::memcpy(o, p.q, 32); std::memcpy(o + 4, p.t, 24); }
                                     static void set(OptPose& p, const double* i) { std::memcpy(p.q, i, 32); std::memcpy(p.t, i + 4, 24); } };
template <> struct Traits<Vec3> { static void get(const Vec3& p, double* o) { std::memcpy(o, p.v, 24); } static void set(Vec3& p, const double* i) { std::memcpy(p.v, i, 24); } };
template <> struct Traits<Vec2> { static void get(const Vec2& p, double* o) { std::memcpy(o, p.v, 16); } static void set(Vec2& p, const double* i) { std::memcpy(p.v, i, 16); } };
}

static void QuatMul(const double* a, const double* b, double* o) {
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
static void ConjRotate(const double* q, const double* v, double* o) {
    const double u0 = -q[0], u1 = -q[1], u2 = -q[2], w = q[3];
    const double a0 = 2 * (u1 * v[2] - u2 * v[1]), a1 = 2 * (u2 * v[0] - u0 * v[2]), a2 = 2 * (u0 * v[1] - u1 * v[0]);
    o[0] = v[0] + w * a0 + (u1 * a2 - u2 * a1); o[1] = v[1] + w * a1 + (u2 * a0 - u0 * a2); o[2] = v[2] + w * a2 + (u0 * a1 - u1 * a0);
}

struct VertexCamera : public g2o::BaseVertex<6, OptPose> {          // test_g2o.h:19-48
    bool read(istream&) override { return false; }
    bool write(ostream&) const override { return false; }
    Vec2 Project(const Vec3& landmark) const {
        double d[3] = {landmark.v[0] - _estimate.t[0], landmark.v[1] - _estimate.t[1], landmark.v[2] - _estimate.t[2]}, pc[3];
        ConjRotate(_estimate.q, d, pc);
        Vec2 r; r.v[0] = pc[0] / pc[2]; r.v[1] = pc[1] / pc[2];
        return r;
    }
protected:
    void oplusImpl(const g2o::number_t* v) override {                 // :36-39
        const double th = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        const double im = th < 1e-12 ? 0.5 : std::sin(0.5 * th) / th;
        const double e[4] = {im * v[0], im * v[1], im * v[2], std::cos(0.5 * th)};
        double q[4]; QuatMul(_estimate.q, e, q); std::memcpy(_estimate.q, q, 32);
        for (int i = 0; i < 3; ++i) _estimate.t[i] += v[3 + i];
    }
    void setToOriginImpl() override { _estimate = OptPose(); }
};

struct VertexLandmark : public g2o::BaseVertex<3, Vec3> {            // :50-71
    bool read(istream&) override { return true; }
    bool write(ostream&) const override { return true; }
protected:
    void oplusImpl(const g2o::number_t* v) override { for (int i = 0; i < 3; ++i) _estimate.v[i] += v[i]; }
    void setToOriginImpl() override { _estimate = Vec3(); }
};

struct EdgeProject : public g2o::BaseBinaryEdge<2, Vec2, VertexCamera, VertexLandmark> {   // :73-92
    void computeError() override {
        auto cameraPoseVertex = dynamic_cast<VertexCamera*>(_vertices[0]);
        auto landmarkVertex = dynamic_cast<VertexLandmark*>(_vertices[1]);
        auto p = cameraPoseVertex->Project(landmarkVertex->estimate());
        _error.v[0] = p.v[0] - _measurement.v[0]; _error.v[1] = p.v[1] - _measurement.v[1];
    }
    bool read(istream&) override { return true; }
    bool write(ostream&) const override { return true; }
};

// NOT the reference's edge: a residual of another kind hidden among the projection edges (negative test of the probe)
struct EdgeProjectScaled : public EdgeProject {
    void computeError() override { EdgeProject::computeError(); _error.v[0] *= 2.0; _error.v[1] *= 2.0; }
};

int main(int argc, char** argv) {
    if (argc < 2) return 1;
    std::ifstream f(argv[1], std::ios::binary);
    int h[3]; f.read((char*)h, sizeof h);
    const int nc = h[0], np = h[1], no = h[2];
    std::vector<double> cams(nc * 7), pts(np * 3), feat(no * 2); std::vector<int> oc(no), op(no); std::vector<unsigned char> fixed(nc);
    f.read((char*)cams.data(), cams.size() * 8); f.read((char*)pts.data(), pts.size() * 8);
    f.read((char*)oc.data(), no * 4); f.read((char*)op.data(), no * 4); f.read((char*)feat.data(), feat.size() * 8); f.read((char*)fixed.data(), nc);

    // SolveWithG2O, test_g2o.h:94-147
    using BlockSolverType = g2o::BlockSolver<g2o::BlockSolverTraits<6, 3>>;
    using LinearSolverType = g2o::LinearSolverCSparse<BlockSolverType::PoseMatrixType>;
    auto solver = new g2o::OptimizationAlgorithmLevenberg(g2o::make_unique<BlockSolverType>(g2o::make_unique<LinearSolverType>()));
    g2o::SparseOptimizer optimizer;
    optimizer.setAlgorithm(solver);
    optimizer.setVerbose(false);
    std::vector<VertexCamera*> cameraVertexVec;
    std::vector<VertexLandmark*> landmarkVertexVec;
    for (int i = 0; i < nc; ++i) {
        OptPose camera; std::memcpy(camera.q, &cams[i * 7], 32); std::memcpy(camera.t, &cams[i * 7 + 4], 24);
        auto cameraVertex = new VertexCamera();
        cameraVertex->setId(i);
        cameraVertex->setEstimate(camera);
        if (argc > 2 && std::strcmp(argv[2], "fix") == 0 && fixed[i]) cameraVertex->setFixed(true);       // (the reference fixes none; optional here)
        optimizer.addVertex(cameraVertex);
        cameraVertexVec.push_back(cameraVertex);
    }
    std::vector<std::vector<int>> obs_of(np);
    for (int k = 0; k < no; ++k) obs_of[op[k]].push_back(k);
    for (int i = 0; i < np; ++i) {
        Vec3 lm; std::memcpy(lm.v, &pts[i * 3], 24);
        auto* landmarkVertex = new VertexLandmark();
        landmarkVertex->setId(i + nc);
        landmarkVertex->setEstimate(lm);
        landmarkVertex->setMarginalized(true);
        optimizer.addVertex(landmarkVertex);
        landmarkVertexVec.push_back(landmarkVertex);
        for (int k : obs_of[i]) {
            // ("odd": ONE edge in the middle of the graph is of another type)
            EdgeProject* e = (argc > 2 && std::strcmp(argv[2], "odd") == 0 && k == no / 2) ? new EdgeProjectScaled : new EdgeProject;
            e->setVertex(0, cameraVertexVec.at(oc[k]));
            e->setVertex(1, landmarkVertex);
            Vec2 z; z.v[0] = feat[k * 2]; z.v[1] = feat[k * 2 + 1];
            e->setMeasurement(z);
            e->setInformation(1.0);
            optimizer.addEdge(e);
        }
    }
    optimizer.initializeOptimization();
    const int it = optimizer.optimize(40);
    std::printf("g2o_iters %d chi2 %.17g msg [%s]\n", it, optimizer.chi2(), optimizer.message().c_str());
    std::printf("g2o_cams");
    for (auto* v : cameraVertexVec) { double o[7]; g2o::Traits<OptPose>::get(v->estimate(), o); for (double x : o) std::printf(" %.17g", x); }
    std::printf("\ng2o_pts");
    for (int i = 0; i < std::min(np, 50); ++i) for (double x : landmarkVertexVec[i]->estimate().v) std::printf(" %.17g", x);
    std::printf("\n");
    return 0;
}

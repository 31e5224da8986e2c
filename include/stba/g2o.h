// stba/g2o.h -- the slice of the g2o API that st20-g2o/src/include/test_g2o.h uses
// (BaseVertex / BaseBinaryEdge / SparseOptimizer / OptimizationAlgorithmLevenberg /
// BlockSolver<BlockSolverTraits<6,3>> / LinearSolverCSparse), header-only over the C ABI (stba.h).
//
//     #include "stba/g2o.h"
//     namespace g2o = stba_g2o;
//
// The reference's vertex / edge classes stay user code (VertexCamera::oplusImpl test_g2o.h:36-39,
// VertexLandmark::oplusImpl :60-63, EdgeProject::computeError :75-80).  SparseOptimizer::optimize(n)
// recognises the bundle-adjustment graph they form -- 6-dof pose vertices, 3-dof landmark vertices,
// binary 2-residual edges -- VERIFIES numerically that the user's oplus and computeError are the
// built-in ones (q <- q (x) exp(v[0:3]), t <- t + v[3:6]; L <- L + v; e = proj(R^T (L - t)) - z) and
// then runs the whole optimisation in the device-resident engine (setMarginalized(true) on the
// landmarks = the Schur complement).  Anything else is rejected loudly: there is no CPU solver here.
//
// The estimate / measurement types are the user's own (the reference: OptPose with Sophus members,
// Eigen::Vector3d, Eigen::Vector2d); the engine needs their raw doubles, which a trait provides:
//     template <> struct stba_g2o::Traits<OptPose> { static void get(const OptPose&, double* q4t3);
//                                                    static void set(OptPose&, const double* q4t3); };
// (INTEGRATION.md shows the three specialisations for the reference's types.)
#ifndef STBA_G2O_H
#define STBA_G2O_H

#include <cmath>
#include <cstdio>
#include <cstring>
#include <iosfwd>
#include <memory>
#include <string>
#include <typeinfo>
#include <utility>
#include <vector>

#include "../stba.h"

namespace stba_g2o {

typedef double number_t;
using std::istream;
using std::ostream;

template <class T, class... A>
std::unique_ptr<T> make_unique(A&&... a) { return std::unique_ptr<T>(new T(std::forward<A>(a)...)); }

// raw-double view of user types; specialise for estimate / measurement types
template <class T> struct Traits;   // { static void get(const T&, double*); static void set(T&, const double*); }

struct Vertex {
    virtual ~Vertex() = default;
    virtual int dimension() const = 0;
    virtual void oplus(const number_t* v) = 0;
    virtual void get_raw(double* out) const = 0;
    virtual void set_raw(const double* in) = 0;
    virtual void push() = 0;      // save / restore the estimate (used by the semantic probe)
    virtual void pop() = 0;
    int id() const { return _id; }
    void setId(int i) { _id = i; }
    void setFixed(bool f) { _fixed = f; }
    bool fixed() const { return _fixed; }
    void setMarginalized(bool m) { _marginalized = m; }
    bool marginalized() const { return _marginalized; }
protected:
    int _id = -1;
    bool _fixed = false, _marginalized = false;
};

template <int D, typename T>
class BaseVertex : public Vertex {
public:
    static const int Dimension = D;
    typedef T EstimateType;
    int dimension() const override { return D; }
    const T& estimate() const { return _estimate; }
    void setEstimate(const T& e) { _estimate = e; }
    void oplus(const number_t* v) override { oplusImpl(v); }
    void get_raw(double* out) const override { Traits<T>::get(_estimate, out); }
    void set_raw(const double* in) override { Traits<T>::set(_estimate, in); }
    void push() override { _backup = _estimate; }
    void pop() override { _estimate = _backup; }
    virtual bool read(istream& is) = 0;
    virtual bool write(ostream& os) const = 0;
protected:
    virtual void oplusImpl(const number_t* v) = 0;
    virtual void setToOriginImpl() = 0;
    T _estimate{};
    T _backup{};
};

struct Edge {
    virtual ~Edge() = default;
    virtual int dimension() const = 0;
    virtual void computeError() = 0;
    virtual void error_raw(double* out) const = 0;
    virtual void measurement_raw(double* out) const = 0;
    void setVertex(size_t i, Vertex* v) { if (_vertices.size() <= i) _vertices.resize(i + 1, nullptr); _vertices[i] = v; }
    const std::vector<Vertex*>& vertices() const { return _vertices; }
protected:
    std::vector<Vertex*> _vertices;
};

template <int D, typename E, typename VertexXi, typename VertexXj>
class BaseBinaryEdge : public Edge {
public:
    static const int Dimension = D;
    BaseBinaryEdge() { _vertices.resize(2, nullptr); }
    int dimension() const override { return D; }
    void setMeasurement(const E& m) { _measurement = m; }
    const E& measurement() const { return _measurement; }
    template <class M> void setInformation(const M&) {}   // the reference sets identity (test_g2o.h:129)
    void error_raw(double* out) const override { Traits<E>::get(_error, out); }
    void measurement_raw(double* out) const override { Traits<E>::get(_measurement, out); }
    virtual bool read(istream& is) = 0;
    virtual bool write(ostream& os) const = 0;
protected:
    E _measurement{};
    E _error{};
};

// ---- solver scaffolding (types only: the engine replaces what they would configure) --------
template <int P, int L> struct BlockSolverTraits { struct PoseMatrixType {}; static const int PoseDim = P, LandmarkDim = L; };
template <class MatrixType> struct LinearSolverCSparse {};
template <class TraitsT> struct BlockSolver {
    typedef typename TraitsT::PoseMatrixType PoseMatrixType;
    template <class LS> explicit BlockSolver(std::unique_ptr<LS>) {}
};
struct OptimizationAlgorithm { virtual ~OptimizationAlgorithm() = default; };
struct OptimizationAlgorithmLevenberg : OptimizationAlgorithm {
    template <class BS> explicit OptimizationAlgorithmLevenberg(std::unique_ptr<BS>) {}
};

class SparseOptimizer {
public:
    ~SparseOptimizer() { for (auto* v : _vertices) delete v; for (auto* e : _edges) delete e; delete _alg; }
    void setAlgorithm(OptimizationAlgorithm* a) { delete _alg; _alg = a; }
    void setVerbose(bool v) { _verbose = v; }
    bool addVertex(Vertex* v) { _vertices.push_back(v); return true; }
    bool addEdge(Edge* e) { _edges.push_back(e); return true; }
    bool initializeOptimization() { _initialised = true; return true; }
    double chi2() const { return _chi2; }
    const std::string& message() const { return _message; }

    // returns the number of iterations performed (0 on failure, message() says why)
    int optimize(int iterations) {
        _message.clear();
        if (!_initialised || _vertices.empty() || _edges.empty()) { _message = "graph not initialised / empty"; return 0; }
        std::vector<int> cam_of(_vertices.size(), -1), pt_of(_vertices.size(), -1);
        std::vector<Vertex*> cams, pts;
        for (size_t i = 0; i < _vertices.size(); ++i) {
            if (_vertices[i]->dimension() == 6) { cam_of[i] = (int)cams.size(); cams.push_back(_vertices[i]); }
            else if (_vertices[i]->dimension() == 3) { pt_of[i] = (int)pts.size(); pts.push_back(_vertices[i]); }
            else { _message = "unsupported vertex dimension (only 6-dof poses and 3-dof landmarks run on the device)"; return 0; }
        }
        std::vector<size_t> index_of_ptr;
        auto find = [&](Vertex* v) -> int { for (size_t i = 0; i < _vertices.size(); ++i) if (_vertices[i] == v) return (int)i; return -1; };
        // vertex ids are dense in the reference; fall back to a pointer search otherwise
        auto vidx = [&](Vertex* v) -> int { const int id = v->id(); if (id >= 0 && id < (int)_vertices.size() && _vertices[id] == v) return id; return find(v); };
        const int no = (int)_edges.size();
        std::vector<int> oc(no), op(no);
        std::vector<double> feat((size_t)no * 2);
        for (int k = 0; k < no; ++k) {
            Edge* e = _edges[k];
            if (e->dimension() != 2 || e->vertices().size() != 2) { _message = "unsupported edge (only binary 2-residual projection edges)"; return 0; }
            const int a = vidx(e->vertices()[0]), b = vidx(e->vertices()[1]);
            if (a < 0 || b < 0 || cam_of[a] < 0 || pt_of[b] < 0) { _message = "edge does not connect (pose, landmark)"; return 0; }
            oc[k] = cam_of[a]; op[k] = pt_of[b];
            e->measurement_raw(&feat[(size_t)k * 2]);
            // EVERY edge must be the reprojection residual (a graph with one edge of another type must not be treated as pure
            // bundle adjustment): one computeError per edge at its current estimates, host work, no device involved
            if (!probe_edge(e, k)) return 0;
        }
        // the manifold updates: once per vertex CLASS (typeid) that occurs in the graph
        {
            std::vector<const std::type_info*> seen;
            auto fresh = [&](Vertex* v) { for (auto* t : seen) if (*t == typeid(*v)) return false; seen.push_back(&typeid(*v)); return true; };
            for (Vertex* v : cams) if (fresh(v) && !probe_pose_oplus(v)) return 0;
            for (Vertex* v : pts) if (fresh(v) && !probe_landmark_oplus(v)) return 0;
        }
        const int nc = (int)cams.size(), np = (int)pts.size();
        std::vector<double> c7((size_t)nc * 7), p3((size_t)np * 3);
        std::vector<unsigned char> cfix((size_t)nc * 6, 0), pfix((size_t)np, 0);
        for (int c = 0; c < nc; ++c) { cams[c]->get_raw(&c7[(size_t)c * 7]); if (cams[c]->fixed()) for (int a = 0; a < 6; ++a) cfix[c * 6 + a] = 1; }
        for (int j = 0; j < np; ++j) { pts[j]->get_raw(&p3[(size_t)j * 3]); pfix[j] = pts[j]->fixed() ? 1 : 0; }
        stba_ba* ba = nullptr;
        if (stba_ba_create(&ba, nc, np, no, c7.data(), p3.data(), oc.data(), op.data(), feat.data(), cfix.data(), pfix.data(), nullptr) != STBA_OK) {
            _message = std::string("stba_ba_create: ") + stba_last_error();
            return 0;
        }
        stba_lm_options o;
        stba_lm_default_options(&o);
        o.max_num_iterations = iterations;
        stba_lm_summary s;
        std::vector<double> trace((size_t)(iterations + 1) * STBA_TRACE_COLS, 0.0);
        const int rc = stba_ba_solve(ba, &o, &s, trace.data(), nullptr, nullptr);
        int done = 0;
        if (rc == STBA_OK) {
            const std::vector<double> c7_start = c7, p3_start = p3;
            stba_ba_get_params(ba, c7.data(), p3.data());
            for (int c = 0; c < nc; ++c) cams[c]->set_raw(&c7[(size_t)c * 7]);
            for (int j = 0; j < np; ++j) pts[j]->set_raw(&p3[(size_t)j * 3]);
            // the edges were the reprojection residual at the START point; they must still be at the point the solve ended at (a
            // computeError with a clamp or a weight that sets in on the way is not what the device minimised): if one is not, the
            // estimates go back to where they were and nothing is reported as optimised
            for (int k = 0; k < no; ++k)
                if (!probe_edge(_edges[k], k)) {
                    for (int c = 0; c < nc; ++c) cams[c]->set_raw(&c7_start[(size_t)c * 7]);
                    for (int j = 0; j < np; ++j) pts[j]->set_raw(&p3_start[(size_t)j * 3]);
                    _message += " (at the solution: the estimates were put back)";
                    stba_ba_destroy(ba);
                    return 0;
                }
            _chi2 = 2.0 * s.final_cost;   // g2o reports chi^2 = sum r^2, Ceres 1/2 sum r^2
            done = s.num_iterations;
            if (_verbose)
                for (int i = 0; i <= s.num_iterations; ++i)
                    std::printf("iteration= %d\t chi2= %.6f\t edges= %d\t schur= 1\t lambda-equivalent radius= %.3e\n", i,
                                2.0 * trace[(size_t)i * STBA_TRACE_COLS], no, trace[(size_t)i * STBA_TRACE_COLS + 5]);
        } else {
            _message = std::string("stba_ba_solve: ") + stba_last_error();
        }
        stba_ba_destroy(ba);
        return done;
    }

private:
    // the user's oplus / computeError must be the built-in manifold update and reprojection residual
    bool probe_pose_oplus(Vertex* cam) {
        const double d[6] = {0.011, -0.017, 0.005, 0.03, -0.02, 0.04};
        double before[7], after[7];
        cam->get_raw(before);
        cam->push(); cam->oplus(d); cam->get_raw(after); cam->pop();
        const double th = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), im = std::sin(0.5 * th) / th, re = std::cos(0.5 * th);
        const double b[4] = {im * d[0], im * d[1], im * d[2], re}, *a = before;
        double q[4] = {a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1], a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0],
                       a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3], a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2]};
        double e1 = 0, e2 = 0;
        for (int i = 0; i < 4; ++i) { e1 = std::fmax(e1, std::fabs(after[i] - q[i])); e2 = std::fmax(e2, std::fabs(after[i] + q[i])); }
        double et = 0;
        for (int i = 0; i < 3; ++i) et = std::fmax(et, std::fabs(after[4 + i] - (before[4 + i] + d[3 + i])));
        if (std::fmin(e1, e2) > 1e-12 || et > 1e-12) { _message = "pose vertex oplusImpl is not (SO3 right-plus, translation add): unsupported on the device"; return false; }
        return true;
    }
    bool probe_landmark_oplus(Vertex* pt) {
        const double d[3] = {0.011, -0.017, 0.005};
        double pb[3], pa[3];
        pt->get_raw(pb); pt->push(); pt->oplus(d); pt->get_raw(pa); pt->pop();
        for (int i = 0; i < 3; ++i) if (std::fabs(pa[i] - (pb[i] + d[i])) > 1e-12) { _message = "landmark vertex oplusImpl is not additive"; return false; }
        return true;
    }
    // edge k: computeError == proj(R^T (L - t)) - z at the current estimates of its two vertices
    bool probe_edge(Edge* e, int k) {
        e->computeError();
        double err[2], z[2], c[7], L[3];
        e->error_raw(err); e->measurement_raw(z);
        e->vertices()[0]->get_raw(c); e->vertices()[1]->get_raw(L);
        const double x = c[0], y = c[1], zq = c[2], w = c[3];
        const double R[9] = {1 - 2 * (y * y + zq * zq), 2 * (x * y - w * zq), 2 * (x * zq + w * y), 2 * (x * y + w * zq), 1 - 2 * (x * x + zq * zq),
                             2 * (y * zq - w * x), 2 * (x * zq - w * y), 2 * (y * zq + w * x), 1 - 2 * (x * x + y * y)};
        const double dd[3] = {L[0] - c[4], L[1] - c[5], L[2] - c[6]};
        const double px = R[0] * dd[0] + R[3] * dd[1] + R[6] * dd[2], py = R[1] * dd[0] + R[4] * dd[1] + R[7] * dd[2],
                     pz = R[2] * dd[0] + R[5] * dd[1] + R[8] * dd[2];
        const double ex = px / pz - z[0], ey = py / pz - z[1];
        if (!(std::fabs(err[0] - ex) <= 1e-10 * (1.0 + std::fabs(ex))) || !(std::fabs(err[1] - ey) <= 1e-10 * (1.0 + std::fabs(ey)))) {
            _message = "edge " + std::to_string(k) + ": computeError is not the reprojection residual proj(R^T (L - t)) - z: unsupported on the device";
            return false;
        }
        return true;
    }
    std::vector<Vertex*> _vertices;
    std::vector<Edge*> _edges;
    OptimizationAlgorithm* _alg = nullptr;
    bool _verbose = false, _initialised = false;
    double _chi2 = 0.0;
    std::string _message;
};

}  // namespace stba_g2o
#endif

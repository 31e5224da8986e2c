// A C++ host of the sharded solver, as the reference's executables are (st20-g2o/src/src/test_ceres.cpp:7-19):
// it creates a native RCCL communicator through the C ABI (stba_comm_*), all-reduces a device buffer with it
// (ncclAllReduce, ncclDouble, ncclSum, on a HIP stream) and runs the bundle-adjustment engine with the
// communicator attached.  One rank here (one GPU per box); N ranks differ only in the id exchange.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <vector>

#include "stba.h"

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

static int solve(Scene& s, stba_comm* comm, double* final_cost, int* iters, std::vector<double>* cams_out) {
    std::vector<unsigned char> cf((size_t)s.nc * 6, 0);
    for (int c = 0; c < s.nc; ++c) if (s.fixed[c]) for (int a = 0; a < 6; ++a) cf[c * 6 + a] = 1;
    stba_ba* ba = nullptr;
    int rc = stba_ba_create(&ba, s.nc, s.np, s.no, s.cams.data(), s.pts.data(), s.oc.data(), s.op.data(), s.feat.data(), cf.data(), nullptr, nullptr);
    if (rc != STBA_OK) return rc;
    if (comm && (rc = stba_ba_set_comm(ba, comm)) != STBA_OK) return rc;
    stba_lm_options o; stba_lm_default_options(&o);
    stba_lm_summary sum;
    rc = stba_ba_solve(ba, &o, &sum, nullptr, nullptr, nullptr);
    if (rc == STBA_OK) {
        *final_cost = sum.final_cost; *iters = sum.num_iterations;
        cams_out->resize((size_t)s.nc * 7);
        rc = stba_ba_get_params(ba, cams_out->data(), nullptr);
    }
    stba_ba_destroy(ba);
    return rc;
}

int main(int argc, char** argv) {
    char id[STBA_COMM_ID_BYTES];
    int rc = stba_comm_unique_id(id);
    std::printf("unique_id rc %d %s\n", rc, rc ? stba_last_error() : "");
    if (rc != STBA_OK) return 0;                    // (no device: reported, the python test checks the code)
    stba_comm* comm = nullptr;
    rc = stba_comm_create(&comm, id, 0, 1, 0);
    std::printf("create rc %d %s\n", rc, rc ? stba_last_error() : "");
    if (rc != STBA_OK) return 1;
    int rank = -1, world = -1;
    stba_comm_rank(comm, &rank, &world);
    std::printf("rank %d world %d\n", rank, world);
    // ncclAllReduce on a device buffer, on an explicit stream
    const size_t n = 1 << 16;
    std::vector<double> h(n), back(n);
    for (size_t i = 0; i < n; ++i) h[i] = std::sin(0.001 * (double)i);
    double* d = nullptr; hipStream_t st;
    if (hipMalloc((void**)&d, n * sizeof(double)) != hipSuccess || hipStreamCreate(&st) != hipSuccess) return 1;
    hipMemcpyAsync(d, h.data(), n * sizeof(double), hipMemcpyHostToDevice, st);
    rc = stba_comm_allreduce_sum(comm, d, n, st);
    hipMemcpyAsync(back.data(), d, n * sizeof(double), hipMemcpyDeviceToHost, st);
    hipStreamSynchronize(st);
    double err = 0; for (size_t i = 0; i < n; ++i) err = std::fmax(err, std::fabs(back[i] - h[i]));
    std::printf("allreduce rc %d max_err %.3g\n", rc, err);
    hipFree(d); hipStreamDestroy(st);
    if (argc > 1) {
        Scene s;
        if (!s.load(argv[1])) { std::printf("scene_load_failed\n"); return 2; }
        double c0 = 0, c1 = 0; int i0 = 0, i1 = 0; std::vector<double> k0, k1;
        Scene a = s, b = s;
        const int r0 = solve(a, nullptr, &c0, &i0, &k0), r1 = solve(b, comm, &c1, &i1, &k1);
        double dc = 0; for (size_t i = 0; i < k0.size() && i < k1.size(); ++i) dc = std::fmax(dc, std::fabs(k0[i] - k1[i]));
        std::printf("solve rc %d %d iters %d %d cost %.17g %.17g cams_diff %.3g\n", r0, r1, i0, i1, c0, c1, dc);
    }
    rc = stba_comm_destroy(comm);
    std::printf("destroy rc %d\n", rc);
    return 0;
}

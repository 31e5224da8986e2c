// small_dense.hip -- the LM step of a SMALL dense problem (<= 32 local parameters) as ONE kernel launch per step, for the
// host-callback path of the operator API ("gpu-dense-callback": the reference's PnP call sites st17-ceres/src/include/solver.hpp:247-385,
// its per-landmark triangulation st20-g2o/src/src/sim_data.cpp:299-311, the bounds demo st17-ceres/src/ceres_bound.cpp:25-68).
//
// Why (round 6): these problems are six unknowns and forty residuals, and the reference PUBLISHES their wall time (0.12-0.22 ms per
// solve, st17-ceres/img/release.png).  Until round 5 every LM iteration of such a problem went through the machinery built for 6000
// unknowns -- two uploads, a memset, the normal-equation kernel, two downloads, a synchronise, a padded 128 x 128 system, the
// persistent factorisation program, two more downloads and synchronisations -- and every solve created a stream and a dozen
// allocations: 3.5 ms per solve.  Here:
//   * a pooled workspace (stream, device scratch, PINNED + MAPPED host buffers) that outlives the solve -- no allocation per solve;
//   * the user's Evaluate writes residuals and Jacobian straight into the mapped buffers; the kernel reads them over the bus once;
//   * one workgroup: J^T J and J^T r (fixed summation order), Jacobi scaling at the first linearisation, LM damping, Cholesky,
//     the two triangular solves and the model cost change -- then the step, the gradient and a sequence stamp into mapped host
//     memory; the host polls the stamp (no event, no copy, no synchronise).
// A rejected step launches the same kernel with `relinearize = 0`: H and g stay on the device.
#include "ba_kernels.hpp"

#include <atomic>
#include <chrono>
#include <mutex>
#include <vector>

namespace stba {

namespace {

constexpr int SD_THREADS = 256;
constexpr int SD_OUT_PAYLOAD_MAX = 2 * SMALL_DENSE_MAX_N + 2;     // dx | g | model cost change | pivot flag
constexpr int SD_STAGE_DOUBLES = 5632;     // J and r staged in LDS when n_res * (n + 1) fits (44 KB); device scratch otherwise

struct SmallStepArgs {
    const double* J;       // n_res x n row-major, mapped host memory
    const double* r;       // n_res, mapped host memory
    double* scratch;       // device: n_res * (n + 1) doubles (staging when LDS is too small)
    double* H;             // device: n x n (kept between launches)
    double* g;             // device: n
    double* scale;         // device: n (Jacobi scaling, fixed at the first linearisation)
    double* out;           // mapped host, a STAMPED BLOCK (common.hpp) of 2n + 2 payload doubles: [0, n) dx | [n, 2n) g | [2n] model cost change | [2n + 1] pivot flag
    int n_res, n, relinearize, first, jacobi;
    double radius, dmin, dmax, stamp;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(SD_THREADS) void dense_small_step_kernel(SmallStepArgs a) {
    __shared__ double stage[SD_STAGE_DOUBLES];
    __shared__ double Hs[SMALL_DENSE_MAX_N][SMALL_DENSE_MAX_N + 1];
    __shared__ double gs[SMALL_DENSE_MAX_N], dv[SMALL_DENSE_MAX_N], idiag[SMALL_DENSE_MAX_N];
    __shared__ int bad_pivot;
    __shared__ double pay[SD_OUT_PAYLOAD_MAX];
    const int t = threadIdx.x, n = a.n, nres = a.n_res;
    const int lane = t & 63, wv = t >> 6;
    if (t == 0) bad_pivot = 0;
    if (a.relinearize) {
        // ---- stage J | r (each element crosses the bus once, coalesced), then H = J^T J (lower) and g = J^T r: one entry per wave
        // and round, the rows strided over the lanes, a fixed shuffle tree -- the same bits from launch to launch
        const int total = nres * n;
        const bool in_lds = total + nres <= SD_STAGE_DOUBLES;
        double* Jl = in_lds ? stage : a.scratch;
        double* rl = Jl + total;
        for (int k = t; k < total; k += SD_THREADS) Jl[k] = a.J[k];
        for (int k = t; k < nres; k += SD_THREADS) rl[k] = a.r[k];
        __syncthreads();
        const int n_ent = n * (n + 1) / 2 + n;
        for (int e = wv; e < n_ent; e += SD_THREADS / 64) {
            int row, col;                               // e < n(n+1)/2: H(row, col), col <= row; behind: g(row)
            const int tri = n * (n + 1) / 2;
            if (e < tri) { row = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5); while ((row + 1) * (row + 2) / 2 <= e) ++row; while (row * (row + 1) / 2 > e) --row; col = e - row * (row + 1) / 2; }
            else { row = e - tri; col = -1; }
            double s = 0.0;
            for (int i = lane; i < nres; i += 64) s += Jl[i * n + row] * (col >= 0 ? Jl[i * n + col] : rl[i]);
            s = wave_sum(s);
            if (lane == 0) {
                if (col >= 0) { Hs[row][col] = s; Hs[col][row] = s; a.H[row * n + col] = s; a.H[col * n + row] = s; }
                else { gs[row] = s; a.g[row] = s; }
            }
        }
    } else {
        for (int k = t; k < n * n; k += SD_THREADS) Hs[k / n][k % n] = a.H[k];
        if (t < n) gs[t] = a.g[t];
    }
    __syncthreads();
    // ---- Jacobi scaling (first linearisation only), LM diagonal clamp(H_aa s^2, dmin, dmax) / radius / s^2 (Ceres'
    // LevenbergMarquardtStrategy on the scaled problem, written in unscaled coordinates as stba_dense_solve always did)
    if (t < n) {
        if (a.first) a.scale[t] = a.jacobi ? 1.0 / (1.0 + sqrt(Hs[t][t])) : 1.0;
        const double sc = a.scale[t], s2 = sc * sc;
        const double d = fmin(fmax(Hs[t][t] * s2, a.dmin), a.dmax) / a.radius / s2;
        dv[t] = d;
        Hs[t][t] += d;
    }
    __syncthreads();
    // ---- Cholesky, right-looking, row t owned by thread t
    for (int k = 0; k < n; ++k) {
        const double piv = Hs[k][k];
        if (!(piv > 0.0) || !(piv < 1e300)) { if (t == 0 && bad_pivot == 0) bad_pivot = k + 1; break; }    // (uniform: every thread reads the same value)
        const double d = sqrt(piv);
        double lik = 0.0;
        if (t > k && t < n) lik = Hs[t][k] / d;
        __syncthreads();
        if (t == k) { Hs[k][k] = d; idiag[k] = 1.0 / d; }
        if (t > k && t < n) Hs[t][k] = lik;
        __syncthreads();
        if (t > k && t < n)
            for (int j = k + 1; j <= t; ++j) Hs[t][j] -= lik * Hs[j][k];
        __syncthreads();
    }
    __syncthreads();
    // ---- L y = -g, L^T x = y: column-oriented in wave 0, the running right-hand side in a register per lane
    if (wv == 0) {
        const bool failed = bad_pivot != 0;
        double b = (lane < n && !failed) ? -gs[lane] : 0.0;
        if (!failed) {
            for (int k = 0; k < n; ++k) {
                const double yk = __shfl(b, k, 64) * idiag[k];
                if (lane == k) b = yk;
                else if (lane > k && lane < n) b -= Hs[lane][k] * yk;
            }
            for (int k = n - 1; k >= 0; --k) {
                const double xk = __shfl(b, k, 64) * idiag[k];
                if (lane == k) b = xk;
                else if (lane < k) b -= Hs[k][lane] * xk;
            }
        }
        double m = (lane < n && !failed) ? (-0.5 * gs[lane] * b + 0.5 * dv[lane] * b * b) : 0.0;
        m = wave_sum(m);
        if (lane < n) { pay[lane] = b; pay[n + lane] = gs[lane]; }
        if (lane == 0) { pay[2 * n] = m; pay[2 * n + 1] = (double)bad_pivot; }
    }
    __syncthreads();
    // the step, the gradient, the model change and the flag as ONE stamped block: every 64-byte line carries the stamp and a check
    // word of its own (common.hpp: a stamp behind the payload was seen by the host BEFORE payload in another line, 1 in ~50 000 steps)
    if (wv == 0) stamped_store_wave(a.out, pay, 2 * n + 2, a.stamp, lane);
}

std::mutex g_pool_mutex;
std::vector<SmallDenseWs*> g_pool;

double wall_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

struct SmallDenseWs {
    int device = -1;
    hipStream_t st = nullptr;
    double *hJ = nullptr, *hr = nullptr, *hout = nullptr;      // pinned, mapped
    double *dJ = nullptr, *dr = nullptr, *dout = nullptr;      // their device addresses
    double *scratch = nullptr, *H = nullptr, *g = nullptr, *scale = nullptr;
    size_t cap_j = 0, cap_r = 0;
    double stamp = 0.0;
    double result[2 * SMALL_DENSE_MAX_N + 2];      // the host's validated copy of the last step's block
};

bool small_dense_fits(int n_res, int n) {
    return n >= 1 && n <= SMALL_DENSE_MAX_N && n_res >= 1 && (size_t)n_res * (size_t)(n + 1) <= ((size_t)1 << 16);
}

static int ws_grow(SmallDenseWs* w, size_t need_j, size_t need_r) {
    if (need_j > w->cap_j || need_r > w->cap_r) {
        if (w->hJ) (void)hipHostFree(w->hJ);
        if (w->hr) (void)hipHostFree(w->hr);
        if (w->scratch) (void)hipFree(w->scratch);
        w->hJ = w->hr = w->scratch = nullptr; w->cap_j = w->cap_r = 0;
        const size_t cj = std::max<size_t>(need_j, 4096), cr = std::max<size_t>(need_r, 1024);
        STBA_HIP(hipHostMalloc(reinterpret_cast<void**>(&w->hJ), cj * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
        STBA_HIP(hipHostMalloc(reinterpret_cast<void**>(&w->hr), cr * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
        STBA_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&w->dJ), w->hJ, 0));
        STBA_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&w->dr), w->hr, 0));
        STBA_HIP(hipMalloc(reinterpret_cast<void**>(&w->scratch), (cj + cr) * sizeof(double)));
        w->cap_j = cj; w->cap_r = cr;
    }
    return STBA_OK;
}

int small_dense_acquire(SmallDenseWs** out, int n_res, int n) {
    *out = nullptr;
    int dev = 0;
    STBA_HIP(hipGetDevice(&dev));
    SmallDenseWs* w = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        for (size_t k = 0; k < g_pool.size(); ++k)
            if (g_pool[k]->device == dev) { w = g_pool[k]; g_pool.erase(g_pool.begin() + (long)k); break; }
    }
    if (!w) {
        w = new SmallDenseWs();
        w->device = dev;
        auto fail_new = [&](int rc) { delete w; return rc; };       // (a half-made workspace is dropped; its few buffers leak with the failed device)
        if (hipStreamCreateWithFlags(&w->st, hipStreamNonBlocking) != hipSuccess) return fail_new(fail(STBA_ERR_HIP, "small dense workspace: hipStreamCreate"));
        const size_t nn = SMALL_DENSE_MAX_N;
        if (hipHostMalloc(reinterpret_cast<void**>(&w->hout), (size_t)stamped_doubles(SD_OUT_PAYLOAD_MAX) * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipHostGetDevicePointer(reinterpret_cast<void**>(&w->dout), w->hout, 0) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&w->H), nn * nn * sizeof(double)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&w->g), nn * sizeof(double)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&w->scale), nn * sizeof(double)) != hipSuccess)
            return fail_new(fail(STBA_ERR_ALLOC, "small dense workspace: allocation failed"));
        memset(w->hout, 0, (size_t)stamped_doubles(SD_OUT_PAYLOAD_MAX) * sizeof(double));
    }
    const int rc = ws_grow(w, (size_t)n_res * (size_t)n, (size_t)n_res);
    if (rc != STBA_OK) { small_dense_release(w); return rc; }
    *out = w;
    return STBA_OK;
}

void small_dense_release(SmallDenseWs* w) {
    if (!w) return;
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    g_pool.push_back(w);          // kept for the next solve of this process (a few MB; never freed: the HIP runtime may be gone at exit)
}

double* small_dense_J(SmallDenseWs* w) { return w->hJ; }
double* small_dense_r(SmallDenseWs* w) { return w->hr; }

int small_dense_step(SmallDenseWs* w, int n_res, int n, bool relinearize, bool first, bool jacobi, double radius, double dmin,
                     double dmax, const double** dx, const double** g, double* model_change, int* pivot_flag) {
    SmallStepArgs a;
    a.J = w->dJ; a.r = w->dr; a.scratch = w->scratch; a.H = w->H; a.g = w->g; a.scale = w->scale; a.out = w->dout;
    a.n_res = n_res; a.n = n; a.relinearize = relinearize ? 1 : 0; a.first = first ? 1 : 0; a.jacobi = jacobi ? 1 : 0;
    a.radius = radius; a.dmin = dmin; a.dmax = dmax;
    w->stamp += 1.0;
    a.stamp = w->stamp;
    std::atomic_thread_fence(std::memory_order_release);          // the callback's stores into the mapped buffers come first
    hipLaunchKernelGGL(dense_small_step_kernel, dim3(1), dim3(SD_THREADS), 0, w->st, a);
    STBA_HIP(hipGetLastError());
    volatile double* h = w->hout;
    const double t0 = wall_now();
    const double want = a.stamp;
    auto is_mine = [want](double st) { return st == want; };
    for (unsigned long spin = 1; !stamped_try_read(h, 2 * n + 2, is_mine, w->result); ++spin) {
        if ((spin & 0xfff) == 0) {
            const hipError_t q = hipStreamQuery(w->st);
            if (q != hipSuccess && q != hipErrorNotReady) return fail(STBA_ERR_HIP, std::string("small dense step: ") + hipGetErrorString(q));
            if (q == hipSuccess) {
                // the kernel is done: one synchronise settles what the host may read
                STBA_HIP(hipStreamSynchronize(w->st));
                if (!stamped_try_read(h, 2 * n + 2, is_mine, w->result)) return fail(STBA_ERR_HIP, "small dense step: the result never arrived in mapped host memory");
                break;
            }
            if (wall_now() - t0 > 60.0) return fail(STBA_ERR_HIP, "small dense step: timed out");
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    *dx = w->result; *g = w->result + n;
    *model_change = w->result[2 * n];
    *pivot_flag = (int)w->result[2 * n + 1];
    return STBA_OK;
}

}  // namespace stba

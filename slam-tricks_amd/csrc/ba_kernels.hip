// ba_kernels.hip -- HIP kernels of the bundle-adjustment hot path for gfx950 (MI355X), FP64.
//
// Reference semantics (paths relative to /root/reference):
//   residual            st20-g2o/src/include/test_ceres.h:63-80   r = proj(R^T (L - t)) - feature
//   Jacobian            st17-ceres/src/include/solver.hpp:183-209 with the hat(pInC) rotation block
//                       (SURVEY.md header fact 2; == autodiff composed with solver.hpp:48-54)
//   J^T J / J^T r       solver.hpp:402-436; block structure st20-g2o/src/include/sim_data.h:108-159
//   Schur elimination   SPARSE_SCHUR, test_ceres.h:145 / setMarginalized, test_g2o.h:121
//   manifold update     solver.hpp:38-45, test_g2o.h:36-39,60-63
//
// Data layout in HBM (all device resident for the whole solve):
//   cams   [n_cams][7]   qx qy qz qw tx ty tz          (56 B/camera; staged into LDS per workgroup)
//   pts    [n_pts][3]
//   obs    landmark-major: feat double2[n_obs], obs_cam int[n_obs], obs_pt int[n_obs]
//   r      double2[n_obs];  J8 [n_obs][8] = {xn, yn, P (2x3 row-major)} with P = A R^T, A = d proj / d pInC: the
//          COMPACT Jacobian.  The 2x6 camera block is [A hat(pInC) | -P] and its first part depends on (xn, yn)
//          only, the 2x3 landmark block is P: 64 B per observation instead of 144 B, expanded in registers where it
//          is used (expand_j8); constant dofs / landmarks are a per-observation mask byte (omask), not stored zeros
//   Hpp6   [n_pts][6] (xx xy xz yy yz zz), gp [n_pts][3], Hinv6 [n_pts][6]
//   Hcc    [n_cams][36], gc [n_cams][6]
//   S      [lda][lda] dense reduced camera system (lower triangle), lda = padded 6*n_cams
#include <algorithm>

#include "ba_kernels.hpp"

namespace stba {

// ===========================================================================================
// residual + Jacobian, one observation per lane.
// Algorithmic HBM traffic per observation: 24 B (feature + 2 indices) + landmark 24 B / (obs per
// landmark) read; 16 B (r) + 64 B (compact Jacobian) written = 106.5 B at 10 obs/landmark (the
// materialised 2x6 | 2x3 form of SURVEY.md 8d would be 186.5 B: 80 B of it are redundant).  Camera
// blocks are staged once per workgroup in LDS; the per-lane 64 B Jacobian rows are transposed through LDS
// so that every global store instruction writes 16 B x 64 contiguous lanes.
// ===========================================================================================
template <bool CAMS_IN_LDS, bool WITH_JAC>
__global__ __launch_bounds__(LIN_THREADS) void ba_linearize_kernel(LinArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* s_j8 = smem;                                   // [LIN_THREADS][9]
    double* s_cam = smem + (WITH_JAC ? LIN_THREADS * 9 : 0);   // [n_cams][7]
    const int tid = threadIdx.x;
    if (CAMS_IN_LDS) {
        for (int i = tid; i < a.n_cams * 7; i += LIN_THREADS) s_cam[i] = a.cams[i];
        __syncthreads();
    }
    double cost = 0.0;
    const int n_tiles = (a.n_obs + LIN_THREADS - 1) / LIN_THREADS;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int base = tile * LIN_THREADS;
        const int i = base + tid;
        if (i < a.n_obs) {
            const double2 f = a.feat[i];
            const int c = a.obs_cam[i], j = a.obs_pt[i];
            const double* cam = CAMS_IN_LDS ? (s_cam + c * 7) : (a.cams + (size_t)c * 7);
            double q[4] = {cam[0], cam[1], cam[2], cam[3]};
            const double t0 = cam[4], t1 = cam[5], t2 = cam[6];
            const double* L = a.pts + (size_t)j * 3;
            const double d0 = L[0] - t0, d1 = L[1] - t1, d2 = L[2] - t2;
            double R[9];
            quat_to_rot(q, R);
            const double x = R[0] * d0 + R[3] * d1 + R[6] * d2;   // R^T (L - t)
            const double y = R[1] * d0 + R[4] * d1 + R[7] * d2;
            const double z = R[2] * d0 + R[5] * d1 + R[8] * d2;
            const double iz = 1.0 / z;
            const double xn = x * iz, yn = y * iz;
            const double r0 = xn - f.x, r1 = yn - f.y;
            if (a.r) a.r[i] = make_double2(r0, r1);
            cost += r0 * r0 + r1 * r1;
            if (WITH_JAC) {
                // A = [[iz,0,-xn iz],[0,iz,-yn iz]];  P = A R^T;  (A hat(pInC) is a function of xn, yn: expand_j8)
                double* j8 = s_j8 + tid * 9;
                j8[0] = xn; j8[1] = yn;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    // (A R^T)_{0k} = iz*R[k][0] - xn*iz*R[k][2];  row 1 with R[k][1], yn
                    j8[2 + k] = iz * (R[k * 3 + 0] - xn * R[k * 3 + 2]);
                    j8[5 + k] = iz * (R[k * 3 + 1] - yn * R[k * 3 + 2]);
                }
            }
        }
        if (WITH_JAC) {
            __syncthreads();
            const int n_here = min(LIN_THREADS, a.n_obs - base);
            // n_here*8 doubles, written as double2 (16 B per lane, contiguous across lanes)
            double* gj = a.J8 + (size_t)base * 8;
            for (int e = tid; e < n_here * 4; e += LIN_THREADS) {
                const int o = e >> 2, k = (e & 3) * 2;
                const double* sj = s_j8 + o * 9 + k;
                *reinterpret_cast<double2*>(gj + 2 * (size_t)e) = make_double2(sj[0], sj[1]);
            }
            __syncthreads();
        }
    }
    // deterministic block reduction of the cost
    __shared__ double s_red[LIN_THREADS / 64];
    for (int off = 32; off > 0; off >>= 1) cost += __shfl_down(cost, off, 64);
    if ((tid & 63) == 0) s_red[tid >> 6] = cost;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < LIN_THREADS / 64; ++w) s += s_red[w];
        a.cost_partial[blockIdx.x] = s;
    }
}

size_t lin_lds_bytes(int n_cams, bool cams_in_lds, bool with_jac) {
    size_t d = 0;
    if (with_jac) d += (size_t)LIN_THREADS * 9;
    if (cams_in_lds) d += (size_t)n_cams * 7;
    return d * sizeof(double) + 16;
}

int launch_linearize(const LinArgs& a, bool with_jac, int grid, hipStream_t st) {
    const bool in_lds = lin_lds_bytes(a.n_cams, true, with_jac) <= LIN_MAX_LDS;
    const size_t lds = lin_lds_bytes(a.n_cams, in_lds, with_jac);
    static DeviceOnce attr;
    STBA_TRY(attr.run([]() -> int {
        STBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ba_linearize_kernel<true, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LIN_MAX_LDS));
        STBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ba_linearize_kernel<true, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LIN_MAX_LDS));
        STBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ba_linearize_kernel<false, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LIN_MAX_LDS));
        return STBA_OK;
    }));
    if (in_lds && with_jac)
        hipLaunchKernelGGL((ba_linearize_kernel<true, true>), dim3(grid), dim3(LIN_THREADS), lds, st, a);
    else if (in_lds)
        hipLaunchKernelGGL((ba_linearize_kernel<true, false>), dim3(grid), dim3(LIN_THREADS), lds, st, a);
    else if (with_jac)
        hipLaunchKernelGGL((ba_linearize_kernel<false, true>), dim3(grid), dim3(LIN_THREADS), lds, st, a);
    else
        hipLaunchKernelGGL((ba_linearize_kernel<false, false>), dim3(grid), dim3(LIN_THREADS), lds, st, a);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

// ===========================================================================================
// scalar reductions (fixed order => deterministic)
// ===========================================================================================
// out[k] = sum_i partial[i*stride + k], k < K ; one workgroup
__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ partial, int n,
                                                           int stride, int K, double* __restrict__ out) {
    __shared__ double s[256];
    for (int k = 0; k < K; ++k) {
        double v = 0.0;
        for (int i = threadIdx.x; i < n; i += 256) v += partial[(size_t)i * stride + k];
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[k] = s[0];
        __syncthreads();
    }
}

int launch_sum_partials(const double* partial, int n, int stride, int K, double* out, hipStream_t st) {
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, partial, n, stride, K, out);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

// max |v[i]| over n (+ n2) entries -> out[0]: grid-stride partial maxima, last block finishes
// (max is order-independent, so this stays deterministic)
__global__ __launch_bounds__(256) void absmax_partial_kernel(const double* __restrict__ v, size_t n,
                                                             const double* __restrict__ v2, size_t n2,
                                                             double* __restrict__ partial) {
    __shared__ double s[256];
    double m = 0.0;
    const size_t stride = (size_t)gridDim.x * 256, i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (size_t i = i0; i < n; i += stride) m = fmax(m, fabs(v[i]));
    for (size_t i = i0; i < n2; i += stride) m = fmax(m, fabs(v2[i]));
    s[threadIdx.x] = m;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) s[threadIdx.x] = fmax(s[threadIdx.x], s[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = s[0];
}

__global__ __launch_bounds__(256) void absmax_final_kernel(const double* __restrict__ partial, int n,
                                                           double* __restrict__ out) {
    __shared__ double s[256];
    double m = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) m = fmax(m, partial[i]);
    s[threadIdx.x] = m;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) s[threadIdx.x] = fmax(s[threadIdx.x], s[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = s[0];
}

int launch_absmax(const double* v, size_t n, const double* v2, size_t n2, double* out, double* partial, int n_partial,
                  hipStream_t st) {
    const size_t total = n + n2;
    int grid = (int)std::min<size_t>((size_t)n_partial, std::max<size_t>(1, (total + 2047) / 2048));
    hipLaunchKernelGGL(absmax_partial_kernel, dim3(grid), dim3(256), 0, st, v, n, v2, n2, partial);
    hipLaunchKernelGGL(absmax_final_kernel, dim3(1), dim3(256), 0, st, partial, grid, out);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

// The whole trial block of one LM iteration in ONE launch behind the residual-only kernel: out[0] = sum of that kernel's
// cost partials (summation order of sum_partials_kernel), out[1..3] / out[5..7] the three sums of the landmark / camera
// update's partials (order of trial_sums_kernel), out[4] = 1 if the factorisation timed out on this rank; and, if host_out is given, the block plus the
// factorisation's flag written straight into mapped host memory (export_trial_kernel): three launches less.
__global__ __launch_bounds__(256) void trial_finish_kernel(const double* __restrict__ cost_partial, int n_cost,
                                                           const double* __restrict__ part_p, int n_p, const double* __restrict__ part_c,
                                                           int n_c, const int* __restrict__ flag, double* __restrict__ out,
                                                           double* __restrict__ host_out, double host_seq) {
    // the seven sums side by side: per thread a strided share of each, then ONE tree for all of them (every sum in the order it
    // always had: strided shares, then halving)
    __shared__ double s[7][256];
    __shared__ double res[8];
    {
        double v[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        // (unrolled: the loads of several trips are in flight together, the additions keep their order)
#pragma unroll 4
        for (int i = threadIdx.x; i < n_cost; i += 256) v[0] += cost_partial[i];
#pragma unroll 4
        for (int i = threadIdx.x; i < n_p; i += 256) {
#pragma unroll
            for (int k = 0; k < 3; ++k) v[1 + k] += part_p[(size_t)i * 4 + k];
        }
        for (int i = threadIdx.x; i < n_c; i += 256) {
#pragma unroll
            for (int k = 0; k < 3; ++k) v[4 + k] += part_c[(size_t)i * 4 + k];
        }
#pragma unroll
        for (int q = 0; q < 7; ++q) s[q][threadIdx.x] = v[q];
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (threadIdx.x < off) {
#pragma unroll
                for (int q = 0; q < 7; ++q) s[q][threadIdx.x] += s[q][threadIdx.x + off];
            }
            __syncthreads();
        }
        if (threadIdx.x < 7) res[threadIdx.x] = s[threadIdx.x][0];
        __syncthreads();
    }
    __shared__ double hp[9];
    if (threadIdx.x < 8) {
        // block layout (stba_engine.hip, TS_*): [cost2, step2, x2, model | timed out | the three camera sums]; entry 4 = 1.0 if the
        // factorisation of this iteration timed out on this rank (CHOL_FLAG_TIMEOUT): it lies inside the prefix the ranks sum
        const int k = threadIdx.x;
        const double v = k < 4 ? res[k] : k == 4 ? ((flag[0] == CHOL_FLAG_TIMEOUT) ? 1.0 : 0.0) : res[k - 1];
        out[k] = v;
        hp[k] = v;
    } else if (threadIdx.x == 8) hp[8] = (double)flag[0];
    if (host_out) {
        // the host polls the block (no event on the stream: an event record is ~5 us of idle GPU): the eight sums and the
        // factorisation's flag as a STAMPED BLOCK, every 64-byte line with the sequence number and a check word of its own (common.hpp)
        __syncthreads();
        if (threadIdx.x < 64) stamped_store_wave(host_out, hp, 9, host_seq, threadIdx.x);
    }
}

int launch_trial_finish(const double* cost_partial, int n_cost, const double* part_p, int n_p, const double* part_c, int n_c,
                        const int* flag, double* out, double* host_out, double host_seq, hipStream_t st) {
    hipLaunchKernelGGL(trial_finish_kernel, dim3(1), dim3(256), 0, st, cost_partial, n_cost, part_p, n_p, part_c, n_c, flag, out, host_out,
                       host_seq);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

// compact Jacobian of observation i -> the 2x6 camera block (d/dtheta | d/dt) and the 2x3 landmark block; columns of
// constant dofs (mask bits 0..5) and of constant landmarks (bit 6) are zero, as the stored form used to have them
// GEN (host-linearised factors, stba_ba_set_host_linearizer): the landmark block is still the record's P (so the landmark kernels
// need no second form), the camera block is NOT a function of it and comes from its own array Jc12 [n_obs][12] (2 x 6 row-major).
template <bool GEN = false>
__device__ inline void load_jc_jp(const double* __restrict__ J8, const unsigned char* __restrict__ omask, int i,
                                  double jc[12], double jp[6], const double* __restrict__ Jc12 = nullptr) {
    const double2* pj = reinterpret_cast<const double2*>(J8 + (size_t)i * 8);
    const double2 v0 = pj[0], v1 = pj[1], v2 = pj[2], v3 = pj[3];
    const unsigned m = omask ? omask[i] : 0u;
    if constexpr (GEN) {
        const double2* pc = reinterpret_cast<const double2*>(Jc12 + (size_t)i * 12);
        const double P[6] = {v1.x, v1.y, v2.x, v2.y, v3.x, v3.y};
        const bool pf = (m & 64u) != 0u;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double2 v = pc[k];
            jc[2 * k] = ((m >> ((2 * k) % 6)) & 1u) ? 0.0 : v.x;
            jc[2 * k + 1] = ((m >> ((2 * k + 1) % 6)) & 1u) ? 0.0 : v.y;
            jp[k] = pf ? 0.0 : P[k];
        }
        return;
    }
    const double xn = v0.x, yn = v0.y;
    const double P[6] = {v1.x, v1.y, v2.x, v2.y, v3.x, v3.y};
    jc[0] = (m & 1u) ? 0.0 : xn * yn;
    jc[1] = (m & 2u) ? 0.0 : -(1.0 + xn * xn);
    jc[2] = (m & 4u) ? 0.0 : yn;
    jc[6] = (m & 1u) ? 0.0 : 1.0 + yn * yn;
    jc[7] = (m & 2u) ? 0.0 : -xn * yn;
    jc[8] = (m & 4u) ? 0.0 : -xn;
    const bool pf = (m & 64u) != 0u;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const bool fx = (m >> (3 + k)) & 1u;
        jc[3 + k] = fx ? 0.0 : -P[k];
        jc[9 + k] = fx ? 0.0 : -P[3 + k];
        jp[k] = pf ? 0.0 : P[k];
        jp[3 + k] = pf ? 0.0 : P[3 + k];
    }
}

// stage entry points (stba_ba_evaluate) hand out the expanded form
__global__ __launch_bounds__(256) void ba_expand_jacobian_kernel(int n_obs, const double* __restrict__ J8,
                                                                 const unsigned char* __restrict__ omask,
                                                                 double* __restrict__ Jc, double* __restrict__ Jp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_obs) return;
    double jc[12], jp[6];
    load_jc_jp(J8, omask, i, jc, jp);
    if (Jc) for (int k = 0; k < 12; ++k) Jc[(size_t)i * 12 + k] = jc[k];
    if (Jp) for (int k = 0; k < 6; ++k) Jp[(size_t)i * 6 + k] = jp[k];
}
int launch_expand_jacobian(int n_obs, const double* J8, const unsigned char* omask, double* Jc, double* Jp, hipStream_t st) {
    if (n_obs > 0) hipLaunchKernelGGL(ba_expand_jacobian_kernel, dim3((n_obs + 255) / 256), dim3(256), 0, st, n_obs, J8, omask, Jc, Jp);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

// ===========================================================================================
// point blocks: Hpp_j = sum Jp^T Jp, gp_j = sum Jp^T r over the landmark's observation segment
// ===========================================================================================
// A workgroup owns PB_LM consecutive landmarks, i.e. ONE contiguous stretch of observation records (the observations
// are sorted by landmark).  Pass 1, one record per lane and trip: full-line loads of the records, the nine products
// of a record into LDS (component-major).  Pass 2, one OUTPUT entry per lane: the entry's landmark's records are summed
// in observation order (bitwise independent of the launch geometry) and stored -- consecutive lanes, consecutive
// addresses.  (Before: one landmark per lane walking its ~10 records, every lane of a wave in a different line: 25 us
// for 80 MB.)
// (records per pass = PB_TRIPS x PB_THREADS: 2 x 128 measured against 3 x 128 -- 18.5 KB of LDS instead of 27.7, eight
// workgroups per CU instead of five: landmark blocks 19.4 -> 17.3 us; 24 landmarks per workgroup: the same, more partials)
constexpr int PB_LM = 32, PB_THREADS = 128, PB_TRIPS = 2, PB_CHUNK = PB_TRIPS * PB_THREADS, PB_LD = PB_CHUNK + 1;
__global__ __launch_bounds__(PB_THREADS) void ba_point_blocks_kernel(int n_pts, const int* __restrict__ pt_start,
                                                                     const double* __restrict__ J8,
                                                                     const unsigned char* __restrict__ omask,
                                                                     const double2* __restrict__ r,
                                                                     double* __restrict__ Hpp6, double* __restrict__ gp,
                                                                     double* __restrict__ gpmax_partial) {
    __shared__ double u[9 * PB_LD];
    __shared__ int seg[PB_LM + 1];
    __shared__ double smax[PB_THREADS / 64];
    const int t = threadIdx.x;
    const int j0 = blockIdx.x * PB_LM, nl = min(PB_LM, n_pts - j0);
    if (t <= nl) seg[t] = pt_start[j0 + t];
    __syncthreads();
    const int rb = seg[0], re = seg[nl];
    // this lane's output entries: o = t and t + 128 of the 6 nl entries of Hpp6, o = t of the 3 nl entries of gp
    double acc[3] = {0.0, 0.0, 0.0};
    int lj[3], lk[3];
    lj[0] = t / 6; lk[0] = t % 6;
    lj[1] = (t + PB_THREADS) / 6; lk[1] = (t + PB_THREADS) % 6;
    lj[2] = t / 3; lk[2] = 6 + t % 3;
    if (t >= 3 * PB_LM) lj[2] = nl;
#pragma unroll
    for (int q = 0; q < 3; ++q) if (lj[q] > nl) lj[q] = nl;        // (seg[nl] .. seg[nl]: nothing to sum)
    for (int cb = rb; cb < re; cb += PB_CHUNK) {
        const int ce = min(cb + PB_CHUNK, re);
#pragma unroll
        for (int s2 = 0; s2 < PB_TRIPS; ++s2) {
            const int x = t + PB_THREADS * s2, i = cb + x;
            if (i < ce) {
                const double2* p = reinterpret_cast<const double2*>(J8 + (size_t)i * 8);
                const double2 a = p[1], b = p[2], c = p[3];   // a.x a.y b.x | b.y c.x c.y
                const double2 ri = r[i];
                const bool off = omask && (omask[i] & 64u);          // constant landmark: zero block
                const double j00 = off ? 0.0 : a.x, j01 = off ? 0.0 : a.y, j02 = off ? 0.0 : b.x;
                const double j10 = off ? 0.0 : b.y, j11 = off ? 0.0 : c.x, j12 = off ? 0.0 : c.y;
                u[0 * PB_LD + x] = j00 * j00 + j10 * j10; u[1 * PB_LD + x] = j00 * j01 + j10 * j11; u[2 * PB_LD + x] = j00 * j02 + j10 * j12;
                u[3 * PB_LD + x] = j01 * j01 + j11 * j11; u[4 * PB_LD + x] = j01 * j02 + j11 * j12; u[5 * PB_LD + x] = j02 * j02 + j12 * j12;
                u[6 * PB_LD + x] = j00 * ri.x + j10 * ri.y; u[7 * PB_LD + x] = j01 * ri.x + j11 * ri.y; u[8 * PB_LD + x] = j02 * ri.x + j12 * ri.y;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (lj[q] < nl) {
                const int b0 = max(seg[lj[q]], cb), e0 = min(seg[lj[q] + 1], ce);
                const double* uk = u + lk[q] * PB_LD - cb;
                for (int i = b0; i < e0; ++i) acc[q] += uk[i];
            }
        }
        __syncthreads();
    }
    if (lj[0] < nl) Hpp6[(size_t)j0 * 6 + t] = acc[0];
    if (lj[1] < nl) Hpp6[(size_t)j0 * 6 + t + PB_THREADS] = acc[1];
    if (lj[2] < nl) gp[(size_t)j0 * 3 + t] = acc[2];
    if (gpmax_partial) {
        // max |gp| of this workgroup's landmarks (the gradient max-norm of the LM loop: one partial per workgroup, finished
        // by linear_finish_kernel; a maximum does not depend on the order)
        double m = (lj[2] < nl) ? fabs(acc[2]) : 0.0;
        for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, 64));
        if ((t & 63) == 0) smax[t >> 6] = m;
        __syncthreads();
        if (t == 0) {
            double mm = 0.0;
            for (int w = 0; w < PB_THREADS / 64; ++w) mm = fmax(mm, smax[w]);
            gpmax_partial[blockIdx.x] = mm;
        }
    }
}

int point_blocks_grid(int n_pts) { return (n_pts + PB_LM - 1) / PB_LM; }

int launch_point_blocks(int n_pts, const int* pt_start, const double* J8, const unsigned char* omask, const double2* r,
                        double* Hpp6, double* gp, double* gpmax_partial, hipStream_t st) {
    if (n_pts > 0)
        hipLaunchKernelGGL(ba_point_blocks_kernel, dim3(point_blocks_grid(n_pts)), dim3(PB_THREADS), 0, st, n_pts, pt_start,
                           J8, omask, r, Hpp6, gp, gpmax_partial);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

// Behind a linearisation (residual + Jacobian kernel, landmark blocks): the cost partial sums of the linearise kernel's
// workgroups -> cost2_out[0] (same summation order as sum_partials_kernel), the |gp| partial maxima -> this rank's slot,
// and the scalar block behind S filled: slots[cost_slot] = cost2, slots[max_slot] = |gp|max, every other slot zero.
// One launch instead of three (sum_partials, absmax_partial + scalar_slots_final).
__global__ __launch_bounds__(256) void linear_finish_kernel(const double* __restrict__ cost_partial, int n_cost,
                                                            const double* __restrict__ gpmax_partial, int n_gp,
                                                            double* __restrict__ cost2_out, double* __restrict__ slots, int n_slots,
                                                            int cost_slot, int max_slot) {
    __shared__ double s[256], smx[256];
    double v = 0.0, m = 0.0;
#pragma unroll 4
    for (int i = threadIdx.x; i < n_cost; i += 256) v += cost_partial[i];
#pragma unroll 4
    for (int i = threadIdx.x; i < n_gp; i += 256) m = fmax(m, gpmax_partial[i]);
    s[threadIdx.x] = v;
    smx[threadIdx.x] = m;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {          // sum and maximum in one tree
        if (threadIdx.x < off) {
            s[threadIdx.x] += s[threadIdx.x + off];
            smx[threadIdx.x] = fmax(smx[threadIdx.x], smx[threadIdx.x + off]);
        }
        __syncthreads();
    }
    const double cost2 = s[0];
    if (threadIdx.x == 0) cost2_out[0] = cost2;
    for (int i = threadIdx.x; i < n_slots; i += 256) slots[i] = (i == cost_slot) ? cost2 : (i == max_slot) ? smx[0] : 0.0;
}

int launch_linear_finish(const double* cost_partial, int n_cost, const double* gpmax_partial, int n_gp, double* cost2_out,
                         double* slots, int n_slots, int cost_slot, int max_slot, hipStream_t st) {
    hipLaunchKernelGGL(linear_finish_kernel, dim3(1), dim3(256), 0, st, cost_partial, n_cost, gpmax_partial, n_gp, cost2_out, slots,
                       n_slots, cost_slot, max_slot);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

// ===========================================================================================
// camera blocks: wave-segmented reduction over the camera-sorted permutation.
// One wave per chunk (<= CAM_CHUNK observations of ONE camera): each lane accumulates the 21
// unique entries of Jc^T Jc and the 6 of Jc^T r over its strided share, a wave shuffle tree
// reduces them, lane 0 writes the chunk partial; a second kernel adds the chunks of a camera
// in order (deterministic, no atomics).
// ===========================================================================================
// (GEN: host-linearised factors -- the camera block comes from its own array Jc12, not from the closed form behind J8)
template <bool GEN>
__global__ __launch_bounds__(256) void ba_camera_partial_kernel(int n_chunks, const int* __restrict__ chunk_begin,
                                                                const int* __restrict__ chunk_end,
                                                                const int* __restrict__ cam_perm,
                                                                const double* __restrict__ J8,
                                                                const unsigned char* __restrict__ omask,
                                                                const double* __restrict__ Jc12,
                                                                const double2* __restrict__ r,
                                                                double* __restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ch >= n_chunks) return;
    double acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0.0;
    const int e = chunk_end[ch];
    for (int p = chunk_begin[ch] + lane; p < e; p += 64) {
        const int i = cam_perm[p];
        double j[12], jpu[6];
        load_jc_jp<GEN>(J8, omask, i, j, jpu, Jc12);
        const double2 ri = r[i];
        int idx = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) acc[idx++] += j[a] * j[b] + j[6 + a] * j[6 + b];
#pragma unroll
        for (int a = 0; a < 6; ++a) acc[21 + a] += j[a] * ri.x + j[6 + a] * ri.y;
    }
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        double v = acc[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        acc[k] = v;
    }
    if (lane == 0) {
        double* o = partial + (size_t)ch * 28;
#pragma unroll
        for (int k = 0; k < 27; ++k) o[k] = acc[k];
    }
}

__global__ __launch_bounds__(256) void ba_camera_final_kernel(int n_cams, const int* __restrict__ cam_chunk_start,
                                                              const double* __restrict__ partial,
                                                              double* __restrict__ Hcc, double* __restrict__ gc) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int c = gid / 27, k = gid - c * 27;
    if (c >= n_cams) return;
    double s = 0.0;
    const int e = cam_chunk_start[c + 1];
    for (int ch = cam_chunk_start[c]; ch < e; ++ch) s += partial[(size_t)ch * 28 + k];
    if (k < 21) {
        int a = 0, b = k;
        while (b > a) { ++a; b -= a; }   // k = a(a+1)/2 + b
        Hcc[(size_t)c * 36 + a * 6 + b] = s;
        Hcc[(size_t)c * 36 + b * 6 + a] = s;
    } else {
        gc[(size_t)c * 6 + (k - 21)] = s;
    }
}

int launch_camera_blocks(int n_cams, int n_chunks, const int* chunk_begin, const int* chunk_end,
                         const int* cam_chunk_start, const int* cam_perm, const double* J8, const unsigned char* omask,
                         const double* Jc12, const double2* r, double* partial, double* Hcc, double* gc, hipStream_t st) {
    if (n_chunks > 0) {
        if (Jc12) hipLaunchKernelGGL(ba_camera_partial_kernel<true>, dim3((n_chunks + 3) / 4), dim3(256), 0, st, n_chunks,
                                     chunk_begin, chunk_end, cam_perm, J8, omask, Jc12, r, partial);
        else hipLaunchKernelGGL(ba_camera_partial_kernel<false>, dim3((n_chunks + 3) / 4), dim3(256), 0, st, n_chunks,
                                chunk_begin, chunk_end, cam_perm, J8, omask, Jc12, r, partial);
    }
    hipLaunchKernelGGL(ba_camera_final_kernel, dim3((n_cams * 27 + 255) / 256), dim3(256), 0, st, n_cams,
                       cam_chunk_start, partial, Hcc, gc);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

// ===========================================================================================
// Levenberg-Marquardt diagonal (Ceres LevenbergMarquardtStrategy + jacobi_scaling):
//   scale_i = 1/(1+sqrt(H_ii))            (first call only)
//   d_i = clamp(H_ii scale_i^2, min, max) / radius / scale_i^2
// diag entry i of a block array with `bs` dofs per block, `bstride` doubles per block and
// diagonal element k at offset doff(k).
// ===========================================================================================
__global__ __launch_bounds__(256) void lm_diagonal_kernel(int n, int bs, int bstride, int kind,
                                                          const double* __restrict__ H, double* __restrict__ scale,
                                                          int init_scale, int use_scaling,
                                                          const double* __restrict__ radius_dev, double radius_host,
                                                          double dmin, double dmax, double* __restrict__ d) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int blk = i / bs, k = i - blk * bs;
    // kind 0: full bs x bs row-major block; kind 1: symmetric 3x3 packed (xx xy xz yy yz zz)
    // kind 2: H is the diagonal itself
    const int off = (kind == 0) ? k * (bs + 1) : (kind == 1 ? (k == 0 ? 0 : (k == 1 ? 3 : 5)) : 0);
    const double h = H[(size_t)blk * bstride + off];
    double s = 1.0;
    if (use_scaling) {
        if (init_scale) { s = 1.0 / (1.0 + sqrt(h)); scale[i] = s; }
        else s = scale[i];
    } else if (init_scale) scale[i] = 1.0;
    const double radius = radius_dev ? radius_dev[0] : radius_host;
    const double s2 = s * s;
    const double v = fmin(fmax(h * s2, dmin), dmax);
    d[i] = v / radius / s2;
}

int launch_lm_diagonal(int n, int bs, int bstride, int kind, const double* H, double* scale, int init_scale,
                       int use_scaling, double radius, double dmin, double dmax, double* d, hipStream_t st) {
    hipLaunchKernelGGL(lm_diagonal_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, bs, bstride, kind, H,
                       scale, init_scale, use_scaling, (const double*)nullptr, radius, dmin, dmax, d);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

// (Hpp + diag(dp))^-1 per landmark; zero for constant / degenerate landmarks
__global__ __launch_bounds__(256) void ba_point_invert_kernel(int n_pts, const double* __restrict__ Hpp6,
                                                              const double* __restrict__ dp,
                                                              const unsigned char* __restrict__ pt_fixed,
                                                              double* __restrict__ Hinv6) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_pts) return;
    double H[6], Hi[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 6; ++k) H[k] = Hpp6[(size_t)j * 6 + k];
    H[0] += dp[(size_t)j * 3]; H[3] += dp[(size_t)j * 3 + 1]; H[5] += dp[(size_t)j * 3 + 2];
    const bool fixed = pt_fixed ? (pt_fixed[j] != 0) : false;
    if (fixed || !inv3_sym6(H, Hi)) {
#pragma unroll
        for (int k = 0; k < 6; ++k) Hi[k] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) Hinv6[(size_t)j * 6 + k] = Hi[k];
}

// the same with the landmark's LM diagonal computed on the way (lm_diagonal_kernel, kind 1: diagonal of the packed 3x3
// block) and stored in dp / scale_p as the separate kernel does
__global__ __launch_bounds__(256) void ba_point_damp_invert_kernel(int n_pts, const double* __restrict__ Hpp6,
                                                                   const unsigned char* __restrict__ pt_fixed,
                                                                   double* __restrict__ scale, int init_scale, int use_scaling,
                                                                   double radius, double dmin, double dmax, double* __restrict__ dp,
                                                                   double* __restrict__ Hinv6, double* __restrict__ zero_buf, int zero_n) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < zero_n) zero_buf[j] = 0.0;          // (the three extras vectors behind S: one memset launch less per iteration)
    if (j >= n_pts) return;
    double H[6], Hi[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 6; ++k) H[k] = Hpp6[(size_t)j * 6 + k];
    const int dg[3] = {0, 3, 5};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const size_t i = (size_t)j * 3 + k;
        const double h = H[dg[k]];
        double s = 1.0;
        if (use_scaling) {
            if (init_scale) { s = 1.0 / (1.0 + sqrt(h)); scale[i] = s; }
            else s = scale[i];
        } else if (init_scale) scale[i] = 1.0;
        const double s2 = s * s;
        const double v = fmin(fmax(h * s2, dmin), dmax);
        const double d = v / radius / s2;
        dp[i] = d;
        H[dg[k]] = h + d;
    }
    const bool fixed = pt_fixed ? (pt_fixed[j] != 0) : false;
    if (fixed || !inv3_sym6(H, Hi)) {
#pragma unroll
        for (int k = 0; k < 6; ++k) Hi[k] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) Hinv6[(size_t)j * 6 + k] = Hi[k];
}

int launch_point_damp_invert(int n_pts, const double* Hpp6, const unsigned char* pt_fixed, double* scale, int init_scale,
                             int use_scaling, double radius, double dmin, double dmax, double* dp, double* Hinv6,
                             double* zero_buf, int zero_n, hipStream_t st) {
    hipLaunchKernelGGL(ba_point_damp_invert_kernel, dim3((std::max(n_pts, zero_n) + 255) / 256), dim3(256), 0, st, n_pts, Hpp6, pt_fixed, scale,
                       init_scale, use_scaling, radius, dmin, dmax, dp, Hinv6, zero_buf, zero_n);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

int launch_point_invert(int n_pts, const double* Hpp6, const double* dp, const unsigned char* pt_fixed,
                        double* Hinv6, hipStream_t st) {
    hipLaunchKernelGGL(ba_point_invert_kernel, dim3((n_pts + 255) / 256), dim3(256), 0, st, n_pts, Hpp6, dp,
                       pt_fixed, Hinv6);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

// ===========================================================================================
// Schur complement of the landmark blocks:  S[c_i, c_l] -= E_i W_l^T,  rhs[c_i] += E_i gp_j  for every pair of
// observations (i, l) of one landmark j with c_l <= c_i,  W = Jc^T Jp (6x3),  E_i = W_i Hpp_j^-1.
// One workgroup per TASK = (camera row c, a slice of that row's non-zero 6x6 blocks).  The blocks of the slice are
// accumulated in LDS with ds_add_f64 and written to HBM once: no global atomics and no read-modify-write of the 288 MB
// matrix (global FP64 atomics into S: 8.9 ms; a serial partner loop per observation: 0.52 ms; this form 0.27 ms in round
// 2).  The pairs are enumerated ON THE HOST once (the structure is static), with the LDS slot of every pair's block
// resolved; one lane handles one PAIR per trip: two independent 64 B gathers (J_i, J_l) plus the landmark's inverse
// block, ~220 FMAs, 36 LDS atomics -- every iteration of every lane is independent.
// The task zeroes its own stretch of its six rows of S (no 288 MB memset in front of the kernel), and the slice that
// holds the diagonal block also makes the camera blocks Hcc = sum Jc^T Jc, gc = sum Jc^T r of its camera -- a wave
// reduction over the camera's observations, whose records are in this CU's caches at that moment (as a kernel of its
// own that gather cost 46 us per iteration).  Fixed summation order: Hcc and gc are bitwise reproducible.
// ===========================================================================================
template <int ROTS, bool GEN = false, int MODE = 0>
__global__ __launch_bounds__(SCHUR_THREADS, GEN ? 2 : 4) void ba_schur_pairs_kernel(SchurArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int task = blockIdx.x;
    const int c = a.task_cam[task];
    const int row_ncols = a.row_col_ptr[c + 1] - a.row_col_ptr[c];
    const int clo = a.task_col_lo[task], chi = a.task_col_hi[task];       // this task's slice of the row's blocks
    const int col0 = a.row_col_ptr[c] + clo, ncols = chi - clo;
    double* acc = smem;                          // [accumulator slots][SCHUR_BLK_LD]: a block has one slot, or several consecutive ones (parts)
    double* racc = smem + (size_t)a.max_cols * SCHUR_BLK_LD;   // [8]
    double* cpart = racc + 8;                    // [8 waves][SCHUR_CAM_LD]: the waves' partial camera blocks and (i, i) terms
    int* cols = reinterpret_cast<int*>(cpart + 8 * SCHUR_CAM_LD);    // [ncols]
    int* vsf = cols + a.max_cols;                // [ncols + 1]: first accumulator slot of every block, and the slot count
    const int tid = threadIdx.x;
    const bool diag_piece = (chi == row_ncols);  // (the diagonal block is the last one of its row)
    const int* vs_g = a.vs_first + a.task_vs_ptr[task];
    const int nslots = vs_g[ncols];
    for (int e = tid; e < nslots * SCHUR_BLK_LD; e += SCHUR_THREADS) acc[e] = 0.0;
    if (tid < 8) racc[tid] = 0.0;
    for (int e = tid; e < ncols; e += SCHUR_THREADS) cols[e] = a.row_cols[col0 + e];
    for (int e = tid; e <= ncols; e += SCHUR_THREADS) vsf[e] = vs_g[e];
    if (tid == 0) vsf[a.max_cols + 1] = 0;       // the token of mode 1
    const bool lms = a.part != nullptr;          // landmark-range slices: partial blocks out, the reduce kernel writes (and zeroes) S
    if (!lms) {
        // this slice's stretch of the camera's six rows of S is zeroed HERE (the whole row up to the end of the diagonal
        // 128-tile, shared out between the slices; nothing right of it is ever read): the stores drain under the pair loop
        const int cend = min(a.lda, ((c * 6 + 5) / 128 + 1) * 128);
        const int z0 = (clo == 0) ? 0 : 6 * a.row_cols[col0];
        const int z1 = diag_piece ? cend : 6 * a.row_cols[col0 + ncols];
        for (int q = 0; q < 6; ++q) {
            double* row = a.S + (size_t)(c * 6 + q) * a.lda;
            for (int e = z0 + tid; e < z1; e += SCHUR_THREADS) row[e] = 0.0;
        }
    }
    __syncthreads();                             // (also orders the zero stores before the block stores below)
    // One pair per lane and trip.  The record carries (i, l, landmark, slot), so all gathers of a pair are one level
    // deep.  (Two pairs in flight per lane -- the second pair's gathers requested before the first is computed -- was
    // measured 7 % slower: the kernel is not bound by the gather latency.)
    // Every wave walks ITS OWN list: the host dealt the task's blocks (parts of heavy blocks) to the eight waves, so all the LDS
    // adds that meet in one address come from one wave, in program order -- S is bitwise reproducible (see stba_ba_create).
    // a.mode 1 (round 5, second form): ONE list per task walked by all 512 lanes as until round 4 (lane t takes pair t + 512 trip),
    // and the eight waves add their contributions of a trip IN WAVE ORDER: a token in LDS -- wave w of trip n waits for 8 n + w,
    // adds, waits for its LDS operations, passes 8 n + w + 1 on.  Same order of additions into every address in every run.
    // a.mode 2: the same list without the token (the round-4 kernel: arrival order; kept as the timing reference).
    const int wv = tid >> 6;
    constexpr int mode = MODE;                   // (the product build instantiates mode 0 only; debug builds all three: tools/dbg/schur_modes.py)
    const int stride = mode == 0 ? 64 : SCHUR_THREADS;
    const int ent = task * (SCHUR_THREADS / 64) + (mode == 0 ? wv : 0);
    const int kb = a.pair_begin[ent], ke = a.pair_end[ent];
    const int ntrips = (ke - kb + stride - 1) / stride;
    int* turn = vsf + a.max_cols + 1;
    // (the pair record runs one trip ahead: one dependent memory round trip less per trip)
    int k = kb + (mode == 0 ? (tid & 63) : tid);
    int4 rn = (k < ke) ? a.pair_rec[k] : make_int4(0, 0, 0, 0);
    // (mode 0: every lane leaves when ITS share of the wave's list is done, as in the round-4 loop; the shared-list modes run a common
    // trip count -- the token must pass through every wave)
    for (int trip = 0; mode == 0 ? (k < ke) : (trip < ntrips); ++trip, k += stride) {
        const int4 rc = rn;
        const bool valid = mode == 0 ? true : (k < ke);
        if (k + stride < ke) rn = a.pair_rec[k + stride];
        if (mode == 1) {
            // (the gathers below are requested first: the wait for the token hides behind them only if they are in flight)
        }
        double Hi[6], jc[12], jp[6], jc2[12], jp2[6];
#ifdef STBA_DEBUG_KNOBS
        const int gi = (a.ablate & 1) ? (rc.x & 63) : rc.x, gl = (a.ablate & 1) ? (rc.y & 63) : rc.y, gz = (a.ablate & 1) ? (rc.z & 63) : rc.z;
#else
        const int gi = rc.x, gl = rc.y, gz = rc.z;
#endif
#pragma unroll
        for (int k2 = 0; k2 < 6; ++k2) Hi[k2] = a.Hinv6[(size_t)gz * 6 + k2];
        load_jc_jp<GEN>(a.J8, a.omask, gi, jc, jp, a.Jc12);
        load_jc_jp<GEN>(a.J8, a.omask, gl, jc2, jp2, a.Jc12);
        const unsigned sl = (unsigned)rc.w;
        const int rot = (int)((sl >> 16) & 7u);          // the column this lane starts at: dealt by the host (stba_ba_create), 0 .. ROTS - 1
        // E_i = (Jc_i^T Jp_i) Hinv_j recomputed per pair (cheaper than a pre-pass that stores it)
        double E[18];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const double w0 = jc[q] * jp[0] + jc[6 + q] * jp[3];
            const double w1 = jc[q] * jp[1] + jc[6 + q] * jp[4];
            const double w2 = jc[q] * jp[2] + jc[6 + q] * jp[5];
            E[q * 3 + 0] = w0 * Hi[0] + w1 * Hi[1] + w2 * Hi[2];
            E[q * 3 + 1] = w0 * Hi[1] + w1 * Hi[3] + w2 * Hi[4];
            E[q * 3 + 2] = w0 * Hi[2] + w1 * Hi[4] + w2 * Hi[5];
        }
        double* blk = acc + (size_t)(sl & 0x3fffu) * SCHUR_BLK_LD;
        if (mode == 1) {
            const int want = trip * (SCHUR_THREADS / 64) + wv;
            while (__hip_atomic_load(turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != want) __builtin_amdgcn_s_sleep(1);
        }
        if (valid) {
        // The lanes of a wave walk the six columns of their blocks in ROTS different rotations: the pairs of one wave hit the same
        // block again and again -- the cameras next to c share most of its landmarks: 25 of 64 lanes share their block with an
        // earlier lane, the busiest block of a wave has 7 -- and lanes that add to the SAME address in the same instruction are
        // served one after the other.  With the rotation they meet in different columns.  The rotation comes with the record:
        // the host deals it as the pair's rank among the pairs of its trip that share its slot (round 6; lane mod 3 until then).
#pragma unroll
        for (int s2 = 0; s2 < 6; ++s2) {
            double ja = jc2[s2], jb = jc2[6 + s2];
            int b = s2;
#pragma unroll
            for (int r = 1; r < ROTS; ++r) {
                const int bb = (s2 + r * (6 / ROTS)) % 6;
                if (rot == r) { ja = jc2[bb]; jb = jc2[6 + bb]; b = bb; }
            }
            const double w0 = ja * jp2[0] + jb * jp2[3];
            const double w1 = ja * jp2[1] + jb * jp2[4];
            const double w2 = ja * jp2[2] + jb * jp2[5];
            double* col = blk + b;
            // (no tests for zero contributions of constant dofs: adding a zero is harmless)
            // (every entry, also the upper triangle of a diagonal block, which the write-out never reads -- pairs in the
            // diagonal block are the rare (i, l != i) of one camera, and a test per atomic costs an exec-mask branch each)
#ifdef STBA_DEBUG_KNOBS
            if (a.ablate & 2) {
                double sacc = 0.0;
#pragma unroll
                for (int q = 0; q < 6; ++q) sacc += E[q * 3] * w0 + E[q * 3 + 1] * w1 + E[q * 3 + 2] * w2;
                if (sacc == 1.2345e300) unsafeAtomicAdd(&col[0], sacc);
            } else
#endif
#pragma unroll
            for (int q = 0; q < 6; ++q)
                unsafeAtomicAdd(&col[q * 6], -(E[q * 3] * w0 + E[q * 3 + 1] * w1 + E[q * 3 + 2] * w2));
        }
        }
        if (mode == 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if ((tid & 63) == 0) __hip_atomic_store(turn, trip * (SCHUR_THREADS / 64) + wv + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (diag_piece) {
        // camera blocks of camera c: 21 unique entries of Jc^T Jc and 6 of Jc^T r summed over the camera's observations --
        // a strided share per lane, a shuffle tree per wave, the eight waves' partial sums added in order below
        // The same pass makes the pairs (i, i): every observation's own term E_i W_i^T of the DIAGONAL block and its share
        // E_i gp of the right-hand side.  As LDS atomics they were the worst ones of the kernel -- a thousand pairs per row
        // adding to the same 21 + 6 addresses, a dozen of them in every wave (27 of the 198 M atomics of a C5 launch);
        // here they are register sums like the camera block.
        // (two passes over the camera's records, 27 running sums each: all 54 in one loop need 238 registers and halve
        // the occupancy of the whole kernel; the second pass finds the records in the L1 / L2)
        const int pb = lms ? a.task_p_lo[task] : a.cam_start[c];      // (a landmark-range slice: its own stretch of the camera's list)
        const int pe = lms ? a.task_p_hi[task] : a.cam_start[c + 1];
        {
            double h[27];
#pragma unroll
            for (int k = 0; k < 27; ++k) h[k] = 0.0;
            for (int p = pb + tid; p < pe; p += SCHUR_THREADS) {
                const int i = a.cam_perm[p];
                double j[12], jpu[6];
                load_jc_jp<GEN>(a.J8, a.omask, i, j, jpu, a.Jc12);
                const double2 ri = a.r[i];
                int idx = 0;
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int b = 0; b <= q; ++b) h[idx++] += j[q] * j[b] + j[6 + q] * j[6 + b];
#pragma unroll
                for (int q = 0; q < 6; ++q) h[21 + q] += j[q] * ri.x + j[6 + q] * ri.y;
            }
#pragma unroll
            for (int k = 0; k < 27; ++k) {
                double v = h[k];
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                if ((tid & 63) == 0) cpart[(tid >> 6) * SCHUR_CAM_LD + k] = v;
            }
        }
        {
            double h[27];       // [0 .. 20] = -sum E W^T (lower triangle, row-wise), [21 .. 26] = sum E gp
#pragma unroll
            for (int k = 0; k < 27; ++k) h[k] = 0.0;
            for (int p = pb + tid; p < pe; p += SCHUR_THREADS) {
                const int i = a.cam_perm[p];
                double j[12], jpu[6];
                load_jc_jp<GEN>(a.J8, a.omask, i, j, jpu, a.Jc12);
                const int lm = a.obs_pt[i];
                double Hi[6];
#pragma unroll
                for (int k2 = 0; k2 < 6; ++k2) Hi[k2] = a.Hinv6[(size_t)lm * 6 + k2];
                const double g0 = a.gp[(size_t)lm * 3], g1 = a.gp[(size_t)lm * 3 + 1], g2 = a.gp[(size_t)lm * 3 + 2];
                double W[18];
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    W[q * 3 + 0] = j[q] * jpu[0] + j[6 + q] * jpu[3];
                    W[q * 3 + 1] = j[q] * jpu[1] + j[6 + q] * jpu[4];
                    W[q * 3 + 2] = j[q] * jpu[2] + j[6 + q] * jpu[5];
                }
                int idx = 0;
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const double e0 = W[q * 3] * Hi[0] + W[q * 3 + 1] * Hi[1] + W[q * 3 + 2] * Hi[2];
                    const double e1 = W[q * 3] * Hi[1] + W[q * 3 + 1] * Hi[3] + W[q * 3 + 2] * Hi[4];
                    const double e2 = W[q * 3] * Hi[2] + W[q * 3 + 1] * Hi[4] + W[q * 3 + 2] * Hi[5];
#pragma unroll
                    for (int b = 0; b <= q; ++b) h[idx++] -= e0 * W[b * 3] + e1 * W[b * 3 + 1] + e2 * W[b * 3 + 2];
                    h[21 + q] += e0 * g0 + e1 * g1 + e2 * g2;
                }
            }
#pragma unroll
            for (int k = 0; k < 27; ++k) {
                double v = h[k];
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                if ((tid & 63) == 0) cpart[(tid >> 6) * SCHUR_CAM_LD + 27 + k] = v;
            }
        }
        __syncthreads();
        // the eight waves' sums of the diagonal block and of the right-hand side join the LDS accumulators (in wave order)
        if (tid < 27 && ncols > 0) {
            double s2 = 0.0;
#pragma unroll
            for (int w = 0; w < SCHUR_THREADS / 64; ++w) s2 += cpart[w * SCHUR_CAM_LD + 27 + tid];
            if (tid < 21) {
                int q = 0, b = tid;
                while (b > q) { ++q; b -= q; }   // tid = q(q+1)/2 + b
                acc[(size_t)vsf[ncols - 1] * SCHUR_BLK_LD + q * 6 + b] += s2;      // (the diagonal block is the slice's last one)
            } else {
                racc[tid - 21] += s2;
            }
        }
    }
    __syncthreads();
    if (lms) {
        // a landmark-range slice: everything it has summed goes to ITS stretch of the partial buffer, dense and coalesced --
        // [ncols x 36 block entries | rhs at +0..5 | camera sums (21 + 6) at +8..34]; ba_schur_reduce_slices_kernel adds the slices
        double* out = a.part + a.task_part_off[task];
        for (int e = tid; e < ncols * 36; e += SCHUR_THREADS) {
            const int slot = e / 36, k = e - slot * 36;
            const int v1 = vsf[slot + 1];
            double sum = acc[vsf[slot] * SCHUR_BLK_LD + k];
            for (int v = vsf[slot] + 1; v < v1; ++v) sum += acc[v * SCHUR_BLK_LD + k];
            out[e] = sum;
        }
        double* tail = out + (size_t)ncols * 36;
        if (tid < 6) tail[tid] = racc[tid];
        if (tid >= 64 && tid < 64 + 27) {
            const int k = tid - 64;
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < SCHUR_THREADS / 64; ++w) s += cpart[w * SCHUR_CAM_LD + k];
            tail[8 + k] = s;
        }
        return;
    }
    for (int e = tid; e < ncols * 36; e += SCHUR_THREADS) {
        const int slot = e / 36, k = e - slot * 36, q = k / 6, b = k - q * 6;
        const int c2 = cols[slot];
        if (c2 == c && b > q) continue;
        const int v1 = vsf[slot + 1];
        double sum = acc[vsf[slot] * SCHUR_BLK_LD + k];
        for (int v = vsf[slot] + 1; v < v1; ++v) sum += acc[v * SCHUR_BLK_LD + k];     // the parts of a heavy block, in order
        a.S[(size_t)(c * 6 + q) * a.lda + c2 * 6 + b] = sum;
    }
    if (diag_piece) {                            // (the slice with the diagonal block carries the right-hand side and the camera blocks)
        if (tid < 6) a.rhs[c * 6 + tid] = racc[tid];
        if (tid >= 64 && tid < 64 + 27) {
            const int k = tid - 64;
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < SCHUR_THREADS / 64; ++w) s += cpart[w * SCHUR_CAM_LD + k];
            if (k < 21) {
                int q = 0, b = k;
                while (b > q) { ++q; b -= q; }   // k = q(q+1)/2 + b
                a.Hcc[(size_t)c * 36 + q * 6 + b] = s;
                a.Hcc[(size_t)c * 36 + b * 6 + q] = s;
            } else {
                a.gc[(size_t)c * 6 + (k - 21)] = s;
            }
        }
    }
}

// ===========================================================================================
// The same Schur complement for DENSE visibility (most cameras see most landmarks): per observation i of landmark j the 6 x 3 block
// Y_i = (Jc_i^T Jp_i) R_j with R_j R_j^T = Hpp_j^-1 (Cholesky of the inverse landmark block), written into a dense matrix
// Y [6 n_cams padded] x [3 n_pts padded]; then S = -(Y Y^T) (lower triangle) is ONE symmetric rank-k product on the matrix cores
// (chol_yyt_lower_dev) and rhs = Y v, v_j = R_j^T gp_j.  No pair plan: the pair formulation costs 16 B of plan and 36 LDS atomics
// per pair of observations of a landmark -- k (k + 1) / 2 pairs for a landmark seen by k cameras; the product costs 6 C x 6 C x 3
// multiply-adds per landmark whatever k is, and wins from k / C ~ 0.3 on.  The zero blocks of Y (camera does not see landmark)
// are zeroed once, at the first launch: the visibility pattern is static.
// ===========================================================================================
__device__ inline void chol3_of_sym6(const double h[6], double R[6]) {
    // h = (00, 01, 02, 11, 12, 22) of an SPD (or zero) 3 x 3 block; R = (r00, r10, r20, r11, r21, r22) lower triangular, R R^T = h
    const double r00 = h[0] > 0.0 ? sqrt(h[0]) : 0.0;
    const double i0 = r00 > 0.0 ? 1.0 / r00 : 0.0;
    const double r10 = h[1] * i0, r20 = h[2] * i0;
    const double d1 = h[3] - r10 * r10;
    const double r11 = d1 > 0.0 ? sqrt(d1) : 0.0;
    const double i1 = r11 > 0.0 ? 1.0 / r11 : 0.0;
    const double r21 = (h[4] - r20 * r10) * i1;
    const double d2 = h[5] - r20 * r20 - r21 * r21;
    const double r22 = d2 > 0.0 ? sqrt(d2) : 0.0;
    R[0] = r00; R[1] = r10; R[2] = r20; R[3] = r11; R[4] = r21; R[5] = r22;
}
// One WAVE per chunk of <= CAM_CHUNK observations of ONE camera (the camera-sorted permutation, landmarks ascending inside a camera:
// the lanes of a wave write neighbouring 24-byte pieces of the camera's six rows of Y).  On the way the wave sums what else the
// Schur step owes for its camera: the camera block Jc^T Jc (21), Jc^T r (6) and the right-hand side sum_i W_i Hpp^-1 gp (6); a
// second kernel adds a camera's chunks in order (no atomics: bitwise reproducible).
constexpr int YCAM_LD = 36;      // doubles per chunk partial: 21 + 6 + 6, padded
template <bool GEN, bool DUP>
__global__ __launch_bounds__(256) void ba_schur_dense_chunk_kernel(int n_chunks, const int* __restrict__ chunk_begin, const int* __restrict__ chunk_end,
                                                                   const int* __restrict__ cam_perm, const int* __restrict__ obs_cam,
                                                                   const int* __restrict__ obs_pt, const double* __restrict__ J8,
                                                                   const unsigned char* __restrict__ omask, const double* __restrict__ Jc12,
                                                                   const double2* __restrict__ r, const double* __restrict__ Hinv6,
                                                                   const double* __restrict__ gp, double* __restrict__ Y, size_t ldy,
                                                                   double* __restrict__ partial, const unsigned char* __restrict__ dup_run) {
    const int lane = threadIdx.x & 63;
    const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ch >= n_chunks) return;
    double acc[33];
#pragma unroll
    for (int k = 0; k < 33; ++k) acc[k] = 0.0;
    const int e = chunk_end[ch];
    for (int p = chunk_begin[ch] + lane; p < e; p += 64) {
        const int i = cam_perm[p];
        const int c = obs_cam[i], j = obs_pt[i];
        double jc[12], jp[6], h[6], R[6];
        load_jc_jp<GEN>(J8, omask, i, jc, jp, Jc12);
        const double2 ri = r[i];
#pragma unroll
        for (int k = 0; k < 6; ++k) h[k] = Hinv6[(size_t)j * 6 + k];
        const double g0 = gp[(size_t)j * 3], g1 = gp[(size_t)j * 3 + 1], g2 = gp[(size_t)j * 3 + 2];
        chol3_of_sym6(h, R);
        const double t0 = h[0] * g0 + h[1] * g1 + h[2] * g2, t1 = h[1] * g0 + h[3] * g1 + h[4] * g2, t2 = h[2] * g0 + h[4] * g1 + h[5] * g2;
        int idx = 0;
        double wrow[DUP ? 18 : 1];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const double w0 = jc[q] * jp[0] + jc[6 + q] * jp[3];
            const double w1 = jc[q] * jp[1] + jc[6 + q] * jp[4];
            const double w2 = jc[q] * jp[2] + jc[6 + q] * jp[5];
            if constexpr (DUP) { wrow[q * 3] = w0; wrow[q * 3 + 1] = w1; wrow[q * 3 + 2] = w2; }
            else {
                double* y = Y + (size_t)(c * 6 + q) * ldy + (size_t)j * 3;
                y[0] = w0 * R[0] + w1 * R[1] + w2 * R[2];
                y[1] = w1 * R[3] + w2 * R[4];
                y[2] = w2 * R[5];
            }
#pragma unroll
            for (int b = 0; b <= q; ++b) acc[idx++] += jc[q] * jc[b] + jc[6 + q] * jc[6 + b];
            acc[21 + q] += jc[q] * ri.x + jc[6 + q] * ri.y;
            acc[27 + q] += w0 * t0 + w1 * t1 + w2 * t2;
        }
        // Several observations of the SAME (camera, landmark) pair (stereo residuals on one pose block, two factors on one pair): the
        // block of Y is the SUM of their W (S = -sum_j (sum_i W_i) Hpp^-1 (sum_l W_l)^T is bilinear in the per-camera sums).  They sit
        // next to each other in the camera's list; the first one of a run adds its followers' W in list order and writes the block,
        // the followers only take part in the camera sums above.  dup_run: 0 = alone, k = leader of k followers, 255 = follower
        // (DUP = false when the problem has no such pair -- the usual case: the blocks are written in the loop above).
        if constexpr (DUP) {
        const unsigned run = dup_run[p];
        if (run == 255u) continue;
        for (unsigned d = 1; d <= run; ++d) {
            double jc2[12], jp2[6];
            load_jc_jp<GEN>(J8, omask, cam_perm[p + d], jc2, jp2, Jc12);
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                wrow[q * 3] += jc2[q] * jp2[0] + jc2[6 + q] * jp2[3];
                wrow[q * 3 + 1] += jc2[q] * jp2[1] + jc2[6 + q] * jp2[4];
                wrow[q * 3 + 2] += jc2[q] * jp2[2] + jc2[6 + q] * jp2[5];
            }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            double* y = Y + (size_t)(c * 6 + q) * ldy + (size_t)j * 3;
            y[0] = wrow[q * 3] * R[0] + wrow[q * 3 + 1] * R[1] + wrow[q * 3 + 2] * R[2];
            y[1] = wrow[q * 3 + 1] * R[3] + wrow[q * 3 + 2] * R[4];
            y[2] = wrow[q * 3 + 2] * R[5];
        }
        }
    }
#pragma unroll
    for (int k = 0; k < 33; ++k) {
        double v = acc[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        acc[k] = v;
    }
    if (lane == 0) {
        double* o = partial + (size_t)ch * YCAM_LD;
#pragma unroll
        for (int k = 0; k < 33; ++k) o[k] = acc[k];
    }
}
__global__ __launch_bounds__(256) void ba_schur_dense_final_kernel(int n_cams, const int* __restrict__ cam_chunk_start, const double* __restrict__ partial,
                                                                   double* __restrict__ Hcc, double* __restrict__ gc, double* __restrict__ rhs) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int c = gid / 33, k = gid - c * 33;
    if (c >= n_cams) return;
    double s = 0.0;
    const int e = cam_chunk_start[c + 1];
    for (int ch = cam_chunk_start[c]; ch < e; ++ch) s += partial[(size_t)ch * YCAM_LD + k];
    if (k < 21) {
        int a = 0, b = k;
        while (b > a) { ++a; b -= a; }   // k = a(a+1)/2 + b
        Hcc[(size_t)c * 36 + a * 6 + b] = s;
        Hcc[(size_t)c * 36 + b * 6 + a] = s;
    } else if (k < 27) gc[(size_t)c * 6 + (k - 21)] = s;
    else rhs[(size_t)c * 6 + (k - 27)] = s;
}
size_t schur_dense_partial_doubles(int n_chunks) { return (size_t)std::max(1, n_chunks) * YCAM_LD; }

int launch_schur_dense(const SchurDenseArgs& a, hipStream_t st) {
    if (a.n_chunks > 0) {
        const dim3 grid((a.n_chunks + 3) / 4);
#define STBA_DENSE_CHUNK(GEN_, DUP_) hipLaunchKernelGGL((ba_schur_dense_chunk_kernel<GEN_, DUP_>), grid, dim3(256), 0, st, a.n_chunks, a.chunk_begin, \
            a.chunk_end, a.cam_perm, a.obs_cam, a.obs_pt, a.J8, a.omask, a.Jc12, a.r, a.Hinv6, a.gp, a.Y, a.ldy, a.partial, a.dup_run)
        if (a.Jc12) { if (a.dup_run) STBA_DENSE_CHUNK(true, true); else STBA_DENSE_CHUNK(true, false); }
        else { if (a.dup_run) STBA_DENSE_CHUNK(false, true); else STBA_DENSE_CHUNK(false, false); }
#undef STBA_DENSE_CHUNK
    }
    STBA_TRY(chol_yyt_lower_dev(a.Y, a.ldy, a.kcols, a.S, a.lda, a.ws, st));
    if (a.n_cams > 0)
        hipLaunchKernelGGL(ba_schur_dense_final_kernel, dim3((a.n_cams * 33 + 255) / 256), dim3(256), 0, st, a.n_cams, a.cam_chunk_start, a.partial,
                           a.Hcc, a.gc, a.rhs);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

// Landmark-range slices (few camera rows): one workgroup per camera row adds the row's slices IN LIST ORDER -- blocks, right-hand
// side, camera block and gradient -- and writes the row of S (zeroed up to the end of its diagonal 128-tile first, as the column
// slices do for themselves).  Fixed order, no atomics: bitwise reproducible.
__global__ __launch_bounds__(256) void ba_schur_reduce_slices_kernel(SchurArgs a) {
    const int c = blockIdx.x, tid = threadIdx.x;
    const int col0 = a.row_col_ptr[c], ncols = a.row_col_ptr[c + 1] - col0;
    const int t0 = a.row_task_ptr[c], t1 = a.row_task_ptr[c + 1];
    const int cend = min(a.lda, ((c * 6 + 5) / 128 + 1) * 128);
    for (int q = 0; q < 6; ++q) {
        double* row = a.S + (size_t)(c * 6 + q) * a.lda;
        for (int e = tid; e < cend; e += 256) row[e] = 0.0;
    }
    __syncthreads();                             // (orders the zero stores before the block stores)
    for (int e = tid; e < ncols * 36; e += 256) {
        const int slot = e / 36, k = e - slot * 36, q = k / 6, b = k - q * 6;
        const int c2 = a.row_cols[col0 + slot];
        if (c2 == c && b > q) continue;
        double sum = 0.0;
        for (int t = t0; t < t1; ++t) sum += a.part[a.task_part_off[a.row_tasks[t]] + e];
        a.S[(size_t)(c * 6 + q) * a.lda + c2 * 6 + b] = sum;
    }
    if (tid < 6 + 27) {
        const int off = tid < 6 ? tid : 8 + (tid - 6);
        double sum = 0.0;
        for (int t = t0; t < t1; ++t) sum += a.part[a.task_part_off[a.row_tasks[t]] + (size_t)ncols * 36 + off];
        if (tid < 6) a.rhs[c * 6 + tid] = sum;
        else {
            const int k = tid - 6;
            if (k < 21) {
                int q = 0, b = k;
                while (b > q) { ++q; b -= q; }
                a.Hcc[(size_t)c * 36 + q * 6 + b] = sum;
                a.Hcc[(size_t)c * 36 + b * 6 + q] = sum;
            } else a.gc[(size_t)c * 6 + (k - 21)] = sum;
        }
    }
}

size_t schur_rows_lds_bytes(int max_cols) {
    return ((size_t)max_cols * SCHUR_BLK_LD + 8 + 8 * SCHUR_CAM_LD) * sizeof(double) + ((size_t)2 * max_cols + 2) * sizeof(int) + 16;
}

// (GEN: host-linearised factors, a.Jc12 -- the same kernel with the camera block read from its own array)
template <bool GEN, int MODE>
static int launch_schur_inst(const SchurArgs& a, int n_tasks, size_t lds, hipStream_t st) {
    // the largest task the engine ever builds (SCHUR_MAX_SLOTS accumulator slots) fixes the LDS limit, once per device
    static DeviceOnce attr;
    STBA_TRY(attr.run([]() -> int {
        STBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ba_schur_pairs_kernel<SCHUR_ROTS, GEN, MODE>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)schur_rows_lds_bytes(SCHUR_MAX_SLOTS)));
        return STBA_OK;
    }));
    hipLaunchKernelGGL((ba_schur_pairs_kernel<SCHUR_ROTS, GEN, MODE>), dim3(n_tasks), dim3(SCHUR_THREADS), lds, st, a);
    return STBA_OK;
}

int launch_schur_rows(const SchurArgs& a, int n_tasks, hipStream_t st) {
    if (n_tasks <= 0) return STBA_OK;
    const size_t lds = schur_rows_lds_bytes(a.max_cols);

    // (column rotations measured at C5: 1 / 2 / 3 / 6 -> 0.283 / 0.268 / 0.264 / 0.266 ms)
#ifdef STBA_DEBUG_KNOBS
    if (a.mode == 1) STBA_TRY(a.Jc12 ? (launch_schur_inst<true, 1>(a, n_tasks, lds, st)) : (launch_schur_inst<false, 1>(a, n_tasks, lds, st)));
    else if (a.mode == 2) STBA_TRY(a.Jc12 ? (launch_schur_inst<true, 2>(a, n_tasks, lds, st)) : (launch_schur_inst<false, 2>(a, n_tasks, lds, st)));
    else
#endif
    STBA_TRY(a.Jc12 ? (launch_schur_inst<true, 0>(a, n_tasks, lds, st)) : (launch_schur_inst<false, 0>(a, n_tasks, lds, st)));
    if (a.part && a.n_cams > 0) hipLaunchKernelGGL(ba_schur_reduce_slices_kernel, dim3(a.n_cams), dim3(256), 0, st, a);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

// S diagonal blocks += Hcc (this rank's partial), rhs -= gc; packs diag(Hcc) and gc behind S
// so that one all-reduce carries everything (extras: [diagHcc | gc], n entries each).
__global__ __launch_bounds__(256) void ba_reduced_add_camera_kernel(int n_cams, const double* __restrict__ Hcc,
                                                                    const double* __restrict__ gc,
                                                                    double* __restrict__ S, int lda,
                                                                    double* __restrict__ rhs,
                                                                    double* __restrict__ ex_diag,
                                                                    double* __restrict__ ex_gc) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int c = gid / 36, k = gid - c * 36;
    if (c >= n_cams) return;
    const int a = k / 6, b = k - a * 6;
    const double h = Hcc[(size_t)c * 36 + k];
    if (b <= a) S[(size_t)(c * 6 + a) * lda + c * 6 + b] += h;
    if (a == b) {
        ex_diag[c * 6 + a] = h;
        const double g = gc[c * 6 + a];
        ex_gc[c * 6 + a] = g;
        rhs[c * 6 + a] -= g;
    }
}

// after the (optional) all-reduce: S_ii += dc_i (or 1 for constant dofs), rhs_i = 0 for constant
__global__ __launch_bounds__(256) void ba_reduced_damp_kernel(int n, const double* __restrict__ dc,
                                                              const unsigned char* __restrict__ cam_fixed,
                                                              double* __restrict__ S, int lda,
                                                              double* __restrict__ rhs) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = i / 6, a = i - c * 6;
    const bool fx = cam_fixed ? ((cam_fixed[c] >> a) & 1u) : false;
    if (fx) { S[(size_t)i * lda + i] += 1.0; rhs[i] = 0.0; }
    else S[(size_t)i * lda + i] += dc[i];
}

int launch_reduced_add_camera(int n_cams, const double* Hcc, const double* gc, double* S, int lda, double* rhs,
                              double* ex_diag, double* ex_gc, hipStream_t st) {
    hipLaunchKernelGGL(ba_reduced_add_camera_kernel, dim3((n_cams * 36 + 255) / 256), dim3(256), 0, st, n_cams,
                       Hcc, gc, S, lda, rhs, ex_diag, ex_gc);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

// One rank: the four small steps between the Schur complement and the factorisation in ONE launch --
// ba_reduced_add_camera_kernel (S += Hcc blocks, rhs -= gc, the diagonal and gc copied out), lm_diagonal_kernel on the
// camera diagonal, ba_reduced_damp_kernel and the padding of the factorisation (rows [n, lda): identity; the last row
// carries the right-hand side).  Same arithmetic in the same order as the four kernels (bit-identical); with several
// ranks the cross-rank sum sits between the first and the second step, and the separate kernels are used.
__global__ __launch_bounds__(256) void ba_reduced_finalize_kernel(int n_cams, int n, const double* __restrict__ Hcc,
                                                                  const double* __restrict__ gc, const unsigned char* __restrict__ cam_fixed,
                                                                  double* __restrict__ S, int lda, double* __restrict__ rhs,
                                                                  double* __restrict__ ex_diag, double* __restrict__ ex_gc,
                                                                  double* __restrict__ scale, int init_scale, int use_scaling,
                                                                  double radius, double dmin, double dmax, double* __restrict__ dc,
                                                                  int blocks_cam, const double* __restrict__ scalars, int n_scalars,
                                                                  double* __restrict__ host_out, int reduced) {
    // (reduced != 0 -- several ranks, behind the cross-rank sum: the camera blocks are in S already and diag(Hcc), gc come
    // summed over the ranks from ex_diag / ex_gc; what is left is the LM diagonal, the damping and the padding)
    // (host_out: the scalar slots and the gradient gc also go straight into mapped host memory -- [scalars | gc] -- where
    // the LM loop reads them one synchronisation later: export_linear_kernel's launch saved)
    if (host_out && blockIdx.x == 0 && (int)threadIdx.x < n_scalars) host_out[threadIdx.x] = scalars[threadIdx.x];
    if ((int)blockIdx.x >= blocks_cam) {
        // padding rows (chol_pad_kernel); the first n entries of the last row are written by the diagonal threads below
        const int r = n + ((int)blockIdx.x - blocks_cam);
        if (r >= lda) return;
        double* row = S + (size_t)r * lda;
        const bool last = (r == lda - 1);
        for (int c = threadIdx.x; c < lda; c += 256) {
            if (last && c < n) continue;
            row[c] = (c == r) ? 1.0 : 0.0;
        }
        return;
    }
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int c = gid / 36, k = gid - c * 36;
    if (c >= n_cams) return;
    const int a = k / 6, b = k - a * 6;
    const int i = c * 6 + a;
    if (reduced && a != b) return;
    const double h = reduced ? ex_diag[i] : Hcc[(size_t)c * 36 + k];
    if (a != b) {
        if (b < a) S[(size_t)(c * 6 + a) * lda + c * 6 + b] += h;
        return;
    }
    double sd = S[(size_t)i * lda + i] + (reduced ? 0.0 : h);
    const double g = reduced ? ex_gc[i] : gc[i];
    if (!reduced) { ex_diag[i] = h; ex_gc[i] = g; }
    if (host_out) host_out[n_scalars + i] = g;
    double rv = rhs[i] - (reduced ? 0.0 : g);
    // lm_diagonal_kernel, kind 2
    double sc = 1.0;
    if (use_scaling) {
        if (init_scale) { sc = 1.0 / (1.0 + sqrt(h)); scale[i] = sc; }
        else sc = scale[i];
    } else if (init_scale) scale[i] = 1.0;
    const double s2 = sc * sc;
    const double v = fmin(fmax(h * s2, dmin), dmax);
    const double d = v / radius / s2;
    dc[i] = d;
    // ba_reduced_damp_kernel
    const bool fx = cam_fixed ? ((cam_fixed[c] >> a) & 1u) : false;
    if (fx) { sd += 1.0; rv = 0.0; }
    else sd += d;
    S[(size_t)i * lda + i] = sd;
    rhs[i] = rv;
    S[(size_t)(lda - 1) * lda + i] = rv;
}

int launch_reduced_finalize(int n_cams, int n, const double* Hcc, const double* gc, const unsigned char* cam_fixed, double* S, int lda,
                            double* rhs, double* ex_diag, double* ex_gc, double* scale, int init_scale, int use_scaling,
                            double radius, double dmin, double dmax, double* dc, const double* scalars, int n_scalars,
                            double* host_out, int reduced, hipStream_t st) {
    const int blocks_cam = (n_cams * 36 + 255) / 256;
    hipLaunchKernelGGL(ba_reduced_finalize_kernel, dim3(blocks_cam + (lda - n)), dim3(256), 0, st, n_cams, n, Hcc, gc, cam_fixed, S, lda,
                       rhs, ex_diag, ex_gc, scale, init_scale, use_scaling, radius, dmin, dmax, dc, blocks_cam, scalars, n_scalars, host_out,
                       reduced);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

int launch_reduced_damp(int n, const double* dc, const unsigned char* cam_fixed, double* S, int lda, double* rhs,
                        hipStream_t st) {
    hipLaunchKernelGGL(ba_reduced_damp_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, dc, cam_fixed, S, lda,
                       rhs);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

// one camera of the manifold update: cams_new[c] = cams[c] (+) dxc[c] (constant dofs stay), v += {|step|^2, |x|^2, model term}
__device__ __forceinline__ void camera_update_lane(int c, const double* __restrict__ cams, const double* __restrict__ dxc,
                                                   const unsigned char* __restrict__ cam_fixed, const double* __restrict__ gc,
                                                   const double* __restrict__ dc, double* __restrict__ cams_new, double v[4]) {
    const unsigned cm = cam_fixed ? cam_fixed[c] : 0u;
    double d[6], q[4], qn[4];
#pragma unroll
    for (int a = 0; a < 6; ++a) d[a] = ((cm >> a) & 1u) ? 0.0 : dxc[c * 6 + a];
#pragma unroll
    for (int a = 0; a < 4; ++a) q[a] = cams[(size_t)c * 7 + a];
    so3_plus(q, d, qn);
    const bool rot_active = (cm & 7u) != 7u, pos_active = (cm & 56u) != 56u;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const double out = rot_active ? qn[a] : q[a];
        cams_new[(size_t)c * 7 + a] = out;
        if (rot_active) { v[0] += (out - q[a]) * (out - q[a]); v[1] += q[a] * q[a]; }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double t = cams[(size_t)c * 7 + 4 + a];
        cams_new[(size_t)c * 7 + 4 + a] = t + d[3 + a];
        if (pos_active) { v[0] += d[3 + a] * d[3 + a]; v[1] += t * t; }
    }
#pragma unroll
    for (int a = 0; a < 6; ++a)
        if (!((cm >> a) & 1u)) v[2] += -0.5 * gc[c * 6 + a] * d[a] + 0.5 * dc[c * 6 + a] * d[a] * d[a];
}

// ===========================================================================================
// back-substitution: dxp_j = Hinv_j ( -gp_j - sum_l Jp_l^T (Jc_l dxc[c_l]) )
// ===========================================================================================
// Same shape as ba_point_blocks_kernel: a workgroup owns PB_LM consecutive landmarks = one contiguous stretch of records.
// Pass 1 (a record per lane and trip, full-line loads): Jp^T (Jc dxc[cam]) of the record into LDS; pass 2 (an entry of dxp
// per lane): the landmark's records summed in observation order, then dxp = Hinv v through LDS.  With `up.pts_new` the
// landmark half of the manifold update and of the step statistics (ba_update_kernel) rides along: the trial point's
// landmarks and one partial[4] per workgroup, no second pass over dxp.
template <bool GEN>
__global__ __launch_bounds__(PB_THREADS) void ba_backsub_kernel(int n_pts, const int* __restrict__ pt_start,
                                                                const int* __restrict__ obs_cam,
                                                                const double* __restrict__ Jc, const unsigned char* __restrict__ Jp,
                                                                const double* __restrict__ Hinv6, const double* __restrict__ gp,
                                                                const double* __restrict__ dxc, double* __restrict__ dxp,
                                                                BacksubUpdate up, const double* __restrict__ Jc12) {
    __shared__ double u[3 * PB_LD];
    __shared__ int seg[PB_LM + 1];
    __shared__ double vv[3 * PB_LM];
    __shared__ double ssum[PB_THREADS / 64][3];
    const int t = threadIdx.x;
    const int gp_blocks = (n_pts + PB_LM - 1) / PB_LM;
    if ((int)blockIdx.x >= gp_blocks) {
        // the camera half of the update, PB_THREADS cameras per workgroup (only launched with up.cams_new)
        const int cblk = blockIdx.x - gp_blocks, c = cblk * PB_THREADS + t;
        double v[4] = {0.0, 0.0, 0.0, 0.0};
        if (c < up.n_cams) camera_update_lane(c, up.cams, dxc, up.cam_fixed, up.gc, up.dc, up.cams_new, v);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if ((t & 63) == 0) ssum[t >> 6][k] = x;
        }
        __syncthreads();
        if (t < 4) {
            double sv = 0.0;
            if (t < 3) for (int w = 0; w < PB_THREADS / 64; ++w) sv += ssum[w][t];
            up.partial_c[(size_t)cblk * 4 + t] = sv;
        }
        return;
    }
    const int j0 = blockIdx.x * PB_LM, nl = min(PB_LM, n_pts - j0);
    if (t <= nl) seg[t] = pt_start[j0 + t];
    __syncthreads();
    const int rb = seg[0], re = seg[nl];
    const int lj = t / 3, lk = t % 3;
    const bool mine = (t < 3 * PB_LM) && (lj < nl);
    double acc = mine ? -gp[(size_t)j0 * 3 + t] : 0.0;
    for (int cb = rb; cb < re; cb += PB_CHUNK) {
        const int ce = min(cb + PB_CHUNK, re);
#pragma unroll
        for (int s2 = 0; s2 < PB_TRIPS; ++s2) {
            const int x = t + PB_THREADS * s2, l = cb + x;
            if (l < ce) {
                const int c = obs_cam[l];
                double jc[12], jp[6];
                load_jc_jp<GEN>(Jc, Jp, l, jc, jp, Jc12);
                double m0 = 0.0, m1 = 0.0;
#pragma unroll
                for (int a = 0; a < 6; ++a) { const double d = dxc[c * 6 + a]; m0 += jc[a] * d; m1 += jc[6 + a] * d; }
                u[0 * PB_LD + x] = jp[0] * m0 + jp[3] * m1;
                u[1 * PB_LD + x] = jp[1] * m0 + jp[4] * m1;
                u[2 * PB_LD + x] = jp[2] * m0 + jp[5] * m1;
            }
        }
        __syncthreads();
        if (mine) {
            const int b0 = max(seg[lj], cb), e0 = min(seg[lj + 1], ce);
            const double* uk = u + lk * PB_LD - cb;
            for (int i = b0; i < e0; ++i) acc -= uk[i];
        }
        __syncthreads();
    }
    if (t < 3 * PB_LM) vv[t] = acc;
    __syncthreads();
    double st[3] = {0.0, 0.0, 0.0};
    if (mine) {
        const double* Hi = Hinv6 + (size_t)(j0 + lj) * 6;
        const double h0 = Hi[lk == 0 ? 0 : lk], h1 = Hi[lk == 0 ? 1 : (lk == 1 ? 3 : 4)], h2 = Hi[lk == 0 ? 2 : (lk == 1 ? 4 : 5)];
        const double d = h0 * vv[3 * lj] + h1 * vv[3 * lj + 1] + h2 * vv[3 * lj + 2];
        dxp[(size_t)j0 * 3 + t] = d;
        if (up.pts_new) {
            const bool fx = up.pt_fixed ? (up.pt_fixed[j0 + lj] != 0) : false;
            const double p = up.pts[(size_t)j0 * 3 + t];
            const double dd = fx ? 0.0 : d;
            up.pts_new[(size_t)j0 * 3 + t] = p + dd;
            if (!fx) {
                st[0] = dd * dd; st[1] = p * p;
                st[2] = -0.5 * gp[(size_t)j0 * 3 + t] * dd + 0.5 * up.dp[(size_t)j0 * 3 + t] * dd * dd;
            }
        }
    }
    if (up.pts_new) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double x = st[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if ((t & 63) == 0) ssum[t >> 6][k] = x;
        }
        __syncthreads();
        if (t < 4) {
            double sv = 0.0;
            if (t < 3) for (int w = 0; w < PB_THREADS / 64; ++w) sv += ssum[w][t];
            up.partial_p[(size_t)blockIdx.x * 4 + t] = sv;
        }
    }
}

int backsub_grid(int n_pts) { return (n_pts + PB_LM - 1) / PB_LM; }
int backsub_cam_grid(int n_cams) { return (n_cams + PB_THREADS - 1) / PB_THREADS; }

int launch_backsub(int n_pts, const int* pt_start, const int* obs_cam, const double* Jc, const unsigned char* Jp,
                   const double* Hinv6, const double* gp, const double* dxc, double* dxp, hipStream_t st, const BacksubUpdate* up,
                   const double* Jc12) {
    BacksubUpdate u0{};
    if (up) u0 = *up;
    const int grid = backsub_grid(n_pts) + (u0.cams_new ? backsub_cam_grid(u0.n_cams) : 0);
    if (grid > 0 && Jc12)
        hipLaunchKernelGGL(ba_backsub_kernel<true>, dim3(grid), dim3(PB_THREADS), 0, st, n_pts, pt_start, obs_cam, Jc,
                           Jp, Hinv6, gp, dxc, dxp, u0, Jc12);
    else if (grid > 0)
        hipLaunchKernelGGL(ba_backsub_kernel<false>, dim3(grid), dim3(PB_THREADS), 0, st, n_pts, pt_start, obs_cam, Jc,
                           Jp, Hinv6, gp, dxc, dxp, u0, (const double*)nullptr);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

// ===========================================================================================
// manifold update + step statistics.  partial[block][4] = { |x_new - x|^2, |x|^2 (current),
// model term sum(-1/2 g d + 1/2 D d^2), unused }
// ===========================================================================================
__device__ inline void block_sum4(double v[4], double* out) {
    __shared__ double s[4][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if (lane == 0) s[w][k] = x;
    }
    __syncthreads();
    if (threadIdx.x < 4) out[threadIdx.x] = s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x];
}

// both manifold updates in one launch: workgroups [0, cb) the cameras, [cb, cb + pb) the landmarks
__global__ __launch_bounds__(256) void ba_update_kernel(int n_cams, int n_pts, int cb, const double* __restrict__ cams, const double* __restrict__ pts,
                                                        const double* __restrict__ dxc, const double* __restrict__ dxp,
                                                        const unsigned char* __restrict__ cam_fixed, const unsigned char* __restrict__ pt_fixed,
                                                        const double* __restrict__ gc, const double* __restrict__ dc,
                                                        const double* __restrict__ gp, const double* __restrict__ dp,
                                                        double* __restrict__ cams_new, double* __restrict__ pts_new,
                                                        double* __restrict__ partial_c, double* __restrict__ partial_p) {
    double v[4] = {0, 0, 0, 0};
    if ((int)blockIdx.x < cb) {
        const int c = blockIdx.x * 256 + threadIdx.x;
        if (c < n_cams) camera_update_lane(c, cams, dxc, cam_fixed, gc, dc, cams_new, v);
        block_sum4(v, partial_c + (size_t)blockIdx.x * 4);
    } else {
        const int blk = blockIdx.x - cb;
        const int j = blk * 256 + threadIdx.x;
        if (j < n_pts) {
            const bool fx = pt_fixed ? (pt_fixed[j] != 0) : false;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double p = pts[(size_t)j * 3 + a];
                const double d = fx ? 0.0 : dxp[(size_t)j * 3 + a];
                pts_new[(size_t)j * 3 + a] = p + d;
                if (!fx) {
                    v[0] += d * d; v[1] += p * p;
                    v[2] += -0.5 * gp[(size_t)j * 3 + a] * d + 0.5 * dp[(size_t)j * 3 + a] * d * d;
                }
            }
        }
        block_sum4(v, partial_p + (size_t)blk * 4);
    }
}

int launch_update(int n_cams, int n_pts, const double* cams, const double* pts, const double* dxc,
                  const double* dxp, const unsigned char* cam_fixed, const unsigned char* pt_fixed,
                  const double* gc, const double* dc, const double* gp, const double* dp, double* cams_new,
                  double* pts_new, double* partial_c, double* partial_p, hipStream_t st) {
    const int cb = (n_cams + 255) / 256, pb = (n_pts + 255) / 256;
    hipLaunchKernelGGL(ba_update_kernel, dim3(cb + pb), dim3(256), 0, st, n_cams, n_pts, cb, cams, pts, dxc, dxp, cam_fixed, pt_fixed,
                       gc, dc, gp, dp, cams_new, pts_new, partial_c, partial_p);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

// ===========================================================================================
// per-landmark triangulation with cameras fixed (sim_data.h:165-194, sim_data.cpp:299-311):
// damped Gauss-Newton on each 3x3 system, one landmark per lane.
// ===========================================================================================
__global__ __launch_bounds__(256) void ba_triangulate_kernel(int n_pts, const int* __restrict__ pt_start,
                                                             const int* __restrict__ obs_cam,
                                                             const double2* __restrict__ feat,
                                                             const double* __restrict__ cams, double* __restrict__ pts,
                                                             const unsigned char* __restrict__ pt_fixed, int max_iter) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_pts) return;
    if (pt_fixed && pt_fixed[j]) return;
    const int b = pt_start[j], e = pt_start[j + 1];
    if (e - b < 1) return;
    double L[3] = {pts[(size_t)j * 3], pts[(size_t)j * 3 + 1], pts[(size_t)j * 3 + 2]};
    auto eval = [&](const double* P, double* H, double* g) -> double {
        double cost = 0.0;
        if (H) { for (int k = 0; k < 6; ++k) H[k] = 0.0; g[0] = g[1] = g[2] = 0.0; }
        for (int i = b; i < e; ++i) {
            const double* cam = cams + (size_t)obs_cam[i] * 7;
            double q[4] = {cam[0], cam[1], cam[2], cam[3]}, R[9];
            quat_to_rot(q, R);
            const double d0 = P[0] - cam[4], d1 = P[1] - cam[5], d2 = P[2] - cam[6];
            const double x = R[0] * d0 + R[3] * d1 + R[6] * d2;
            const double y = R[1] * d0 + R[4] * d1 + R[7] * d2;
            const double z = R[2] * d0 + R[5] * d1 + R[8] * d2;
            const double iz = 1.0 / z, xn = x * iz, yn = y * iz;
            const double2 f = feat[i];
            const double r0 = xn - f.x, r1 = yn - f.y;
            cost += r0 * r0 + r1 * r1;
            if (H) {
                double j0[3], j1[3];
                for (int k = 0; k < 3; ++k) {
                    j0[k] = iz * (R[k * 3 + 0] - xn * R[k * 3 + 2]);
                    j1[k] = iz * (R[k * 3 + 1] - yn * R[k * 3 + 2]);
                }
                H[0] += j0[0] * j0[0] + j1[0] * j1[0]; H[1] += j0[0] * j0[1] + j1[0] * j1[1];
                H[2] += j0[0] * j0[2] + j1[0] * j1[2]; H[3] += j0[1] * j0[1] + j1[1] * j1[1];
                H[4] += j0[1] * j0[2] + j1[1] * j1[2]; H[5] += j0[2] * j0[2] + j1[2] * j1[2];
                for (int k = 0; k < 3; ++k) g[k] -= j0[k] * r0 + j1[k] * r1;
            }
        }
        return cost;
    };
    double lambda = 1e-4;
    double H[6], g[3];
    double cost = eval(L, H, g);
    for (int it = 0; it < max_iter; ++it) {
        double Hd[6] = {H[0] + lambda * (H[0] + 1e-12), H[1], H[2], H[3] + lambda * (H[3] + 1e-12), H[4],
                        H[5] + lambda * (H[5] + 1e-12)};
        double Hi[6];
        if (!inv3_sym6(Hd, Hi)) break;
        const double d0 = Hi[0] * g[0] + Hi[1] * g[1] + Hi[2] * g[2];
        const double d1 = Hi[1] * g[0] + Hi[3] * g[1] + Hi[4] * g[2];
        const double d2 = Hi[2] * g[0] + Hi[4] * g[1] + Hi[5] * g[2];
        double Ln[3] = {L[0] + d0, L[1] + d1, L[2] + d2};
        const double nc = eval(Ln, nullptr, nullptr);
        if (nc < cost && nc == nc) {
            const double dn = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
            const double rel = (cost - nc) / (cost + 1e-300);
            L[0] = Ln[0]; L[1] = Ln[1]; L[2] = Ln[2];
            cost = eval(L, H, g);
            lambda = fmax(lambda * 0.1, 1e-12);
            if (dn < 1e-12 || rel < 1e-14) break;
        } else {
            lambda *= 10.0;
            if (lambda > 1e12) break;
        }
    }
    pts[(size_t)j * 3] = L[0]; pts[(size_t)j * 3 + 1] = L[1]; pts[(size_t)j * 3 + 2] = L[2];
}

int launch_triangulate(int n_pts, const int* pt_start, const int* obs_cam, const double2* feat, const double* cams,
                       double* pts, const unsigned char* pt_fixed, int max_iter, hipStream_t st) {
    hipLaunchKernelGGL(ba_triangulate_kernel, dim3((n_pts + 255) / 256), dim3(256), 0, st, n_pts, pt_start, obs_cam,
                       feat, cams, pts, pt_fixed, max_iter);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

// ===========================================================================================
// Zhang calibration factor (st3-calibration/src/src/calib.cpp:247-262 distortNormPt/normPt2ImgPt,
// :311-391 residual + Jacobian blocks), one chessboard corner per lane.
//   params = [alpha beta u0 v0 k1 k2 k3 p1 p2 | xi_0 .. xi_{V-1}], xi = se3 log [rho, theta]
//   e = predicted - measured pixel (calib.cpp:334)
//   Ji (2x9): d e / d(intrinsics, distortion)   (:337-348)
//   Jx (2x6): d e / d(left perturbation of the view pose), [I | -hat(P')] chain (:352-380)
// ===========================================================================================
// one corner of one view: residual e (2) and, if want_j, Ji (2x9) and Jx (2x6)
__device__ __forceinline__ void calib_corner(const double* __restrict__ params, int v, double X, double Y, double u, double w,
                                             bool want_j, double e[2], double ji[18], double jx[12]) {
    const double alpha = params[0], beta = params[1], u0 = params[2], v0 = params[3];
    const double k1 = params[4], k2 = params[5], k3 = params[6], p1 = params[7], p2 = params[8];
    double xi[6], R[9], t[3];
    for (int k = 0; k < 6; ++k) xi[k] = params[9 + v * 6 + k];
    se3_exp_rt(xi, R, t);
    const double Xp = R[0] * X + R[1] * Y + t[0], Yp = R[3] * X + R[4] * Y + t[1], Zp = R[6] * X + R[7] * Y + t[2];
    const double iz = 1.0 / Zp, xn = Xp * iz, yn = Yp * iz;
    const double r2 = xn * xn + yn * yn, r4 = r2 * r2, r6 = r4 * r2;
    const double rad = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
    const double xd = xn * rad + 2.0 * p1 * xn * yn + p2 * (r2 + 2.0 * xn * xn);
    const double yd = yn * rad + 2.0 * p2 * xn * yn + p1 * (r2 + 2.0 * yn * yn);
    e[0] = alpha * xd + u0 - u;
    e[1] = beta * yd + v0 - w;
    if (!want_j) return;
    const double jt[18] = {xd, 0, 1, 0, alpha * xn * r2, alpha * xn * r4, alpha * xn * r6, 2.0 * alpha * xn * yn,
                           alpha * (r2 + 2.0 * xn * xn),
                           0, yd, 0, 1, beta * yn * r2, beta * yn * r4, beta * yn * r6, beta * (r2 + 2.0 * yn * yn),
                           2.0 * beta * xn * yn};
    for (int k = 0; k < 18; ++k) ji[k] = jt[k];
    const double dx = 2.0 * k1 * xn + 4.0 * k2 * r2 * xn + 6.0 * k3 * r4 * xn;
    const double dy = 2.0 * k1 * yn + 4.0 * k2 * r2 * yn + 6.0 * k3 * r4 * yn;
    const double d00 = rad + xn * dx + 2.0 * p1 * yn + 6.0 * p2 * xn;
    const double d01 = xn * dy + 2.0 * p1 * xn + 2.0 * p2 * yn;
    const double d10 = yn * dx + 2.0 * p1 * xn + 2.0 * p2 * yn;
    const double d11 = rad + yn * dy + 2.0 * p2 * xn + 6.0 * p1 * yn;
    const double N[6] = {iz, 0, -Xp * iz * iz, 0, iz, -Yp * iz * iz};
    double M[6];
    for (int b = 0; b < 3; ++b) {
        M[b] = alpha * (d00 * N[b] + d01 * N[3 + b]);
        M[3 + b] = beta * (d10 * N[b] + d11 * N[3 + b]);
    }
    // [I | -hat(P')]:  -hat(P') = [[0, Zp, -Yp], [-Zp, 0, Xp], [Yp, -Xp, 0]]
    const double nH[9] = {0, Zp, -Yp, -Zp, 0, Xp, Yp, -Xp, 0};
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 3; ++b) {
            jx[a * 6 + b] = M[a * 3 + b];
            jx[a * 6 + 3 + b] = M[a * 3] * nH[b] + M[a * 3 + 1] * nH[3 + b] + M[a * 3 + 2] * nH[6 + b];
        }
}

__global__ __launch_bounds__(256) void calib_linearize_kernel(int n_views, int n_corners, const double* __restrict__ params,
                                                              const double* __restrict__ obj, const double* __restrict__ img,
                                                              double* __restrict__ e, double* __restrict__ Ji,
                                                              double* __restrict__ Jx, double* __restrict__ sse_partial) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    const int total = n_views * n_corners;
    double sq = 0.0;
    if (o < total) {
        double ee[2], ji[18], jx[12];
        calib_corner(params, o / n_corners, obj[(size_t)o * 2], obj[(size_t)o * 2 + 1], img[(size_t)o * 2], img[(size_t)o * 2 + 1],
                     Ji || Jx, ee, ji, jx);
        sq = ee[0] * ee[0] + ee[1] * ee[1];
        if (e) { e[(size_t)o * 2] = ee[0]; e[(size_t)o * 2 + 1] = ee[1]; }
        if (Ji) for (int k = 0; k < 18; ++k) Ji[(size_t)o * 18 + k] = ji[k];
        if (Jx) for (int k = 0; k < 12; ++k) Jx[(size_t)o * 12 + k] = jx[k];
    }
    __shared__ double s_red[4];
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) sse_partial[blockIdx.x] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// ---- Gauss-Newton with the ARROW structure of the calibration's normal equations --------------------------------
// The unknowns are the 9 intrinsics (shared by all corners) and one 6-dof pose per view (touched by that view's
// corners only): H = [A B; B^T C] with C block-diagonal.  The reference builds the dense (9+6V)^2 matrix from
// full-width Jacobian rows and calls ldlt() (calib.cpp:383-393); here neither exists:
//   calib_view_gram_kernel   per view: the Gram matrix of the corner rows [Ji (9) | Jx (6) | e], 16 x 16 symmetric =
//                            136 sums = {A_v, B_v, C_v, b_v, c_v, sse_v}; the rows never leave the workgroup's LDS;
//   calib_arrow_step_kernel  one workgroup: per view C_v = L L^T, Y_v = C_v^-1 B_v^T, z_v = C_v^-1 c_v; the 9 x 9 Schur
//                            complement S = sum A_v - B_v Y_v, its right-hand side, the intrinsics step; per view the
//                            pose step, the left-multiplicative pose update (calib.cpp:397-402); the iteration's record
//                            (sse, |step|, stop test).  Parameters, blocks and the iteration state stay on the device:
//                            the host enqueues max_iter iteration pairs and reads the result once.
// Every sum runs in a fixed order (bitwise reproducible).
// ================================================================================================================
constexpr int CALIB_GRAM = 136;          // pairs (p <= q) of the 16-vector [ji(9) | jx(6) | e]
constexpr int CALIB_CHUNK = 128;         // corners staged per pass

__global__ __launch_bounds__(256) void calib_view_gram_kernel(int n_corners, const double* __restrict__ params,
                                                              const double* __restrict__ obj, const double* __restrict__ img,
                                                              const int* __restrict__ state, double* __restrict__ gram) {
    if (state[1] != 0) return;                          // converged (or failed) earlier: the remaining launches are empty
    __shared__ double s_rows[CALIB_CHUNK][2][17];       // (padded: 17)
    const int v = blockIdx.x, t = threadIdx.x;
    // this thread's pair (p, q), p <= q, in row-major order of the upper triangle
    int p = 0, q = 0;
    if (t < CALIB_GRAM) {
        int k = t;
        while (k >= 16 - p) { k -= 16 - p; ++p; }
        q = p + k;
    }
    double acc = 0.0;
    for (int c0 = 0; c0 < n_corners; c0 += CALIB_CHUNK) {
        const int nc = min(CALIB_CHUNK, n_corners - c0);
        if (t < nc) {
            const size_t o = (size_t)v * n_corners + c0 + t;
            double e[2], ji[18], jx[12];
            calib_corner(params, v, obj[o * 2], obj[o * 2 + 1], img[o * 2], img[o * 2 + 1], true, e, ji, jx);
            for (int a = 0; a < 2; ++a) {
                for (int k = 0; k < 9; ++k) s_rows[t][a][k] = ji[a * 9 + k];
                for (int k = 0; k < 6; ++k) s_rows[t][a][9 + k] = jx[a * 6 + k];
                s_rows[t][a][15] = e[a];
            }
        }
        __syncthreads();
        if (t < CALIB_GRAM)
            for (int c = 0; c < nc; ++c) acc += s_rows[c][0][p] * s_rows[c][0][q] + s_rows[c][1][p] * s_rows[c][1][q];
        __syncthreads();
    }
    if (t < CALIB_GRAM) gram[(size_t)v * CALIB_GRAM + t] = acc;
}

__device__ __forceinline__ int calib_gram_index(int p, int q) {      // p <= q
    return p * 16 - p * (p - 1) / 2 + (q - p);
}

// state: {iterations completed, done, failed pivot (1-based unknown) }; scratch: per view Y (6x9) | z (6) | B Y (45) | B z (9)
__global__ __launch_bounds__(256) void calib_arrow_step_kernel(int n_views, const double* __restrict__ gram, double* __restrict__ params,
                                                               double* __restrict__ scratch, int* __restrict__ state,
                                                               double* __restrict__ sse_trace) {
    if (state[1] != 0) return;
    constexpr int SCR = 54 + 6 + 45 + 9;
    __shared__ double s_S[45], s_rhs[9], s_di[9], s_un[256], s_sse;
    __shared__ int s_fail;
    const int t = threadIdx.x;
    for (int v = t; v < n_views; v += 256) {
        const double* G = gram + (size_t)v * CALIB_GRAM;
        double* sc = scratch + (size_t)v * SCR;
        // C_v = L L^T (6 x 6, unknowns 9..14 of the 16-vector)
        double L[21];
        bool ok = true;
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j <= i; ++j) {
                double sum = G[calib_gram_index(9 + j, 9 + i)];
                for (int k = 0; k < j; ++k) sum -= L[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
                if (i == j) { if (!(sum > 0.0)) { ok = false; sum = 1.0; } L[i * (i + 1) / 2 + i] = sqrt(sum); }
                else L[i * (i + 1) / 2 + j] = sum / L[j * (j + 1) / 2 + j];
            }
        if (!ok) atomicCAS(&state[2], 0, 9 + 6 * v + 1);
        auto solve6 = [&](double* x) {                 // x <- C_v^-1 x
            for (int i = 0; i < 6; ++i) {
                double sum = x[i];
                for (int k = 0; k < i; ++k) sum -= L[i * (i + 1) / 2 + k] * x[k];
                x[i] = sum / L[i * (i + 1) / 2 + i];
            }
            for (int i = 5; i >= 0; --i) {
                double sum = x[i];
                for (int k = i + 1; k < 6; ++k) sum -= L[k * (k + 1) / 2 + i] * x[k];
                x[i] = sum / L[i * (i + 1) / 2 + i];
            }
        };
        double Y[54];                                   // Y[k][a] = (C^-1 B^T)[k][a], k < 6, a < 9
        for (int a = 0; a < 9; ++a) {
            double x[6];
            for (int k = 0; k < 6; ++k) x[k] = G[calib_gram_index(a, 9 + k)];
            solve6(x);
            for (int k = 0; k < 6; ++k) Y[k * 9 + a] = x[k];
        }
        double z[6];
        for (int k = 0; k < 6; ++k) z[k] = G[calib_gram_index(9 + k, 15)];
        solve6(z);
        for (int k = 0; k < 54; ++k) sc[k] = Y[k];
        for (int k = 0; k < 6; ++k) sc[54 + k] = z[k];
        int idx = 0;
        for (int a = 0; a < 9; ++a)
            for (int b = a; b < 9; ++b, ++idx) {
                double sum = 0.0;
                for (int k = 0; k < 6; ++k) sum += G[calib_gram_index(a, 9 + k)] * Y[k * 9 + b];
                sc[60 + idx] = sum;
            }
        for (int a = 0; a < 9; ++a) {
            double sum = 0.0;
            for (int k = 0; k < 6; ++k) sum += G[calib_gram_index(a, 9 + k)] * z[k];
            sc[105 + a] = sum;
        }
    }
    __syncthreads();
    // Schur complement of the pose blocks and its right-hand side: S = sum_v (A_v - B_v Y_v), rhs = -(sum_v b_v - B_v z_v)
    if (t < 45) {
        int a = 0, k = t;
        while (k >= 9 - a) { k -= 9 - a; ++a; }
        const int b = a + k;
        double sum = 0.0;
        for (int v = 0; v < n_views; ++v) sum += gram[(size_t)v * CALIB_GRAM + calib_gram_index(a, b)] - scratch[(size_t)v * SCR + 60 + t];
        s_S[t] = sum;
    } else if (t < 54) {
        const int a = t - 45;
        double sum = 0.0;
        for (int v = 0; v < n_views; ++v) sum += gram[(size_t)v * CALIB_GRAM + calib_gram_index(a, 15)] - scratch[(size_t)v * SCR + 105 + a];
        s_rhs[a] = -sum;
    } else if (t == 54) {
        double sum = 0.0;
        for (int v = 0; v < n_views; ++v) sum += gram[(size_t)v * CALIB_GRAM + calib_gram_index(15, 15)];
        s_sse = sum;
    }
    __syncthreads();
    if (t == 0) {                                       // 9 x 9 Cholesky solve
        double L[45], x[9];
        auto at = [](int i, int j) { return j * 9 - j * (j - 1) / 2 + (i - j); };      // (i >= j) in the upper-triangle order of s_S
        bool ok = true;
        for (int i = 0; i < 9; ++i)
            for (int j = 0; j <= i; ++j) {
                double sum = s_S[at(i, j)];
                for (int k = 0; k < j; ++k) sum -= L[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
                if (i == j) { if (!(sum > 0.0)) { if (ok) atomicCAS(&state[2], 0, i + 1); ok = false; sum = 1.0; } L[i * (i + 1) / 2 + i] = sqrt(sum); }
                else L[i * (i + 1) / 2 + j] = sum / L[j * (j + 1) / 2 + j];
            }
        for (int i = 0; i < 9; ++i) {
            double sum = s_rhs[i];
            for (int k = 0; k < i; ++k) sum -= L[i * (i + 1) / 2 + k] * x[k];
            x[i] = sum / L[i * (i + 1) / 2 + i];
        }
        for (int i = 8; i >= 0; --i) {
            double sum = x[i];
            for (int k = i + 1; k < 9; ++k) sum -= L[k * (k + 1) / 2 + i] * x[k];
            x[i] = sum / L[i * (i + 1) / 2 + i];
        }
        for (int i = 0; i < 9; ++i) s_di[i] = x[i];
        s_fail = atomicAdd(&state[2], 0);               // a failed pivot of any view block or of the 9 x 9 complement
    }
    __syncthreads();
    if (s_fail != 0) {                                  // the parameters stay as they were (the caller gets them back untouched)
        if (t == 0) { sse_trace[state[0]] = s_sse; state[1] = 2; }
        return;
    }
    // pose steps dx_v = -z_v - Y_v di, left-multiplicative update
    double un = 0.0;
    for (int v = t; v < n_views; v += 256) {
        const double* sc = scratch + (size_t)v * SCR;
        double d[6];
        for (int k = 0; k < 6; ++k) {
            double sum = -sc[54 + k];
            for (int a = 0; a < 9; ++a) sum -= sc[k * 9 + a] * s_di[a];
            d[k] = sum;
            un += sum * sum;
        }
        double xi[6];
        for (int k = 0; k < 6; ++k) xi[k] = params[9 + v * 6 + k];
        se3_left_update(d, xi);
        for (int k = 0; k < 6; ++k) params[9 + v * 6 + k] = xi[k];
    }
    s_un[t] = un;
    __syncthreads();
    if (t == 0) {
        double tot = 0.0;
        for (int k = 0; k < 256; ++k) tot += s_un[k];
        for (int a = 0; a < 9; ++a) { tot += s_di[a] * s_di[a]; params[a] += s_di[a]; }     // calib.cpp:394
        const int it = state[0];
        sse_trace[it] = s_sse;
        if (sqrt(tot) < 1e-8) state[1] = 1;        // calib.cpp:404 (the iteration counter is not advanced on the break)
        else state[0] = it + 1;
    }
}

int launch_calib_arrow_iteration(int n_views, int n_corners, double* params, const double* obj, const double* img, double* gram,
                                 double* scratch, int* state, double* sse_trace, hipStream_t st) {
    hipLaunchKernelGGL(calib_view_gram_kernel, dim3(n_views), dim3(256), 0, st, n_corners, params, obj, img, state, gram);
    hipLaunchKernelGGL(calib_arrow_step_kernel, dim3(1), dim3(256), 0, st, n_views, gram, params, scratch, state, sse_trace);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

int launch_calib_linearize(int n_views, int n_corners, const double* params, const double* obj, const double* img,
                           double* e, double* Ji, double* Jx, double* sse_partial, hipStream_t st) {
    const int total = n_views * n_corners;
    hipLaunchKernelGGL(calib_linearize_kernel, dim3((total + 255) / 256), dim3(256), 0, st, n_views, n_corners, params, obj,
                       img, e, Ji, Jx, sse_partial);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

// ===========================================================================================
// dense normal equations for the small problems: H = J^T J (n x n, n <= 256), g = J^T r
// one workgroup per (a, b-chunk): plain, these problems are tiny (6..129 unknowns)
// ===========================================================================================
__global__ __launch_bounds__(256) void dense_normal_kernel(int n_res, int n, const double* __restrict__ J,
                                                           const double* __restrict__ r, double* __restrict__ H,
                                                           int ldh, double* __restrict__ g) {
    // block (a): computes row a of H (lower part) and g[a]
    const int a = blockIdx.x;
    __shared__ double s[256];
    for (int b = 0; b <= a + 1; ++b) {     // b == a+1 -> gradient
        double v = 0.0;
        for (int i = threadIdx.x; i < n_res; i += 256) {
            const double ja = J[(size_t)i * n + a];
            v += ja * ((b <= a) ? J[(size_t)i * n + b] : r[i]);
        }
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            if (b <= a) H[(size_t)a * ldh + b] = s[0];
            else g[a] = s[0];
        }
        __syncthreads();
    }
}

int launch_dense_normal(int n_res, int n, const double* J, const double* r, double* H, int ldh, double* g,
                        hipStream_t st) {
    hipLaunchKernelGGL(dense_normal_kernel, dim3(n), dim3(256), 0, st, n_res, n, J, r, H, ldh, g);
    STBA_HIP(hipGetLastError());
    return STBA_OK;
}

}  // namespace stba

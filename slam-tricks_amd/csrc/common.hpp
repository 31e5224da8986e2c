// common.hpp -- shared host/device helpers for the gfx950 NLS engine.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/stba.h"

namespace stba {

// ---- error plumbing: no exceptions cross the C ABI -----------------------------------------
extern thread_local std::string g_last_error;
inline int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
#define STBA_HIP(call)                                                                       \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            return ::stba::fail((e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice)      \
                                    ? STBA_ERR_NO_DEVICE : STBA_ERR_HIP,                     \
                                std::string(#call) + ": " + hipGetErrorString(e_));          \
        }                                                                                    \
    } while (0)
#define STBA_TRY(call)                     \
    do {                                   \
        int s_ = (call);                   \
        if (s_ != STBA_OK) return s_;      \
    } while (0)

int require_device();   // STBA_OK or STBA_ERR_NO_DEVICE (there is no CPU fallback)

// Experiment knobs (environment variables) exist only in builds with -DSTBA_DEBUG_KNOBS (STBA_DEBUG_KNOBS=1 in the
// environment of slam-tricks_amd/build.py); the product library reads no environment variables for its schedules.
#ifdef STBA_DEBUG_KNOBS
inline int knob_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
inline double knob_double(const char* name, double dflt) { const char* e = getenv(name); return e ? atof(e) : dflt; }
inline const char* knob_str(const char* name) { return getenv(name); }
#else
inline int knob_int(const char*, int dflt) { return dflt; }
inline double knob_double(const char*, double dflt) { return dflt; }
inline const char* knob_str(const char*) { return nullptr; }
#endif

// runs fn once per DEVICE (function attributes such as the dynamic-LDS limit are per-device state), thread-safe
struct DeviceOnce {
    std::mutex m;
    unsigned long long done = 0;
    template <class F> int run(F fn) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        std::lock_guard<std::mutex> g(m);
        if (dev >= 0 && dev < 64 && ((done >> dev) & 1ull)) return STBA_OK;
        const int rc = fn();
        if (rc == STBA_OK && dev >= 0 && dev < 64) done |= 1ull << dev;
        return rc;
    }
};

// ---- stamped blocks: a kernel hands a few doubles to the POLLING host through pinned, mapped host memory ------------------
// (no event, no copy, no synchronise: an event record costs ~5 us of idle GPU, a copy ~30.)  The obvious protocol -- the payload,
// __threadfence_system(), a sequence number behind it -- is NOT enough on this platform: MEASURED in round 6 (tools/dbg/tri_repeat.py,
// profiles/r6_tri_repeat*.txt), the host saw the new sequence number and still read the OLD payload from ANOTHER cache line once in
// ~50 000 hand-offs (600 tiny Solve() calls: 23 of 300 runs ended somewhere else; 0 of 300 with what follows), whatever fences the
// kernel used.  So every 64-byte line validates ITSELF: six payload doubles, a CHECK word (a multiplicative hash of their bit patterns
// and the stamp's -- not their xor: the old and the new payload of a line are related numbers, and equal changes in two words cancel
// in an xor) and the stamp.  The host accepts a line only if its stamp is the one it waits for AND the check word fits what it has read;
// a line caught half-written fails the check and is read again.  No assumption about the order or the atomicity of the device's stores.
constexpr int STAMPED_PAYLOAD = 6;                       // payload doubles per 64-byte line
constexpr unsigned long long STAMPED_SALT = 0x9E3779B97F4A7C15ull;
__host__ __device__ inline unsigned long long stamped_mix(unsigned long long h, unsigned long long w) {
    h = (h ^ w) * 0x100000001B3ull;          // (FNV-1a step on a 64-bit word, then the high bits folded down)
    return h ^ (h >> 29);
}
__host__ __device__ inline int stamped_lines(int n_payload) { return (n_payload + STAMPED_PAYLOAD - 1) / STAMPED_PAYLOAD; }
__host__ __device__ inline int stamped_doubles(int n_payload) { return 8 * stamped_lines(n_payload); }     // the block's size (64-byte aligned)
#ifdef __HIPCC__
__device__ inline double stamped_word(const double* src, int n_payload, double stamp, int l) {
    const int line = l >> 3, w = l & 7;
    if (w < STAMPED_PAYLOAD) { const int pi = line * STAMPED_PAYLOAD + w; return pi < n_payload ? src[pi] : 0.0; }
    if (w == 7) return stamp;
    unsigned long long chk = stamped_mix(STAMPED_SALT, (unsigned long long)__double_as_longlong(stamp));
    for (int q = 0; q < STAMPED_PAYLOAD; ++q) {
        const int pi = line * STAMPED_PAYLOAD + q;
        chk = stamped_mix(chk, (unsigned long long)__double_as_longlong(pi < n_payload ? src[pi] : 0.0));
    }
    return __longlong_as_double((long long)chk);
}
// all 64 lanes of ONE wave (lane = 0 .. 63): a line is written by eight neighbouring lanes of one store instruction.
// src: the payload where every lane can read it (LDS or global memory, complete before the call)
__device__ inline void stamped_store_wave(double* out, const double* src, int n_payload, double stamp, int lane) {
    const int total = stamped_doubles(n_payload);
    for (int l = lane; l < total; l += 64) out[l] = stamped_word(src, n_payload, stamp, l);
    __threadfence_system();
}
// ONE thread (a block of a line or two)
__device__ inline void stamped_store_thread(double* out, const double* src, int n_payload, double stamp) {
    const int total = stamped_doubles(n_payload);
    for (int l = 0; l < total; ++l) out[l] = stamped_word(src, n_payload, stamp, l);
    __threadfence_system();
}
#endif
// host: one look at the block.  true: every line carries a stamp that `accept` takes (the same one in every line) and a fitting
// check word; payload[0 .. n_payload) and *stamp_out are then what the device wrote for that stamp
template <class Accept>
inline bool stamped_try_read(const volatile double* in, int n_payload, Accept accept, double* payload, double* stamp_out = nullptr) {
    const int nl = stamped_lines(n_payload);
    double first_stamp = 0.0;
    for (int L = 0; L < nl; ++L) {
        unsigned long long w[8];
        const volatile unsigned long long* src = reinterpret_cast<const volatile unsigned long long*>(in + 8 * L);
        for (int q = 7; q >= 0; --q) w[q] = src[q];          // (the stamp first)
        double st;
        std::memcpy(&st, &w[7], sizeof st);
        if (!accept(st) || (L > 0 && st != first_stamp)) return false;
        first_stamp = st;
        unsigned long long chk = stamped_mix(STAMPED_SALT, w[7]);
        for (int q = 0; q < STAMPED_PAYLOAD; ++q) chk = stamped_mix(chk, w[q]);
        if (chk != w[6]) return false;
        for (int q = 0; q < STAMPED_PAYLOAD; ++q) {
            const int pi = L * STAMPED_PAYLOAD + q;
            if (pi < n_payload) std::memcpy(&payload[pi], &w[q], sizeof(double));
        }
    }
    if (stamp_out) *stamp_out = first_stamp;
    return true;
}

// ---- dense Cholesky on the device (dense_chol.hip) -----------------------------------------
constexpr int CHOL_NB = 128;
// padded order: multiple of CHOL_NB with at least one spare row (the last row carries the rhs)
inline int chol_padded_dim(int n) { return ((n + 1 + CHOL_NB - 1) / CHOL_NB) * CHOL_NB; }
// A_dev: lda x lda row-major, lda = chol_padded_dim(n).  Rows/cols [n, lda-1) must hold the
// identity, row lda-1 holds rhs^T in columns [0, n) (and 1 on its diagonal).  On return the
// lower triangle holds L, row lda-1 holds y = L^-1 rhs, x_dev[0..lda) the solution of A x = rhs.
// flag_dev: int, set to (row+1) of the first non-positive pivot among real rows, or to
// CHOL_FLAG_TIMEOUT if the persistent kernel gave up waiting for a dependency (a bug or a lost
// workgroup; callers turn it into STBA_ERR_HIP through chol_flag_status).
constexpr int CHOL_FLAG_TIMEOUT = -2147483647;
inline int chol_flag_status(int flag_h) {
    return flag_h == CHOL_FLAG_TIMEOUT ? fail(STBA_ERR_HIP, "dense Cholesky: the persistent program timed out waiting for a dependency and the caller has no way to rebuild the matrix") : STBA_OK;
}
int chol_factor_solve_dev(double* A_dev, int lda, int n, double* x_dev, int* flag_dev, hipStream_t st);
// the same factorisation + solve through the stage kernels (one launch per stage and panel): the fallback when the
// persistent program reports CHOL_FLAG_TIMEOUT -- the matrix must be rebuilt first, the aborted run leaves it half factored
int chol_factor_solve_stages(double* A_dev, int lda, int n, double* x_dev, int* flag_dev, hipStream_t st);
// a caller that has SEEN CHOL_FLAG_TIMEOUT reports it: the device's next 64 factorisations take the stage kernels without
// trying the persistent program again (the device is shared with somebody), and the process-wide count goes up
void chol_forget_stream(hipStream_t st);   // before a stream that ran factorisations is destroyed (after synchronising it)
void chol_note_timeout();
void chol_note_peer_timeout();
void chol_count_timeout();
void chol_set_spin_limit_us(double us);    // how long a workgroup waits for a dependency before it gives up; 0: automatic
int chol_timeout_count();

// the production schedule with an event recorded between the factorisation and the backward substitution
int chol_factor_solve_split(double* A_dev, int lda, int n, double* x_dev, int* flag_dev, hipStream_t st, hipEvent_t mid_event,
                            hipEvent_t pre_event = nullptr);   // pre_event: recorded right in front of the persistent kernel (stage schedule: never)
struct CholProfile {
    double ms_diag, ms_trsm, ms_syrk, ms_bwd;
    double syrk_flops;          // algorithmic: sum over steps of m(m+1)*128, m = remaining real rows
    double syrk_flops_padded;   // what the tiles actually execute
    int syrk_launches;
};
int chol_factor_solve_profiled(double* A_dev, int lda, int n, double* x_dev, int* flag_dev, hipStream_t st,
                               CholProfile* prof);
// host-side scheduling model of the persistent factorisation kernel: predicted makespan (us) for nblk block columns
double chol_schedule_makespan(int nblk, int nq, int wg_per_q);
void chol_shard_model(int nblk, int n_gpus, int n_xcd, int wg_per_q, int rows_per_group, double hop_us, double tile_us, double* out3);
int chol_shard_row_owner(int row, int n_gpus, int rows_per_group);
// fills the padding (identity) and the rhs row of a padded system
int chol_prepare_padding_dev(double* A_dev, int lda, int n, const double* rhs_dev, hipStream_t st);
// explicit inverse of a small SPD matrix (pose graph coarse operator): W = [A . ; I 0] (2 np x 2 np, np a multiple of 128)
// -> lower right block = -A^-1 (lower triangle); dense_chol.hip
// lower triangle of S = -(Y Y^T) on the matrix cores (the Schur complement of a bundle adjustment with dense visibility):
// Y is lda x kcols (kcols a multiple of 16, leading dimension ldy), S lda x lda (lda a multiple of 128); ws: workspace of
// chol_yyt_workspace_doubles doubles (0: none needed)
size_t chol_yyt_workspace_doubles(int lda, size_t kcols);
int chol_yyt_lower_dev(const double* Y, size_t ldy, size_t kcols, double* S, int lda, double* ws, hipStream_t st);
size_t chol_spd_inverse_workspace_doubles(int np);
int chol_spd_inverse_dev(double* W, int ldw, int np, int n_real, int* flag_dev, double* work, hipStream_t st);

// ---- small device math ----------------------------------------------------------------------
__host__ __device__ inline void quat_to_rot(const double q[4], double R[9]) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double xx = x * x, yy = y * y, zz = z * z;
    const double xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    R[0] = 1 - 2 * (yy + zz); R[1] = 2 * (xy - wz);     R[2] = 2 * (xz + wy);
    R[3] = 2 * (xy + wz);     R[4] = 1 - 2 * (xx + zz); R[5] = 2 * (yz - wx);
    R[6] = 2 * (xz - wy);     R[7] = 2 * (yz + wx);     R[8] = 1 - 2 * (xx + yy);
}

// Sophus SO3::exp (quaternion x,y,z,w)
__host__ __device__ inline void so3_exp(const double w[3], double q[4]) {
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double imag, real;
    if (th2 < 1e-20) {
        const double th4 = th2 * th2;
        imag = 0.5 - th2 / 48.0 + th4 / 3840.0;
        real = 1.0 - th2 / 8.0 + th4 / 384.0;
    } else {
        const double th = sqrt(th2);
        imag = sin(0.5 * th) / th;
        real = cos(0.5 * th);
    }
    q[0] = imag * w[0]; q[1] = imag * w[1]; q[2] = imag * w[2]; q[3] = real;
}

// q (x) exp(d), renormalised: LieLocalParameterization<SO3d>::Plus (solver.hpp:38-45)
__host__ __device__ inline void so3_plus(const double q[4], const double d[3], double out[4]) {
    double e[4];
    so3_exp(d, e);
    const double ax = q[0], ay = q[1], az = q[2], aw = q[3];
    const double bx = e[0], by = e[1], bz = e[2], bw = e[3];
    const double ox = aw * bx + ax * bw + ay * bz - az * by;
    const double oy = aw * by - ax * bz + ay * bw + az * bx;
    const double oz = aw * bz + ax * by - ay * bx + az * bw;
    const double ow = aw * bw - ax * bx - ay * by - az * bz;
    const double n = sqrt(ox * ox + oy * oy + oz * oz + ow * ow);
    out[0] = ox / n; out[1] = oy / n; out[2] = oz / n; out[3] = ow / n;
}

// Sophus SE3::exp for the tangent [rho, theta]: rotation matrix + translation V(theta) rho
__host__ __device__ inline void se3_exp_rt(const double xi[6], double R[9], double t[3]) {
    double q[4];
    so3_exp(xi + 3, q);
    quat_to_rot(q, R);
    const double w0 = xi[3], w1 = xi[4], w2 = xi[5];
    const double th2 = w0 * w0 + w1 * w1 + w2 * w2;
    double a, b;
    if (th2 < 1e-20) { a = 0.5 - th2 / 24.0; b = 1.0 / 6.0 - th2 / 120.0; }
    else { const double th = sqrt(th2); a = (1.0 - cos(th)) / th2; b = (th - sin(th)) / (th2 * th); }
    // V = I + a K + b K^2,  K = hat(w)
    const double K[9] = {0, -w2, w1, w2, 0, -w0, -w1, w0, 0};
    double K2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) K2[i * 3 + j] = K[i * 3] * K[j] + K[i * 3 + 1] * K[3 + j] + K[i * 3 + 2] * K[6 + j];
    for (int i = 0; i < 3; ++i) {
        double s = 0.0;
        for (int j = 0; j < 3; ++j) s += ((i == j ? 1.0 : 0.0) + a * K[i * 3 + j] + b * K2[i * 3 + j]) * xi[j];
        t[i] = s;
    }
}

// Sophus-style SE3 helpers for the left-multiplicative pose update of the calibration (calib.cpp:397-402)
__host__ __device__ inline void quat_mul(const double* a, const double* b, double* o) {
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
__host__ __device__ inline void so3_log(const double* q, double* w) {
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
    double k;
    if (n2 < 1e-20) k = 2.0 / q[3] - (2.0 / 3.0) * n2 / (q[3] * q[3] * q[3]);
    else { const double nn = sqrt(n2); k = 2.0 * ((q[3] < 0) ? atan2(-nn, -q[3]) : atan2(nn, q[3])) / nn; }
    w[0] = k * q[0]; w[1] = k * q[1]; w[2] = k * q[2];
}
__host__ __device__ inline void so3_left_jacobian(const double* w, double* Vm) {
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double a, b;
    if (th2 < 1e-20) { a = 0.5 - th2 / 24.0; b = 1.0 / 6.0 - th2 / 120.0; }
    else { const double th = sqrt(th2); a = (1.0 - cos(th)) / th2; b = (th - sin(th)) / (th2 * th); }
    const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const double k2 = K[i * 3] * K[j] + K[i * 3 + 1] * K[3 + j] + K[i * 3 + 2] * K[6 + j];
            Vm[i * 3 + j] = (i == j ? 1.0 : 0.0) + a * K[i * 3 + j] + b * k2;
        }
}
// xi <- log(exp(d) * exp(xi))
__host__ __device__ inline void se3_left_update(const double* d, double* xi) {
    double qa[4], qb[4], qc[4], Ra[9], ta[3], Rb[9], tb[3], tc[3];
    so3_exp(d + 3, qa); so3_exp(xi + 3, qb);
    se3_exp_rt(d, Ra, ta); se3_exp_rt(xi, Rb, tb);
    quat_mul(qa, qb, qc);
    const double nn = sqrt(qc[0] * qc[0] + qc[1] * qc[1] + qc[2] * qc[2] + qc[3] * qc[3]);
    for (int k = 0; k < 4; ++k) qc[k] /= nn;
    for (int i = 0; i < 3; ++i) tc[i] = Ra[i * 3] * tb[0] + Ra[i * 3 + 1] * tb[1] + Ra[i * 3 + 2] * tb[2] + ta[i];
    double w[3], Vm[9];
    so3_log(qc, w);
    so3_left_jacobian(w, Vm);
    // rho = V^-1 t (3x3 solve by Cramer)
    const double a = Vm[0], b = Vm[1], c = Vm[2], dd = Vm[3], ee = Vm[4], f = Vm[5], g = Vm[6], h = Vm[7], i9 = Vm[8];
    const double C0 = ee * i9 - f * h, C1 = f * g - dd * i9, C2 = dd * h - ee * g;
    const double inv = 1.0 / (a * C0 + b * C1 + c * C2);
    xi[0] = inv * (C0 * tc[0] + (c * h - b * i9) * tc[1] + (b * f - c * ee) * tc[2]);
    xi[1] = inv * (C1 * tc[0] + (a * i9 - c * g) * tc[1] + (c * dd - a * f) * tc[2]);
    xi[2] = inv * (C2 * tc[0] + (b * g - a * h) * tc[1] + (a * ee - b * dd) * tc[2]);
    xi[3] = w[0]; xi[4] = w[1]; xi[5] = w[2];
}

// inverse of the symmetric 3x3 given as (xx,xy,xz,yy,yz,zz); returns false if not SPD-ish
__host__ __device__ inline bool inv3_sym6(const double A[6], double Ai[6]) {
    const double a = A[0], b = A[1], c = A[2], d = A[3], e = A[4], f = A[5];
    const double C00 = d * f - e * e, C01 = c * e - b * f, C02 = b * e - c * d;
    const double det = a * C00 + b * C01 + c * C02;
    if (!(det > 0.0) || !(fabs(det) < 1e300)) return false;
    const double inv = 1.0 / det;
    Ai[0] = C00 * inv; Ai[1] = C01 * inv; Ai[2] = C02 * inv;
    Ai[3] = (a * f - c * c) * inv; Ai[4] = (b * c - a * e) * inv; Ai[5] = (a * d - b * b) * inv;
    return true;
}

}  // namespace stba

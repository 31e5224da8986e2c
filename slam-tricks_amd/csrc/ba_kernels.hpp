// ba_kernels.hpp -- launch wrappers of the BA hot-path kernels (definitions in ba_kernels.hip)
#pragma once
#include "common.hpp"

namespace stba {

constexpr int LIN_THREADS = 1024;              // observations per workgroup tile
constexpr int LIN_MAX_LDS = 160 * 1024 - 512;  // dynamic LDS budget of the linearize kernel
constexpr int CAM_CHUNK = 256;                 // observations per camera-side reduction chunk

struct LinArgs {
    int n_obs, n_cams;
    const double* cams;              // [n_cams][7]
    const double* pts;               // [n_pts][3]
    const double2* feat;             // [n_obs]
    const int* obs_cam;
    const int* obs_pt;
    const unsigned char* cam_fixed;  // [n_cams] bitmask (bit a = dof a constant) or null
    const unsigned char* pt_fixed;   // [n_pts] or null
    double2* r;                      // [n_obs]
    double* J8;                      // [n_obs][8] compact Jacobian {xn, yn, P 2x3} (see ba_kernels.hip)
    double* cost_partial;            // [grid]
};

size_t lin_lds_bytes(int n_cams, bool cams_in_lds, bool with_jac);
int launch_linearize(const LinArgs& a, bool with_jac, int grid, hipStream_t st);
int launch_sum_partials(const double* partial, int n, int stride, int K, double* out, hipStream_t st);
int launch_absmax(const double* v, size_t n, const double* v2, size_t n2, double* out, double* partial, int n_partial,
                  hipStream_t st);
// (J8: compact Jacobian [n_obs][8]; omask: per-observation mask byte of constant dofs / landmarks, or null)
int launch_expand_jacobian(int n_obs, const double* J8, const unsigned char* omask, double* Jc, double* Jp, hipStream_t st);
// (gpmax_partial: optional, one max |gp| per workgroup of 256 landmarks, finished by launch_linear_finish)
int point_blocks_grid(int n_pts);      // workgroups of the landmark-block kernel = entries of its gpmax partial array
int launch_point_blocks(int n_pts, const int* pt_start, const double* J8, const unsigned char* omask, const double2* r,
                        double* Hpp6, double* gp, double* gpmax_partial, hipStream_t st);
int launch_linear_finish(const double* cost_partial, int n_cost, const double* gpmax_partial, int n_gp, double* cost2_out,
                         double* slots, int n_slots, int cost_slot, int max_slot, hipStream_t st);
int launch_trial_finish(const double* cost_partial, int n_cost, const double* part_p, int n_p, const double* part_c, int n_c,
                        const int* flag, double* out, double* host_out, double host_seq, hipStream_t st);
int launch_camera_blocks(int n_cams, int n_chunks, const int* chunk_begin, const int* chunk_end,
                         const int* cam_chunk_start, const int* cam_perm, const double* J8, const unsigned char* omask,
                         const double* Jc12, const double2* r, double* partial, double* Hcc, double* gc, hipStream_t st);
int launch_lm_diagonal(int n, int bs, int bstride, int kind, const double* H, double* scale, int init_scale,
                       int use_scaling, double radius, double dmin, double dmax, double* d, hipStream_t st);
int launch_point_damp_invert(int n_pts, const double* Hpp6, const unsigned char* pt_fixed, double* scale, int init_scale,
                             int use_scaling, double radius, double dmin, double dmax, double* dp, double* Hinv6,
                             double* zero_buf, int zero_n, hipStream_t st);
int launch_point_invert(int n_pts, const double* Hpp6, const double* dp, const unsigned char* pt_fixed,
                        double* Hinv6, hipStream_t st);
// Schur complement of the landmark blocks, row-wise with LDS accumulation (plan built on the host at create time:
// stba_ba_create).  SCHUR_SPLIT_COLS: non-zero blocks one task accumulates in LDS (two workgroups per CU);
// SCHUR_TASK_PAIRS: most (i, l) observation pairs per task (unlimited: smaller tasks measured slower).
constexpr int SCHUR_PLAN_DEFAULT = 3;                       // SchurArgs::mode of the product build
constexpr int SCHUR_MAX_SLOTS = 256;                        // LDS accumulator slots of a task: two workgroups of 81.5 KB per CU
constexpr int SCHUR_SPLIT_COLS = SCHUR_MAX_SLOTS - 8;       // blocks per task; a heavy block takes up to one extra slot per wave (parts)
constexpr int SCHUR_TASK_PAIRS = 1 << 30;
// LDS stride of one 6x6 accumulator block, in doubles: odd, so that the same entry of different blocks falls
// into different bank pairs (36 = 72 dwords = 8 mod 64 gave 8-way conflicts on every ds_add_f64: measured,
// the kernel was bound by them)
constexpr int SCHUR_BLK_LD = 37;
constexpr int SCHUR_CAM_LD = 56;      // per wave of a diagonal slice: 21 + 6 sums of the camera block, 21 + 6 of the pairs (i, i), padded
constexpr int SCHUR_ROTS = 6;        // column rotations of the Schur kernel's lanes (dealt per pair by the host: bits 16..18 of the record), see its pair loop
constexpr int SCHUR_THREADS = 512;    // 8 waves per task, two tasks per CU: 16 waves hide the L2 gathers
struct SchurArgs {
    const int* task_cam; const int* cam_start;       // camera row of the task; the camera's range of cam_perm
    const int* task_col_lo; const int* task_col_hi;  // the task's slice of its row's column list
    const int* row_col_ptr; const int* row_cols; int max_cols;
    const int* cam_perm;
    const double* J8; const unsigned char* omask;    // compact Jacobian [n_obs][8] | per-observation mask byte (or null)
    const double* Jc12 = nullptr;                    // host-linearised factors only: the camera blocks [n_obs][12] (then J8 holds {0, 0, Jp})
    const double2* r;
    const double* Hinv6; const double* gp;
    double* S; int lda; double* rhs;
    double* Hcc; double* gc;                         // camera blocks J_c^T J_c, J_c^T r: written by the slice that holds the diagonal block
    // every (observation i of the row's camera, observation l of the same landmark with camera(l) <= camera(i)) of the
    // slice, with the LDS slot of its 6x6 block resolved on the host
    const int* pair_begin; const int* pair_end;      // per (task, wave): every block is accumulated by one wave (bitwise reproducible)
    int mode = 0;                                    // 0: per-wave lists (slots dealt to waves) | 1: one list, waves add in turn (token) | 2: one list, arrival order
    const int* task_vs_ptr; const int* vs_first;     // per task: [blocks + 1] first accumulator slot of every block of the slice
    const int* obs_pt;                               // landmark of every observation
    const int4* pair_rec;                            // (i, l, landmark, slot | 0x8000 if diagonal block), l != i
    // round 6, landmark-range slices (few camera rows; null otherwise): the task's range of cam_perm, where it writes its partial
    // blocks [ncols * 36 | rhs 6 (+2) | camera sums 27 ...], and every row's tasks in the order ba_schur_reduce_slices_kernel adds them
    const int* task_p_lo = nullptr; const int* task_p_hi = nullptr; const long long* task_part_off = nullptr; double* part = nullptr;
    const int* row_task_ptr = nullptr; const int* row_tasks = nullptr; int n_cams = 0;
    int ablate = 0;
};
// the Schur complement for dense visibility as a symmetric rank-k product (ba_kernels.hip, "DENSE visibility")
struct SchurDenseArgs {
    int n_cams, n_chunks;                         // chunks of <= CAM_CHUNK observations of one camera (the camera-block machinery's)
    const int* chunk_begin; const int* chunk_end; const int* cam_chunk_start; const int* cam_perm;
    const int* obs_cam; const int* obs_pt;
    const double* J8; const unsigned char* omask; const double* Jc12; const double2* r;
    const double* Hinv6; const double* gp;
    double* Y; size_t ldy; size_t kcols;         // [lda][ldy], kcols = 3 n_pts rounded up to 16 (<= ldy); zero where nothing is observed
    double* partial;                              // schur_dense_partial_doubles(n_chunks)
    const unsigned char* dup_run;                 // per position of cam_perm: repeated (camera, landmark) pairs (null: none), see the chunk kernel
    double* ws;                                   // workspace of chol_yyt_workspace_doubles(lda, kcols) doubles (or null)
    double* S; int lda; double* rhs; double* Hcc; double* gc;
};
size_t schur_dense_partial_doubles(int n_chunks);
int launch_schur_dense(const SchurDenseArgs& a, hipStream_t st);
size_t schur_rows_lds_bytes(int max_cols);
int launch_schur_rows(const SchurArgs& a, int n_tasks, hipStream_t st);
int launch_reduced_add_camera(int n_cams, const double* Hcc, const double* gc, double* S, int lda, double* rhs,
                              double* ex_diag, double* ex_gc, hipStream_t st);
int launch_reduced_finalize(int n_cams, int n, const double* Hcc, const double* gc, const unsigned char* cam_fixed, double* S, int lda,
                            double* rhs, double* ex_diag, double* ex_gc, double* scale, int init_scale, int use_scaling,
                            double radius, double dmin, double dmax, double* dc, const double* scalars, int n_scalars,
                            double* host_out, int reduced, hipStream_t st);
int launch_reduced_damp(int n, const double* dc, const unsigned char* cam_fixed, double* S, int lda, double* rhs,
                        hipStream_t st);
// the landmark half of the trial point made by the back-substitution kernel itself (LM loop): pts_new = pts (+) dxp and one
// partial[4] = {|step|^2, |x|^2, model term, 0} per landmark workgroup of the launch (backsub_grid of them); all null: plain back-substitution
// (and the camera half, in extra workgroups behind the landmark ones, when cams_new is given: backsub_cam_grid partials)
struct BacksubUpdate {
    const double* pts; const unsigned char* pt_fixed; const double* dp;
    double* pts_new; double* partial_p;
    int n_cams; const double* cams; const unsigned char* cam_fixed; const double* gc; const double* dc;
    double* cams_new; double* partial_c;
};
int backsub_grid(int n_pts);
int backsub_cam_grid(int n_cams);
int launch_backsub(int n_pts, const int* pt_start, const int* obs_cam, const double* J8, const unsigned char* omask,
                   const double* Hinv6, const double* gp, const double* dxc, double* dxp, hipStream_t st, const BacksubUpdate* up = nullptr,
                   const double* Jc12 = nullptr);
int launch_update(int n_cams, int n_pts, const double* cams, const double* pts, const double* dxc,
                  const double* dxp, const unsigned char* cam_fixed, const unsigned char* pt_fixed,
                  const double* gc, const double* dc, const double* gp, const double* dp, double* cams_new,
                  double* pts_new, double* partial_c, double* partial_p, hipStream_t st);
int launch_triangulate(int n_pts, const int* pt_start, const int* obs_cam, const double2* feat, const double* cams,
                       double* pts, const unsigned char* pt_fixed, int max_iter, hipStream_t st);
int launch_calib_linearize(int n_views, int n_corners, const double* params, const double* obj, const double* img,
                           double* e, double* Ji, double* Jx, double* sse_partial, hipStream_t st);
constexpr int CALIB_GRAM_DOUBLES = 136, CALIB_SCRATCH_DOUBLES = 114;      // per view
int launch_calib_arrow_iteration(int n_views, int n_corners, double* params, const double* obj, const double* img, double* gram,
                                 double* scratch, int* state, double* sse_trace, hipStream_t st);
int launch_dense_normal(int n_res, int n, const double* J, const double* r, double* H, int ldh, double* g,
                        hipStream_t st);

// ---- small dense problems (small_dense.hip): one kernel launch per LM step, pooled workspace, mapped host buffers
constexpr int SMALL_DENSE_MAX_N = 32;
struct SmallDenseWs;
bool small_dense_fits(int n_res, int n);
int small_dense_acquire(SmallDenseWs** ws, int n_res, int n);
void small_dense_release(SmallDenseWs* ws);
double* small_dense_J(SmallDenseWs* ws);     // where the residual callback writes J (n_res x n) and r (n_res): pinned, mapped
double* small_dense_r(SmallDenseWs* ws);
// H = J^T J, g = J^T r (relinearize) or the H, g of the last linearisation; Jacobi scaling fixed when `first`; solves
// (H + D(radius)) dx = -g.  dx, g: pointers into mapped host memory, valid until the next step.  pivot_flag: 0 or row + 1
int small_dense_step(SmallDenseWs* ws, int n_res, int n, bool relinearize, bool first, bool jacobi, double radius, double dmin,
                     double dmax, const double** dx, const double** g, double* model_change, int* pivot_flag);

}  // namespace stba

/*
 * stba.h -- C ABI of the MI355X-native nonlinear-least-squares engine ("slam-tricks bundle
 * adjustment").  This is the drop-in boundary for the one hot path of Unsigned-Long/slam-tricks:
 * the Ceres-driven NLS solve of st17-ceres / st20-g2o / st3-calibration.
 *
 * Plain C: opaque handles, caller-owned buffers, int status codes, no exceptions, no torch or
 * HIP types in any signature (a HIP stream is passed as void*).  Everything FP64.
 * All `double*`/`int*` arguments are HOST pointers unless the name ends in `_dev`.
 *
 * What each entry point replaces in the reference (paths relative to /root/reference):
 *
 *   stba_ba_create / _set_params / _get_params
 *        ceres::Problem construction for the BA factor graph: AddResidualBlock(ProjectFactor,
 *        {SO3, POS, landmark}) + AddParameterBlock(SO3, 4, LieLocalParameterization<SO3d>) +
 *        SetParameterBlockConstant               st20-g2o/src/include/test_ceres.h:109-130
 *        (same factor with constant landmarks = the PnP problems of
 *                                                st17-ceres/src/include/solver.hpp:247-385)
 *   stba_ba_evaluate      CostFunction::Evaluate over all residual blocks: residuals + Jacobians
 *                         test_ceres.h:63-80 (ProjectFactor), solver.hpp:168-212 (analytic form,
 *                         with the hat(pInC) rotation block -- SURVEY.md header fact 2),
 *                         composed with LieLocalParameterization::ComputeJacobian solver.hpp:48-54
 *   stba_ba_normal_blocks block-sparse J^T J / J^T r      solver.hpp:402-436 (H += J^T J, g += -J^T r),
 *                         block structure sim_data.h:108-159
 *   stba_ba_reduced_system / stba_ba_solve_reduced / stba_ba_back_substitute
 *                         options.linear_solver_type = SPARSE_SCHUR   test_ceres.h:145
 *                         (g2o: BlockSolver<6,3> + setMarginalized    test_g2o.h:95-100,121)
 *   stba_ba_apply_step    LieLocalParameterization::Plus solver.hpp:38-45 / oplusImpl test_g2o.h:36-39,60-63
 *   stba_ba_solve         ceres::Solve(options, &problem, &summary)   test_ceres.h:148, solver.hpp:286
 *   stba_ba_lm_iterations fixed-work LM iterations (bench mode; no reference counterpart)
 *   stba_ba_triangulate   per-landmark Triangulation solves           sim_data.cpp:299-311
 *   stba_dense_*          the small dense problems: DENSE_QR path     solver.hpp:282, ceres_bound.cpp:40
 *   stba_cholesky_*       dense SPD factor/solve used on the reduced camera system
 *                         (hMat.ldlt().solve(gMat) solver.hpp:438; calib.cpp:393)
 */
#ifndef STBA_H
#define STBA_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STBA_VERSION 6

/* status codes */
enum {
    STBA_OK = 0,
    STBA_ERR_INVALID_ARGUMENT = -1,
    STBA_ERR_NO_DEVICE = -2,       /* no HIP device / runtime failure: the product has no CPU fallback */
    STBA_ERR_HIP = -3,
    STBA_ERR_NOT_POSITIVE_DEFINITE = -4,
    STBA_ERR_ALLOC = -5,
    STBA_ERR_STATE = -6,
    STBA_ERR_CALLBACK = -7,
    STBA_ERR_NO_SOLUTION = -8      /* a closed-form initialiser found no (unique) solution */
};

const char* stba_status_string(int status);
const char* stba_last_error(void);      /* thread-local detail of the last failure */
int stba_version(void);                 /* 6: stba_pcg_options grew coarse_async, forcing_step_accuracy (append-only); 5: stba_lm_options grew function_tolerance_takes_step */
/* number of visible HIP devices (0 => every compute entry point returns STBA_ERR_NO_DEVICE) */
int stba_device_count(void);

/* ---- Levenberg-Marquardt options: the fields the reference sets plus Ceres' defaults --------
 * (solver.hpp:272-282, test_ceres.h:133-146; defaults: SURVEY.md 8c) */
typedef struct {
    int    max_num_iterations;            /* 50 */
    double initial_trust_region_radius;   /* 1e4 */
    double max_trust_region_radius;       /* 1e16 */
    double min_trust_region_radius;       /* 1e-32 */
    double min_relative_decrease;         /* 1e-3 */
    double min_lm_diagonal;               /* 1e-6 */
    double max_lm_diagonal;               /* 1e32 */
    double function_tolerance;            /* 1e-6 */
    double gradient_tolerance;            /* 1e-10 */
    double parameter_tolerance;           /* 1e-8 */
    int    jacobi_scaling;                /* 1 */
    int    num_threads;                   /* accepted and ignored (reference sets 1) */
    int    minimizer_progress_to_stdout;  /* solver.hpp:278 */
    int    update_state_every_iteration;  /* solver.hpp:277: copy parameters back before callbacks */
    int    phase_timing;                  /* 0 (default): no per-phase device times.  1: hipEvents between the phases of every
                                           * iteration fill stba_lm_summary::ms_* -- an event is a packet of its own on the queue
                                           * and costs ~5 us of idle GPU, ~1.5 % of a C5 iteration for the eight it takes */
    int    function_tolerance_takes_step; /* what happens to the trial step on which |cost change| <= function_tolerance * cost fires:
                                           * 1 (default): it is taken if it is a decrease (rho > min_relative_decrease), THEN convergence
                                           *    is reported;
                                           * 0: convergence is reported at once and the step is NOT taken -- the order of Ceres'
                                           *    TrustRegionMinimizer::Minimize since the refactoring of 1.12 (through 2.1):
                                           *    ComputeCandidatePointAndEvaluateCost(); if (ParameterToleranceReached()) return;
                                           *    if (FunctionToleranceReached()) return; if (IsStepSuccessful()) HandleSuccessfulStep();
                                           *    -- the candidate becomes the state only in HandleSuccessfulStep.
                                           * So 0 is Ceres' stopping state.  1 stays the default because the step that is given up is a
                                           * full Newton step of a converged model and not a small one: on the C4 pose graph it still moves
                                           * poses by 3.6e-3 (cost by < 1e-6 relative), and it is what brings a run with inexact steps back
                                           * onto the exact-step trajectory -- measured in round 6 (profiles/r6_c4_async.txt): production
                                           * against exact steps, poses 2.5e-6 apart with the step, 5.7e-5 without (north_star asks for
                                           * 1e-5).  Taking it can only lower the final cost.  Engine and oracle (orc_lm_options) carry the
                                           * same switch and agree under both settings; a caller who wants Ceres' own final state sets 0.
                                           * (No Ceres exists in this image to run against; tools/ceres_baseline.cpp on a box that has
                                           * one shows the order in num_successful_steps.) */
} stba_lm_options;

void stba_lm_default_options(stba_lm_options* opt);

/* STBA_FAILURE: the START point could not be evaluated -- a non-finite cost, i.e. a non-finite input or residual (Ceres: "Initial
 * residual and Jacobian evaluation failed"; reason STBA_TERM_SOLVER_FAIL, zero iterations, parameters untouched) -- or a callback
 * failed.  A non-finite cost at a TRIAL point is an unsuccessful step, as in Ceres: rejected, the radius shrinks, the solve goes on. */
enum { STBA_CONVERGENCE = 0, STBA_NO_CONVERGENCE = 1, STBA_FAILURE = 2 };
enum { STBA_TERM_NONE = 0, STBA_TERM_GRADIENT = 1, STBA_TERM_FUNCTION = 2, STBA_TERM_PARAMETER = 3,
       STBA_TERM_MAX_ITER = 4, STBA_TERM_MIN_RADIUS = 5, STBA_TERM_SOLVER_FAIL = 6,
       STBA_TERM_FIXED = 7, STBA_TERM_USER = 8 };

typedef struct {
    int    termination_type;
    int    termination_reason;
    int    num_iterations;
    int    num_successful_steps;
    int    num_unsuccessful_steps;
    double initial_cost;
    double final_cost;
    double final_radius;
    double final_gradient_max_norm;
    double seconds_total;
    /* device time per phase, milliseconds, summed over iterations (hipEvent); zero unless stba_lm_options::phase_timing.
     * ms_backsub: back-substitution + the manifold update to the trial point (one kernel); ms_cost: evaluation of the trial
     * point -- when the loop speculates (no callback, no progress output) that is the FULL linearisation at the trial point,
     * and ms_linearize then holds only the landmark blocks behind it */
    double ms_linearize, ms_schur, ms_solve, ms_backsub, ms_cost;
    /* several ranks: the cross-rank sums of the reduced camera system of this run -- device time between
     * events around the collective (part of ms_schur), bytes handed to it per rank, number of calls; 0 on one rank */
    double ms_allreduce, allreduce_bytes;
    int    allreduce_calls;
} stba_lm_summary;

/* iteration trace row: cost, cost_change, gradient_max_norm, step_norm, relative_decrease,
 * radius, accepted */
#define STBA_TRACE_COLS 7

/* IterationCallback (solver.hpp:215-245, test_ceres.h:83-96): return 0 to continue */
typedef int (*stba_iteration_callback)(void* user, int iteration, double cost, double cost_change,
                                       double gradient_max_norm, double step_norm, double radius,
                                       int step_is_successful);

/* cross-rank sum of `count` doubles resident on the device, enqueued on `hip_stream`
 * (landmark sharding, SURVEY.md 8e: RCCL all-reduce of the reduced camera system).  Return 0. */
typedef int (*stba_allreduce_fn)(void* user, void* buf_dev, size_t count, void* hip_stream);

/* ================================ native RCCL communicator ================================
 * One process per GPU.  What a C++ host of the reference (st20-g2o/src/src/test_ceres.cpp:7-19)
 * uses instead of a Python hook: rank 0 makes the id, hands it to the other ranks over its own side channel,
 * every rank creates its communicator and gives it to the engines (stba_ba_set_comm / stba_pg_set_comm), which
 * then call ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, comm, stream) on THEIR stream.  RCCL is bound
 * lazily (dlopen): hosts that never create a communicator need no librccl.
 * device < 0: the calling thread's current device. */
#define STBA_COMM_ID_BYTES 128
typedef struct stba_comm stba_comm;
int stba_comm_unique_id(char id[STBA_COMM_ID_BYTES]);
int stba_comm_create(stba_comm** out, const char id[STBA_COMM_ID_BYTES], int rank, int world_size, int device);
int stba_comm_destroy(stba_comm* comm);
int stba_comm_rank(const stba_comm* comm, int* rank, int* world_size);
/* in-place sum of `count` doubles on the device across all ranks, enqueued on hip_stream */
int stba_comm_allreduce_sum(stba_comm* comm, void* buf_dev, size_t count, void* hip_stream);
/* the same as an stba_allreduce_fn (user = stba_comm*) */
int stba_comm_allreduce_hook(void* user, void* buf_dev, size_t count, void* hip_stream);

/* ================================ bundle-adjustment engine ================================ */
typedef struct stba_ba stba_ba;

/* cams: n_cams*7 (qx qy qz qw tx ty tz, camera-to-world); pts: n_pts*3; observations in any
 * order (the engine regroups them landmark-major); obs_feat: n_obs*2 normalised image coords.
 * cam_fixed: n_cams*6 bytes (1 = dof constant; NULL = all free; order [rot(3), pos(3)]);
 * pt_fixed: n_pts bytes (1 = landmark constant; NULL = all free).
 * hip_stream: hipStream_t to enqueue on (NULL = the engine creates its own).
 * The Schur complement has two forms (stba_ba_set_schur_mode).  PAIRS: a plan with one 16-byte record per PAIR of observations
 * of the same landmark (sum over landmarks of k (k + 1) / 2, k = cameras seeing it; 5.5 M at 1000 cameras x 100 000 landmarks x
 * 10 observations each), built once here, on the host and on the device; one 6 x 6 block accumulated in LDS per pair.  DENSE: the
 * scaled camera-landmark blocks in a dense [6 n_cams] x [3 n_pts] matrix Y and S = -(Y Y^T) as ONE symmetric rank-k product on the
 * matrix cores -- no plan, 8 x 18 n_cams n_pts bytes of device memory, the right form when most cameras see most landmarks.
 * The engine picks DENSE by itself when the plan would have more than 2^30 pairs (or would not fit half of the free device
 * memory), or more than 2^20 pairs at a visibility of 30 % and more; a problem that fits neither form is refused with
 * STBA_ERR_INVALID_ARGUMENT. */
int stba_ba_create(stba_ba** out, int n_cams, int n_pts, int n_obs, const double* cams,
                   const double* pts, const int* obs_cam, const int* obs_pt, const double* obs_feat,
                   const unsigned char* cam_fixed, const unsigned char* pt_fixed, void* hip_stream);
int stba_ba_destroy(stba_ba* ba);
/* form of the Schur complement (above): STBA_SCHUR_AUTO = what stba_ba_create chose.  STBA_SCHUR_PAIRS fails if the engine was
 * created without a pair plan (too many pairs), STBA_SCHUR_DENSE if Y does not fit the device. */
enum { STBA_SCHUR_AUTO = 0, STBA_SCHUR_PAIRS = 1, STBA_SCHUR_DENSE = 2 };
int stba_ba_set_schur_mode(stba_ba* ba, int mode);
int stba_ba_schur_mode(const stba_ba* ba, int* mode);
int stba_ba_set_params(stba_ba* ba, const double* cams, const double* pts);
int stba_ba_get_params(stba_ba* ba, double* cams, double* pts);
/* (round 6) the observations' features again, n_obs*2 in the order given to stba_ba_create: for a caller who learns them while the engine
 * is being created -- include/stba/ceres.h creates the engine (regrouping, Schur plan, uploads: ~50 ms at 10^6 observations) on a helper
 * thread while the calling thread evaluates the user's cost functions for their features, then hands them over here. */
int stba_ba_set_features(stba_ba* ba, const double* obs_feat);
/* the HIP device of the CALLING thread (a helper thread starts on device 0 whatever its parent had selected): read it on one thread,
 * set it on the other */
int stba_get_device(int* device);
int stba_set_device(int device);
/* BA-SHAPED problems whose factor is NOT the built-in reprojection (the reference's BA cost is a generic
 * DynamicAutoDiffCostFunction, test_ceres.h:56,109-121: a robustified, scaled or otherwise different 4+3+3 -> 2 factor must not
 * fall back to dense normal equations).  With a host lineariser the residuals and Jacobians of every observation come from the
 * caller -- the C++ shim evaluates the user's cost functions in bulk -- and everything behind them runs on the device
 * unchanged: landmark blocks, Schur complement, dense factorisation, back-substitution, manifold update, LM control.
 *   fn(user, cams[n_cams*7], pts[n_pts*3], r[n_obs*2], Jc, Jp) -> 0 on success
 * in the CALLER's observation order (the order given to stba_ba_create; obs_feat is not used in this mode);
 * Jc[n_obs*12]: 2x6 row-major w.r.t. the camera's LOCAL coordinates [dtheta(3) of q <- q (x) exp(dtheta), dt(3)];
 * Jp[n_obs*6]: 2x3 w.r.t. the landmark; Jc = Jp = NULL when only the cost of a trial point is wanted.
 * The callback runs on the calling thread, between kernels: the LM loop does not overlap host and device work in this mode. */
typedef int (*stba_ba_linearize_fn)(void* user, const double* cams, const double* pts, double* r, double* Jc, double* Jp);
int stba_ba_set_host_linearizer(stba_ba* ba, stba_ba_linearize_fn fn, void* user);
/* multi-GPU: this engine holds one landmark shard; all cameras are replicated.  The hook sums
 * the packed reduced system / scalars across ranks.  rank 0 owns the once-only diagonal terms. */
int stba_ba_set_allreduce(stba_ba* ba, stba_allreduce_fn fn, void* user, int rank, int world_size);
/* the same with a native communicator (NULL: back to a single rank) */
int stba_ba_set_comm(stba_ba* ba, stba_comm* comm);
/* padded order of the dense reduced system (multiple of the factorisation block) */
int stba_ba_reduced_dim(const stba_ba* ba, int* n, int* n_padded);

/* --- stage entry points (each runs the HIP kernels and optionally copies results out) ------ */
/* residuals r[n_obs*2], Jc[n_obs*12] (2x6 row-major), Jp[n_obs*6] (2x3), in the caller's
 * observation order; any output may be NULL.  cost = 1/2 sum r^2. */
int stba_ba_evaluate(stba_ba* ba, double* cost, double* r, double* Jc, double* Jp);
/* cost only (residual kernel without Jacobians), at the current parameters */
int stba_ba_cost(stba_ba* ba, double* cost);
/* needs a preceding stba_ba_evaluate.  Hcc[n_cams*36], gc[n_cams*6], Hpp[n_pts*9], gp[n_pts*3] */
int stba_ba_normal_blocks(stba_ba* ba, double* Hcc, double* gc, double* Hpp, double* gp);
/* needs normal blocks.  dc[n_cams*6], dp[n_pts*3]: diagonal damping.  S: n*n row-major with
 * n = 6*n_cams (lower triangle valid), rhs[n].  S/rhs may be NULL (device-only). */
int stba_ba_reduced_system(stba_ba* ba, const double* dc, const double* dp, double* S, double* rhs);
/* Cholesky + solve of the reduced system on the device: dxc[n_cams*6] */
int stba_ba_solve_reduced(stba_ba* ba, double* dxc);
/* dxp[n_pts*3] from the camera step currently on the device */
int stba_ba_back_substitute(stba_ba* ba, double* dxp);
/* trial point = current (+) step on the device; returns its cost; accept != 0 keeps it */
int stba_ba_apply_step(stba_ba* ba, int accept, double* new_cost);

/* --- whole solves ------------------------------------------------------------------------ */
/* trace: (max_num_iterations+1)*STBA_TRACE_COLS doubles or NULL */
int stba_ba_solve(stba_ba* ba, const stba_lm_options* opt, stba_lm_summary* summary, double* trace,
                  stba_iteration_callback cb, void* cb_user);
/* exactly `iterations` LM iterations, each re-linearising, solving and evaluating the trial
 * point; no convergence tests.  The unit of work bench.py times.  Parameters stay on the device. */
int stba_ba_lm_iterations(stba_ba* ba, const stba_lm_options* opt, int iterations,
                          stba_lm_summary* summary, double* trace);
/* per-landmark refinement with cameras fixed (sim_data.cpp:299-311) */
int stba_ba_triangulate(stba_ba* ba, int max_iter);

/* average device time (ms) of the residual+Jacobian kernel over `reps` back-to-back launches,
 * measured with hipEvents on the engine's stream (bench.py roofline leg). */
int stba_ba_time_linearize(stba_ba* ba, int reps, double* ms_avg);
/* the same for the Schur-complement kernel (linearises at the current point first); also hands back the number of LDS
 * FP64 atomics and of observation pairs of one launch (either may be NULL) */
int stba_ba_time_schur(stba_ba* ba, int reps, double* ms_avg, double* lds_atomics_per_launch, double* pairs_per_launch);

/* ================================ dense SPD solver ======================================== */
/* A: n*n row-major SPD (lower triangle read), overwritten by L (lower).  Runs the blocked MFMA
 * Cholesky on the device.  Returns STBA_ERR_NOT_POSITIVE_DEFINITE if a pivot fails. */
int stba_cholesky_factor(double* A, int n, void* hip_stream);
/* solves A x = b for one right-hand side: b overwritten by x */
int stba_cholesky_solve(const double* A, int n, double* b, void* hip_stream);
/* device-resident timing of factor+solve on an n x n synthetic SPD system, ms per solve */
int stba_cholesky_time(int n, int reps, double* ms_avg, void* hip_stream);

/* the same, split by a hipEvent between the factorisation (one persistent kernel, chol_mega_kernel) and
 * the backward substitution: bench.py's MFMA roofline leg divides the n^3/3 flops by ms_factor. */
int stba_cholesky_time_split(int n, int reps, double* ms_factor, double* ms_backward, void* hip_stream);

/* diagnostic, runs on the host (no GPU): the makespan in microseconds that the host-side scheduling model
 * predicts for the persistent factorisation kernel of an n x n system on `n_xcd` XCDs with `wg_per_xcd`
 * workgroups each (MI355X: 8 x 32).  The ticket order of the kernel comes from this model. */
int stba_cholesky_schedule_model(int n, int n_xcd, int wg_per_xcd, double* makespan_us);
/* how often, in this process, the persistent factorisation gave up waiting for a dependency (its workgroups were not all
 * resident: the device is shared with another process) and the stage kernels -- one launch per stage and panel, nothing
 * resident -- took over */
int stba_cholesky_timeout_count(void);
/* how long a workgroup of the persistent factorisation waits for a dependency before the program gives up (microseconds;
 * 0 restores the automatic bound, max(100 ms, 40 x the predicted makespan)).  Process-wide. */
int stba_cholesky_set_timeout_us(double us);

/* design study, runs on the host (no GPU): the same task graph spread over `n_gpus` GPUs (the reduced
 * camera system as the next thing to shard).  Tile rows are dealt to the GPUs block-cyclically, `rows_per_group`
 * consecutive 128-row tile rows at a time (0: one per XCD, i.e. n_xcd rows); a task runs on the GPU that owns the tile
 * row it writes; a dependency that crosses GPUs costs `hop_us` (flag over xGMI) plus the transfer of one 128 x 128
 * FP64 tile (128 KiB) at `link_gb_per_s`.  Link contention is NOT modelled; remote_tiles_busiest_gpu (x 128 KiB) is
 * the ingress volume to hold against the aggregate xGMI bandwidth.  DESIGN.md tabulates it for 1/2/4/8 GPUs.
 * stba_cholesky_shard_owner returns the distribution (tile row -> GPU) the model uses. */
int stba_cholesky_shard_model(int n, int n_gpus, int n_xcd, int wg_per_xcd, int rows_per_group, double hop_us, double link_gb_per_s,
                              double* makespan_us, double* cross_gpu_dependencies, double* remote_tiles_busiest_gpu);
int stba_cholesky_shard_owner(int n_block_rows, int n_gpus, int rows_per_group, int* owner_gpu);

/* hipEvent time (ms) of one factor+solve per kernel class, with the stage-per-kernel schedule (a
 * diagnostic: the production path runs the stages as tasks of one persistent kernel): ms4 = {diagonal blocks, panel solves,
 * MFMA trailing updates, backward substitution}; algorithmic / executed flops of the trailing
 * updates and their launch count (bench.py MFMA roofline leg). */
int stba_cholesky_profile(int n, double* ms4, double* syrk_flops, double* syrk_flops_padded,
                          int* syrk_launches, void* hip_stream);

/* ================================ two-view initialiser ====================== */
/* st22-two-view/src/src/two_view_geometry.cpp:18-126 (FindFunctionalMatrix): fundamental matrix from n >= 8
 * pixel correspondences (x1^T F x2 = 0, no normalisation), essential matrix with K, the four (R, t)
 * hypotheses, the cheirality test over ALL points, DLT triangulation.  f1, f2: n*2 pixels; K: 3x3
 * row-major.  Out: F (9, row-major, unit norm, sign free; may be NULL), R (9) and t (3, unit norm): the
 * pose of frame 2 in frame 1 (p1 = R p2 + t), pts (n*3 landmarks in frame 1, for the UNIT baseline; may be
 * NULL), fails (4 counters of points failing the test per hypothesis; their order follows :61-64 but depends
 * on the sign conventions of the 3x3 SVD; may be NULL).
 * Returns STBA_ERR_NO_SOLUTION if not exactly one hypothesis passes (the reference returns an empty
 * optional).  Device: the n x 9 system is reduced to its triangular factor with Givens rotations, the
 * cheirality test and the triangulation run one correspondence per lane. */
int stba_two_view_init(int n, const double* f1, const double* f2, const double* K, double* F_out, double* R_out,
                       double* t_out, double* pts_out, int* fails_out, void* hip_stream);

/* ================================ trajectories ============================== */
/* Odometry files, st16-pcl-viewer/src/src/scene.cpp:66-110: "format ascii 1.0", "element odometryInfo N",
 * property lines, "end_header", then N lines "timeStamp qx qy qz qw x y z" (quaternion normalised on read,
 * translation parsed through float like the reference's std::stof).  poses: N*7 (qx qy qz qw x y z), the
 * layout of stba_pg_create; pass stamps = poses = NULL to query N. */
int stba_odometry_read(const char* path, int* n_poses, double* stamps, double* poses, int capacity);
int stba_odometry_write(const char* path, int n_poses, const double* stamps, const double* poses);
/* absolute trajectory error, st4-kalman/src/src/pose_simulation.cpp:198-209:
 * sqrt(mean |log(T_truth^-1 T_estimate)|^2), 6-vector SE3 logarithm */
int stba_trajectory_ate(int n_poses, const double* truth, const double* estimate, double* ate);

/* ================================ calibration data formats ================== */
/* Chessboard corner files, st3-calibration/src/src/cbcorner.cpp:34-73: header "rows,cols", then one
 * "i,j,x,y" line per corner (x, y parsed through float like the reference's std::stof; written with
 * 3 decimals).  xy: rows*cols*2 doubles, row-major by (i, j); pass xy = NULL to query the size. */
int stba_corners_read(const char* path, int* rows, int* cols, double* xy, int capacity);
int stba_corners_write(const char* path, int rows, int cols, const double* xy);
/* Zhang's closed-form start point of the refinement (st3-calibration/src/src/calib.cpp:55-173):
 * DLT homographies -> intrinsics (zero skew) -> per-view extrinsics.  obj / img: [V*C*2] board points
 * (X, Y) / pixels; params out: [alpha beta u0 v0 0 0 0 0 0 | xi_0(6) ..] in stba_calib_* layout;
 * homographies out (may be NULL): V*9 row-major, unit Frobenius norm, sign as the null vector came out. */
int stba_zhang_init(int n_views, int n_corners, const double* obj, const double* img, double* params,
                    double* homographies);

/* ================================ Zhang calibration (st3-calibration) ==================== */
/* CalibSolver::totalOptimization's residual / Jacobian blocks (st3-calibration/src/src/calib.cpp:311-391,
 * distortNormPt :254-262, normPt2ImgPt :247-252) evaluated by a HIP kernel, one corner per lane.
 * params: [alpha beta u0 v0 k1 k2 k3 p1 p2 | xi_0(6) .. xi_{V-1}(6)], xi = se3 log [rho, theta];
 * obj / img: [V*C*2] board points (X, Y) / measured pixels.  Outputs (any may be NULL):
 * sse = sum e^2, e[V*C*2], Ji[V*C*18] (2x9 intrinsics+distortion), Jx[V*C*12] (2x6 left pose perturbation). */
int stba_calib_evaluate(int n_views, int n_corners, const double* params, const double* obj, const double* img,
                        double* sse, double* e, double* Ji, double* Jx);
/* totalOptimization (calib.cpp:282-422): plain Gauss-Newton, <= max_iter iterations, stop when
 * |update| < 1e-8; additive update of the 9 shared parameters, left-multiplicative SE3 update of
 * every view.  params in/out; sse_trace[max_iter] receives sum e^2 at the start of each iteration. */
int stba_calib_gauss_newton(int n_views, int n_corners, double* params, const double* obj, const double* img,
                            int max_iter, double* sse_trace, int* iterations);

/* ================================ pose graph (BASELINE config C4) ======================== */
/* BUILD-DEFINED: the reference has no pose-graph code (SURVEY.md header fact 3).  Conventions from the
 * reference's Lie-group notes (st23-lie-group-v2/doc.tex:862-996): node pose T = (qx qy qz qw tx ty tz),
 * right-multiplicative update T <- T exp(delta), tangent [rho, theta]; edge measurement Z_ij ~ T_i^-1 T_j;
 * residual r_ij = log(Z_ij^-1 T_i^-1 T_j).  Levenberg-Marquardt; the damped normal equations are solved matrix-free by
 * conjugate gradients with a two-level preconditioner (6x6 block Jacobi + a coarse space of six rigid-body modes per group
 * of consecutive nodes, inverted densely) as an INEXACT Newton step: the PCG stops at |r| <= eta_k |g|, decided on the
 * device; eta_k follows Eisenstat & Walker's forcing sequence (eta_0 = forcing_eta0, tightening as the gradient falls). */
typedef struct stba_pg stba_pg;
typedef struct {
    int    max_iterations;       /* 1000: cap on the PCG iterations of one linear solve (stba_pcg_summary::hit_cap counts the solves that reach it) */
    double relative_tolerance;   /* 1e-12 on |residual| / |rhs|: used when forcing_eta0 <= 0 (a fixed tolerance, i.e. exact steps) */
    int    check_every;          /* 4: PCG iterations enqueued between the host's looks at the device-side convergence flag */
    double forcing_eta0;         /* 0.1: first and largest forcing term; <= 0: fixed relative_tolerance */
    double forcing_eta_min;      /* 0.01 (round 6; 1e-10 until then): the floor of the forcing sequence.  Eisenstat & Walker's sequence
                                  * falls quadratically near the solution and buys nothing there: the LM loop stops on its function
                                  * tolerance, not on the linear residual.  Measured (floor: PCG iterations | converged poses from the
                                  * exact-step run, north_star's gate 1e-5) -- C4: 1e-10: 217 | 2.9e-6, 0.01: 196 | 3.2e-6 (1444 -> 1505 LM
                                  * it/s), 0.03: 168 | 3.5e-6, 0.1 (a constant eta, Ceres' default for inexact steps): 125 | 7.4e-6; but on
                                  * a 400-node graph with twice the noise 6.8e-6 | 6.6e-6 | 1.1e-5 | 8.3e-6 and on 2000 nodes with three
                                  * times the noise 1.0e-5 | 1.1e-5 | 2.6e-5 | 4.6e-5: 0.01 is the largest floor that leaves the answers
                                  * where the sequence alone puts them (profiles/r6_pg_forcing_floor.txt, r6_c4_forcing.txt) */
    int    coarse_group;         /* nodes per group of the coarse space, a power of two; 0: automatic (<= 200 groups, >= 8 nodes);
                                  * -1: no coarse space (block Jacobi only, the round-3 preconditioner) */
    int    coarse_refresh_every; /* 1: LM iterations between re-inversions of the coarse operator (the first two iterations always make theirs; an
                                  * inverse made one iteration earlier still is a preconditioner, only a weaker one).  2 is FASTER at C4 since the PCG
                                  * iteration costs 13 us instead of 44 (1270 against 1080 LM it/s) but its inexact steps end 3.9e-5 from the
                                  * exact-step oracle's poses -- outside north_star's 1e-5 -- where 1 ends 2.5e-6 away: 1 stays the default */
    int    one_kernel_solve;     /* 1: the PCG solve of an LM iteration as ONE persistent kernel (two stamped exchanges per iteration
                                  * instead of four launches) where the graph allows: one rank, a coarse space of <= 256 groups of
                                  * <= 64 nodes; 0: always four launches per iteration; 2: as 1 with a time-out of zero, so that the way back is
                                  * taken -- a solve whose workgroups are not all resident gives up and is repeated with launches, and the
                                  * engine stays with launches (stba_version() >= 5); 3 (stba_version() >= 6): as 1 with the stamps of the
                                  * two exchanges published behind an agent-scope RELEASE fence and read in front of an ACQUIRE fence -- the
                                  * formally complete protocol; 1 orders the same accesses by the hardware's own rules (stores acknowledged
                                  * before the stamp is issued, loads issued after the stamp was seen) plus compiler barriers */
    int    coarse_async;         /* 1 (stba_version() >= 6): the coarse operator is built and inverted on a SECOND stream, next to the PCG kernel, and
                                  * applied one LM iteration late -- iteration k preconditions with the inverse of iteration k - 1's operator --
                                  * EXCEPT behind a long step (coarse_async_decrease) and in the first iteration(s) (coarse_async_after), which
                                  * wait for the inverse of their own operator.  2: lag always, and with a forcing sequence the first solve runs
                                  * on block Jacobi alone.  0: always in line, as until version 5 (coarse_refresh_every applies).  Ordered by
                                  * events: run-to-run reproducible.  MEASURED at C4 (profiles/r6_c4_async*.txt), LM it/s | distance of the
                                  * converged poses from the exact-step trajectory: in line 1075 | 2.5e-6; lag from iteration 2 on 1380 | 6.0e-5;
                                  * from 3 on 1382 | 3.9e-5; from 4 on (this rule's choice) 1341 | 2.9e-6; mode 2 1515 | 1.5e-4.  The coarse space
                                  * IS the graph's weakly constrained modes: a coarse solver that is stale by a long step leaves its error exactly
                                  * there, and north_star asks for 1e-5 on poses */
    double forcing_eta_final;    /* 0 = off (stba_version() >= 6; measured without effect on C4: profiles/r6_c4_async.txt): cap on the forcing term once the LM iteration is about to converge -- the last
                                  * accepted step changed the cost by less than 100 x function_tolerance (relative).  The error of the LAST
                                  * inexact step is what the converged poses keep (about eta x its length): Eisenstat & Walker's sequence
                                  * alone leaves eta ~ 1e-2 there.  0: no cap */
    double coarse_eta;           /* 0 (stba_version() >= 6; one-kernel solve with a forcing sequence only): the solve also runs until the COARSE
                                  * residual |P^T r| has fallen to this fraction of its start value -- the coarse space holds the smooth, weakly
                                  * constrained modes, where a small residual is a large error */
    int    coarse_async_after;   /* 1 (with coarse_async = 1): the first so many LM iterations wait for the inverse of their OWN coarse operator */
    double coarse_async_decrease;/* 0.5 (with coarse_async = 1): an iteration that follows an accepted step which took MORE than this fraction off
                                  * the cost also waits for its own inverse: the operator is stale by exactly that step */
} stba_pcg_options;
void stba_pcg_default_options(stba_pcg_options* o);
typedef struct {
    int    iterations_total;            /* PCG iterations of the whole solve */
    int    solves;                      /* linear solves (one per LM iteration) */
    int    hit_cap;                     /* solves that ran into max_iterations before reaching their tolerance */
    int    max_iterations_in_a_solve;
    int    coarse_dim;                  /* unknowns of the coarse space (0: none) */
    int    coarse_refreshes;            /* coarse operators inverted */
    double last_eta;                    /* forcing term of the last solve */
    int    coarse_failures;             /* coarse operators whose factorisation met a non-positive pivot: the coarse correction was
                                         * switched off (block Jacobi alone) until the next refresh (stba_version() >= 5) */
    int    one_kernel_solves;           /* linear solves that ran as one persistent kernel (stba_pcg_options::one_kernel_solve); a solve
                                         * whose workgroups were not all resident is repeated with launches and not counted */
    double linear_solve_ms;             /* (stba_version() >= 6) with stba_lm_options::phase_timing: device time of all linear solves of the
                                         * LM solve, hipEvents on the engine's stream around the persistent kernel (or the PCG launches); else 0 */
} stba_pcg_summary;
int stba_pg_create(stba_pg** out, int n_nodes, int n_edges, const double* poses, const int* edge_i, const int* edge_j,
                   const double* meas, const unsigned char* node_fixed, void* hip_stream);
int stba_pg_destroy(stba_pg* pg);
/* multi-GPU: this engine holds one shard of the EDGES (all nodes replicated); the hook sums the gradient and
 * diagonal blocks (42 n doubles per linearisation), every matrix-vector product of the PCG (6 n doubles) and
 * the cost scalars across ranks.  rank 0 owns the once-only damping term.
 * REQUIREMENT on the hook: every rank must receive BIT-IDENTICAL sums (what ncclAllReduce / MPI_Allreduce give: one reduction
 * order for all ranks).  The solve state is replicated and every rank decides from ITS copy of the device-side convergence
 * flag how many further products -- i.e. collectives -- to enqueue; a hook whose result differs in the last bit between ranks
 * (a reduce + broadcast is fine, a per-rank gather-and-sum in rank-dependent order is not) can make ranks disagree by one
 * chunk of PCG iterations and hang in mismatched collectives. */
int stba_pg_set_allreduce(stba_pg* pg, stba_allreduce_fn fn, void* user, int rank, int world_size);
int stba_pg_set_comm(stba_pg* pg, stba_comm* comm);
int stba_pg_get_poses(stba_pg* pg, double* poses);
/* r[n_edges*6], Ji / Jj [n_edges*36] (6x6 row-major, wrt delta_i / delta_j); any may be NULL */
int stba_pg_evaluate(stba_pg* pg, double* cost, double* r, double* Ji, double* Jj);
/* measurement: average device time (hipEvents on the engine's stream, `reps` launches each) of the residual + Jacobian
 * kernel and of one matrix-free product q = (J^T J + D) p */
int stba_pg_time_kernels(stba_pg* pg, int reps, double* ms_linearize, double* ms_matvec);
int stba_pg_solve(stba_pg* pg, const stba_lm_options* opt, const stba_pcg_options* pcg, stba_lm_summary* summary,
                  double* trace, int* pcg_iterations_total);
/* the linear-solver side of the last stba_pg_solve of this engine */
int stba_pg_last_pcg_summary(stba_pg* pg, stba_pcg_summary* out);

/* ================================ small dense LM problems ================================ */
/* Residual blocks evaluated by a HOST callback (user CostFunction::Evaluate, solver.hpp:168-212;
 * autodiff functors are differentiated on the host by the C++ shim), normal equations + LM
 * step on the device.  x: n_params ambient; J row-major n_res x n_local in LOCAL coordinates. */
typedef int (*stba_residual_fn)(void* user, const double* x, double* r, double* J);
typedef void (*stba_plus_fn)(void* user, const double* x, const double* delta, double* x_new);
int stba_dense_solve(stba_residual_fn fn, stba_plus_fn plus, void* user, int n_params, int n_local,
                     int n_res, double* x, const double* lower, const double* upper,
                     const stba_lm_options* opt, stba_lm_summary* summary, double* trace,
                     stba_iteration_callback cb, void* cb_user);

#ifdef __cplusplus
}
#endif
#endif

/* mcba.h -- C ABI of the MI355X bundle-adjustment back-end ("multical bundle adjust").
 *
 * Drop-in boundary for ONE path of oliver-batchelor/multical: the reprojection residual / Jacobian evaluation and
 * the damped normal-equation solve behind
 *     multical.optimization.Calibration.bundle_adjust      (multical/optimization/calibration.py:199-212)
 *     multical.workspace.Workspace.calibrate               (multical/workspace.py:228-247)
 * The reference has no FFI of its own (it is pure Python on top of scipy / OpenCV); this header is the interface a
 * ctypes stub inside `Calibration.bundle_adjust` binds.  Every entry point names the reference
 * function it replaces.
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types; every function returns 0 on success, non-zero on error and
 *     never throws across the ABI; `mcba_last_error()` returns a thread-local message.
 *   - all floating point is IEEE double; masks are uint8 (0/1); arrays are C-contiguous.
 *   - host buffers are caller-owned and only read during the call; the handle owns device memory and runs every
 *     kernel on the HIP stream given at creation (NULL = the handle creates its own stream).
 *   - one handle per thread; handles are independent.
 *   - the library is HIP-only: `mcba_create` fails when no gfx950 device is present.  There is no CPU fallback.
 *   - limits of this implementation (the reference has none; all are checked by `mcba_create`, which fails with a
 *     message instead of producing a handle that cannot be solved):
 *     at most 65535 points per board; about 36 000 (camera, board) pairs when per-frame rig poses are optimised (their
 *     view-rank tables live in the 150 KB of LDS of a workgroup).  Cameras of one rig may carry different numbers of
 *     distortion coefficients (4, 5, 8, 12, 14: mcba_problem.camera_n_dist) and may mix pinhole and fisheye cameras
 *     (mcba_problem.camera_fisheye).
 *
 * Parameter vector `x` (length `n_params`): exactly `Calibration.param_vec` (optimization/parameters.py:44-46):
 * the ENABLED blocks, in the order camera_poses | board_poses | motion | cameras | boards
 * (optimization/calibration.py:146-161):
 *     camera_poses : C x (rx ry rz tx ty tz)                           pose_set.py:51-53
 *     board_poses  : B x 6
 *     motion       : static   F x 6                                    motion/static_frames.py:29
 *                    rolling  F x 6 (start) then F x 6 (end)           motion/rolling_frames.py:135-140
 *                    hand-eye 6 (world_wrt_base) 6 (gripper_wrt_camera) motion/hand_eye.py:75-80
 *     cameras      : C x [fx fy | cx cy | skew | dist(n_dist)]         camera.py:144-155
 *     boards       : sum_b P_b x 3                                     board/charuco.py:112-114
 * Disabled blocks keep the values given in `mcba_problem.x_full`.
 *
 * Residual vector `r` (length `n_residuals` = 2 * #inliers): C-order over (camera, frame, board, point) restricted
 * to the inlier mask, (u,v) interleaved -- `(reprojected.points - point_table.points)[inliers].ravel()`
 * (optimization/calibration.py:204-206).
 */
#ifndef MCBA_H
#define MCBA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCBA_VERSION 3

/* motion models (multical/motion/) */
#define MCBA_MOTION_STATIC   0   /* motion/static_frames.py  */
#define MCBA_MOTION_ROLLING  1   /* motion/rolling_frames.py (linear, scan time from the observed row) */
#define MCBA_MOTION_HAND_EYE 2   /* motion/hand_eye.py       */

/* camera models */
#define MCBA_CAMERA_PINHOLE 0    /* camera.py:124-128         -> cv2.projectPoints, n_dist in {5,8,12,14} */
#define MCBA_CAMERA_FISHEYE 1    /* camera_fisheye.py:113-117 -> cv2.fisheye.projectPoints, n_dist = 4     */

/* trust-region step of the solver (scipy `tr_solver`) */
#define MCBA_TR_EXACT 0          /* regularised Gauss-Newton step from the exact normal equations (Schur + Cholesky): fast,     */
                                 /* ends at the converged optimum                                                              */
#define MCBA_TR_LSMR  1          /* scipy's choice for the reference's sparse Jacobian: LSMR with atol = btol = 1e-6 -- the     */
                                 /* reference's trajectory and end point                                                       */

/* robust losses (scipy.optimize.least_squares `loss`, reachable through OptimizerOpts.loss, config/arguments.py:61) */
#define MCBA_LOSS_LINEAR  0
#define MCBA_LOSS_SOFT_L1 1
#define MCBA_LOSS_HUBER   2
#define MCBA_LOSS_CAUCHY  3
#define MCBA_LOSS_ARCTAN  4

/* parameter blocks, bit flags for mcba_problem.optimize (calibration.py:28-35 `default_optimize`) */
#define MCBA_OPT_CAMERA_POSES 1
#define MCBA_OPT_BOARD_POSES  2
#define MCBA_OPT_MOTION       4
#define MCBA_OPT_CAMERAS      8
#define MCBA_OPT_BOARDS       16

typedef struct mcba_problem {
  int32_t version;              /* MCBA_VERSION */
  int32_t n_cameras;            /* C */
  int32_t n_frames;             /* F */
  int32_t n_boards;             /* B */
  int32_t n_points;             /* P = max points per board (tables.stack_boards, tables.py:385-394) */

  const double*  points;        /* [C,F,B,P,2] point_table.points as float64, or NULL when points_f32   */
                                /* (last member) carries the table                                     */
  const uint8_t* point_valid;   /* [C,F,B,P]   point_table.valid                                       */
  const uint8_t* inlier_mask;   /* [C,F,B,P]   Calibration.inlier_mask, or NULL -> inliers = valid     */

  const int32_t* board_sizes;   /* [B]   P_b (number of real points of each board)                     */
  const uint8_t* camera_valid;  /* [C]   camera_poses.valid                                            */
  const uint8_t* frame_valid;   /* [F]   motion.valid                                                  */
  const uint8_t* board_valid;   /* [B]   board_poses.valid                                             */

  int32_t motion;               /* MCBA_MOTION_*                                                       */
  int32_t camera_model;         /* MCBA_CAMERA_*                                                       */
  int32_t n_dist;               /* distortion coefficients per camera (camera.dist.size)               */
  const double*  image_heights;    /* [C] camera.image_size[1]  (rolling_frames.py:15-19)              */
  const uint8_t* fix_aspect;       /* [C] Camera.fix_aspect (camera.py:147-148,159-160)                */
  const double*  base_wrt_gripper; /* [F,4,4] hand-eye only (motion/hand_eye.py:43-46), else NULL      */

  uint32_t optimize;            /* OR of MCBA_OPT_*                                                    */
  const double* x_full;         /* all five blocks in reference order, length mcba_full_size();        */
                                /* values of disabled blocks are taken from here                        */
  int32_t frame_begin;          /* frame shard owned by this handle: [frame_begin, frame_end) (may be empty);     */
  int32_t frame_end;            /* frame_begin < 0 = all frames.  Arrays above always describe ALL frames.       */
  const int32_t* camera_n_dist; /* [C] distortion coefficients of EACH camera (camera.dist.size), or NULL: every   */
                                /* camera carries n_dist.  The reference's ParamList holds independent Camera      */
                                /* objects (optimization/parameters.py:54-85, camera.py:144-155), so the cameras   */
                                /* block of x / x_full is ragged: camera c contributes 5 + camera_n_dist[c]        */
                                /* entries.  n_dist must then be the maximum; pinhole sizes (4, 5, 8, 12, 14) mix  */
                                /* freely (a smaller model is the larger one with its extra coefficients at zero). */
  const uint8_t* camera_fisheye;/* [C] 1 = CameraFisheye (camera_fisheye.py:28), 0 = Camera, or NULL: every camera is of  */
                                /* `camera_model`.  A rig may MIX the two families (the reference holds independent     */
                                /* objects, parameters.py:54-85); camera_n_dist is then required unless all carry 4.    */
  const float* points_f32;      /* [C,F,B,P,2] point_table.points as float32 -- the dtype the reference's table has in  */
                                /* use: fill_sparse keeps `values.dtype` (tables.py:15-17) and cv2 detects float32      */
                                /* corners.  Uploaded as it is (half the bytes) and widened on the device, exactly as   */
                                /* numpy promotes it in `reprojected.points - point_table.points`.  Exactly one of      */
                                /* points / points_f32 is non-NULL.                                                     */
} mcba_problem;

typedef struct mcba_options {          /* scipy.optimize.least_squares arguments used at calibration.py:209-210 */
  double ftol;                  /* `tolerance`      (default 1e-4, calibration.py:199)                  */
  double xtol;                  /* scipy default 1e-8                                                   */
  double gtol;                  /* scipy default 1e-8                                                   */
  int32_t max_nfev;             /* `max_iterations` (default 100)                                       */
  int32_t loss;                 /* MCBA_LOSS_*                                                          */
  double f_scale;               /* scipy `f_scale` (soft margin of the robust loss)                     */
  int32_t verbose;              /* 2: per-iteration rows are delivered to the log callback              */
  int32_t tr_solver;            /* MCBA_TR_EXACT (0): exact Schur / Cholesky steps; MCBA_TR_LSMR (1): scipy's own step,     */
                                /* gn_h = lsmr(J_h, f, damp) (trf.py:481), with the Jacobian products on the device.        */
                                /* (Until round 3 this field was `reserved`: ZERO-INITIALISE the struct -- an uninitialised */
                                /* value is rejected with "unknown trust-region solver".  MCBA_TR_LSMR is what reproduces   */
                                /* the reference's end point; the Python drop-in selects it by default.)                     */
} mcba_options;

typedef struct mcba_result {           /* scipy OptimizeResult fields the caller needs                    */
  double cost;                  /* 0.5 * sum(rho(f^2))                                                  */
  double initial_cost;
  double optimality;            /* |g|_inf                                                              */
  int32_t nfev;                 /* residual evaluations (trial steps + 1)                               */
  int32_t njev;                 /* linearisations (fused residual + Jacobian -> normal equations)       */
  int32_t status;               /* scipy status: 0 max_nfev, 1 gtol, 2 ftol, 3 xtol, 4 ftol+xtol         */
  int32_t iterations;
  double solve_seconds;         /* wall time inside mcba_solve                                          */
  double linearize_seconds;     /* GPU time (HIP events) spent in the fused linearisation kernels       */
} mcba_result;

typedef struct mcba_round_report {     /* one pass of Calibration.adjust_outliers (calibration.py:254-268)             */
  double rms_all, rms_inliers;  /* report(): RMS over all valid points / over the inliers                           */
  int64_t n_all, n_inliers;
  double quantiles[5];          /* numpy.quantile(errors over all valid points, [0, .25, .5, .75, 1])               */
  double f_scale;               /* f_scale of this round's solve (auto_scale: quantile x factor)                    */
  double threshold;             /* rejection threshold of this round (quantile x factor), -1 = none                 */
  int64_t n_kept, n_valid;      /* reject_outliers: inliers kept / valid points                                      */
  mcba_result solve;            /* bundle_adjust of this round                                                       */
} mcba_round_report;

typedef struct mcba_handle_s* mcba_handle;

/* iteration log: same columns as scipy's verbose=2 table (print_iteration_nonlinear), which the reference pipes
 * into its logger (calibration.py:208, io/logging.py:53-68).  cost_reduction / step_norm are NaN on row 0.      */
typedef void (*mcba_log_fn)(void* ctx, int32_t iteration, int32_t nfev, double cost, double cost_reduction,
                            double step_norm, double optimality);

/* cross-rank reduction hook for frame-sharded problems: called from inside mcba_solve /
 * mcba_normal_equations with a DEVICE buffer of `count` doubles that must be reduced IN PLACE over all ranks,
 * ordered on `stream`.  op: 0 = sum, 1 = max.  The Python side implements it with torch.distributed
 * (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU-side tests).  Return 0 on success.            */
typedef int32_t (*mcba_allreduce_fn)(void* ctx, void* device_buf, size_t count, int32_t op, void* stream);

/* --- sizes ------------------------------------------------------------------------------------------------- */
/* length of mcba_problem.x_full for the given shape (all five blocks)                                          */
int32_t mcba_full_size(const mcba_problem* p, int64_t* out);

/* --- lifetime ---------------------------------------------------------------------------------------------- */
/* Lowers a Calibration (flat arrays) to device tables: frame-major observation table, inlier prefix sums, index
 * maps.  Replaces the per-call object graph of Calibration.with_param_vec (calibration.py:164-171) and the
 * sparsity build (calibration.py:173-196).  `hip_stream` may be NULL.                                          */
int32_t mcba_create(const mcba_problem* p, void* hip_stream, mcba_handle* out);
int32_t mcba_destroy(mcba_handle h);
/* mcba_destroy parks the handle's device buffers, pinned buffers and stream in a process-wide cache (bounded: 4 GB of device
 * memory) so that the next mcba_create of a problem of the same shape does not pay for ~50 allocations again
 * (Workspace.calibrate creates one handle per call).  This returns everything in the cache to the HIP runtime.         */
int32_t mcba_release_cached_memory(void);
const char* mcba_last_error(void);
/* "gfx950:<device name>" of the device the handle lives on                                                    */
int32_t mcba_device_info(mcba_handle h, char* buf, size_t buf_len);

int32_t mcba_num_params(mcba_handle h, int64_t* n_params);
int32_t mcba_num_residuals(mcba_handle h, int64_t* n_residuals);   /* 2 * #inliers (all frames of the shard)   */

/* Replace the inlier mask (Calibration.reject_outliers -> copy(inlier_mask=...), calibration.py:240-252).
 * mask = NULL restores inliers = valid.                                                                       */
int32_t mcba_set_inliers(mcba_handle h, const uint8_t* mask);

int32_t mcba_set_allreduce(mcba_handle h, mcba_allreduce_fn fn, void* ctx);
/* Native alternative to the callback: RCCL over xGMI driven from inside the library (librccl is loaded at run time).
 * One rank calls mcba_rccl_unique_id and hands the 128 bytes to all ranks (multical_amd.distributed broadcasts them with
 * torch.distributed); then EVERY rank calls mcba_rccl_init (collective).  From then on all reductions of the handle are
 * in-place ncclAllReduce calls on the handle's stream.  mcba_rccl_shutdown leaves the native path again.            */
int32_t mcba_rccl_unique_id(uint8_t* id_out /*[128]*/);
int32_t mcba_rccl_init(mcba_handle h, const uint8_t* id /*[128]*/, int32_t rank, int32_t world);
int32_t mcba_rccl_shutdown(mcba_handle h);
/* version of the librccl the native path binds (ncclGetVersion: major * 10000 + minor * 100 + patch; 0 = library not found or
 * its ABI refused) -- reported per rank by bench.py                                                                     */
int32_t mcba_rccl_version(int32_t* version_out);
/* exactly one rank of a sharded problem is the root: it contributes the replicated (shared) right-hand side to the
 * reduced system.  Default: root.  Ranks other than 0 call this with 0.                                          */
int32_t mcba_set_shard_root(mcba_handle h, int32_t is_root);
/* rank of this handle among the `world` (<= 64) handles of one frame-sharded problem; rank 0 is the root.  Needed with a
 * callback (mcba_set_allreduce); mcba_rccl_init sets it itself.  The messages of a sharded handle are per-rank partial sums
 * gathered by summation (block `rank` of a zeroed buffer) and the SHARED entries of [g | diag]: none grows with the number
 * of frames (mcba_allreduce_stats).                                                                                  */
int32_t mcba_set_shard_rank(mcba_handle h, int32_t rank, int32_t world);
int32_t mcba_set_log(mcba_handle h, mcba_log_fn fn, void* ctx);

/* Calibration.adjust_outliers (calibration.py:254-268; what Workspace.calibrate drives, workspace.py:238-244) in ONE call:
 * `num_adjustments` rounds of {report, optional f_scale = quantile(errors, scale_quantile) * scale_factor, reject_outliers at
 * quantile(errors, outlier_quantile) * outlier_factor, bundle_adjust}, then the final report.  rounds[num_adjustments + 1];
 * a negative factor disables that step; inliers_out (may be NULL) receives the final mask [C,F,B,P].                  */
int32_t mcba_adjust_outliers(mcba_handle h, double* x_inout, const mcba_options* opt, int32_t num_adjustments,
                             double outlier_quantile, double outlier_factor, double scale_quantile, double scale_factor,
                             mcba_round_report* rounds, uint8_t* inliers_out);

/* --- evaluation -------------------------------------------------------------------------------------------- */
/* r = evaluate(x)                                                   (calibration.py:204-206)                   */
int32_t mcba_residuals(mcba_handle h, const double* x, double* r);

/* Analytic Jacobian of `evaluate` in the sparsity pattern of Calibration.sparsity_matrix (calibration.py:173-196):
 * row pair i (one inlier observation) has `row_nnz` structurally non-zero columns (same count for every row);
 * cols[i*row_nnz + k] are ascending column indices into x, vals[(2i+a)*row_nnz + k] the entries of row 2i+a.
 * Call with vals = cols = NULL to query row_nnz.                                                               */
int32_t mcba_jacobian(mcba_handle h, const double* x, int32_t* row_nnz, double* vals, int32_t* cols);

/* Reprojection error per table slot: err[c,f,b,p] = |proj - obs|_2, valid = proj.valid & obs.valid, err = 0 where
 * invalid (tables.reprojection_error, tables.py:244-249; feeds Calibration.reprojection_error / reject_outliers /
 * report, calibration.py:134-141,240-252,290-310).  Arrays are in the reference's [C,F,B,P] order.             */
int32_t mcba_reprojection_error(mcba_handle h, const double* x, double* err, uint8_t* valid);

/* Errors reduced ON THE DEVICE (no [C,F,B,P] array crosses PCIe): n = number of masked points, sum_sq = sum of squared
 * errors, values[i] = exact order statistic of (0-based, ascending) rank ranks[i] -- what error_stats / numpy.quantile
 * need (calibration.py:37-40,304-310).  inliers_only = 0: mask of Calibration.reprojection_error; 1: of
 * reprojection_inliers (calibration.py:138-141).  Sharded handles combine over all ranks.                         */
int32_t mcba_error_stats(mcba_handle h, const double* x, int32_t inliers_only, int32_t n_ranks, const int64_t* ranks,
                         double* values, int64_t* n, double* sum_sq);
/* n of mcba_error_stats known without a device pass (single-GPU handles; -1 for a frame-sharded handle): the caller can
 * derive the quantile ranks first and make ONE mcba_error_stats call                                                  */
int32_t mcba_error_count(mcba_handle h, int32_t inliers_only, int64_t* n);
/* Calibration.reject_outliers on the device (calibration.py:240-252): inliers = (err < threshold) & valid at x;
 * replaces the handle's inlier table.  n_inliers / n_valid (over all ranks) may be NULL.                            */
int32_t mcba_reject_outliers(mcba_handle h, const double* x, double threshold, int64_t* n_inliers, int64_t* n_valid);
/* current inlier table in the reference's [C,F,B,P] order                                                         */
int32_t mcba_get_inliers(mcba_handle h, uint8_t* mask);

/* Projected points [C,F,B,P,2] of Calibration.reprojected (calibration.py:124-130).                            */
int32_t mcba_project(mcba_handle h, const double* x, double* projected);

/* Projected points [C,F,B,P,2] of Calibration.projected (calibration.py:113-119): the projection WITHOUT the measured
 * points, consumed by the GUI / reprojection tables (interface/view_table.py:43-52).  Rolling shutter iterates the scan
 * time from the projected row: 0.5 first, then max_iterations fixed-point passes (motion/rolling_frames.py:115-133,
 * RollingFrames.max_iterations, default 4); the other motion models ignore max_iterations.                           */
int32_t mcba_project_model(mcba_handle h, const double* x, int32_t max_iterations, double* projected);

/* One fused residual+Jacobian evaluation reduced to the normal equations at x (what one scipy `jac` call plus
 * J^T J / J^T f would produce): cost = 0.5 |f|^2 (robust-loss scaled), g = J^T f [n_params],
 * diag = diag(J^T J) [n_params].  Any output pointer may be NULL.  Sharded handles all-reduce through the hook. */
int32_t mcba_normal_equations(mcba_handle h, const double* x, const mcba_options* opt,
                              double* cost, double* g, double* diag);
/* The same evaluation at the x already resident on the device (from the last mcba_normal_equations / mcba_solve),
 * enqueued on the handle's stream without host transfer or synchronisation -- the way the solver itself evaluates.
 * Results stay in HBM; mcba_synchronize waits for the stream.                                                    */
int32_t mcba_normal_equations_device(mcba_handle h, const mcba_options* opt);
int32_t mcba_synchronize(mcba_handle h);
/* dense J^T J [n_params x n_params] assembled from the block form (debug / parity tests; small problems only)  */
int32_t mcba_dense_hessian(mcba_handle h, double* H);

/* --- initialisation tables (the producer of the hot path's inputs) ---------------------------- */
/* matrix.align_transforms_robust (transform/matrix.py:140-153) for a batch of problems: problem p owns the pose pairs
 * [offsets[p], offsets[p+1]) of A and B (row-major 4x4 "points-transforming" matrices), `mask` (or NULL = all) selects
 * the pairs that enter the estimate.  Per problem: relative poses B_k A_k^-1 -> robust mean (Ward clustering of the
 * whitened rotation-vector | translation 6-vectors, most common of max(n/10, 3) clusters, transform/common.py:6-21) ->
 * errors |m A_k - B_k|_F of ALL pairs -> pairs with error < threshold * upper quartile stay -> robust mean again.
 * out[p] = the transform (identity and out_valid[p] = 0 when no pair is selected: tables.relative_between,
 * tables.py:326-332); inliers (or NULL) = pairs that passed the test.  invert != 0: relative_between_inv semantics
 * (inputs inverted, result inverted, tables.py:334-335).  This is the numeric core of tables.estimate_transform
 * (tables.py:153-176; default threshold 1.5) and tables.relative_between_n (tables.py:337-345).                      */
int32_t mcba_align_poses_robust(int32_t n_problems, const int64_t* offsets, const double* A, const double* B,
                                const uint8_t* mask, double threshold, int32_t invert, double* out, uint8_t* out_valid,
                                uint8_t* inliers);
/* The same batch with the pairs given as INDICES into pose tables: entry k of a problem is the pair
 * (table_a[index_a[k]], table_b[index_b[k]]), the tables hold n_a / n_b poses (row-major 4x4; table_a == table_b is one upload).
 * tables.estimate_relative_poses (tables.py:207-227) aligns `np.take(table, i, axis)` with `np.take(table, j, axis)` for
 * every pair of its spanning tree: the pose table [C,F,B] goes up ONCE and the pair lists as 32-bit indices, instead of two
 * gathered copies of it per pair.                                                                                      */
int32_t mcba_align_poses_indexed(int32_t n_problems, const int64_t* offsets, const double* table_a, int64_t n_a,
                                 const int32_t* index_a, const double* table_b, int64_t n_b, const int32_t* index_b,
                                 const uint8_t* mask, double threshold, int32_t invert, double* out, uint8_t* out_valid,
                                 uint8_t* inliers);

/* --- solve -------------------------------------------------------------------------------------------------- */
/* Trust-region least squares: replaces scipy.optimize.least_squares(method='trf', x_scale='jac', jac_sparsity=S,
 * loss, f_scale, ftol, max_nfev) at calibration.py:209-210.  x is updated in place to `res.x`.                 */
int32_t mcba_solve(mcba_handle h, double* x_inout, const mcba_options* opt, mcba_result* result);

/* Collectives issued by a frame-sharded handle since the last reset (observability of SURVEY 8(e)): number of
 * all-reduce calls, doubles moved, and the element counts of the first `cap` calls in issue order (negative = max
 * reduction).  Any of the output pointers may be NULL.                                                           */
int32_t mcba_allreduce_stats(mcba_handle h, int32_t reset, int64_t* calls, int64_t* doubles, int64_t* sizes, int32_t cap,
                             int32_t* n_sizes);

/* --- measurement ------------------------------------------------------------------------------------------- */
/* Run the fused linearisation `repeats` times at x and return the average GPU time per pass in milliseconds
 * measured with HIP events on the handle's stream (bench.py roofline leg).                                     */
int32_t mcba_time_linearize(mcba_handle h, const double* x, const mcba_options* opt, int32_t repeats,
                            double* avg_ms);
int32_t mcba_time_residuals(mcba_handle h, const double* x, int32_t repeats, double* avg_ms);
/* ... and of the two kernels of one LSMR iteration of the default solver (tr_solver = MCBA_TR_LSMR): ms[0] = the product kernel
 * (J_h v and J_h^T u from one evaluation of the analytic rows), ms[1] = the gather; bench.py: parity_route.roofline             */
int32_t mcba_time_lsmr_iteration(mcba_handle h, const double* x, int32_t repeats, double* ms /*[2]*/);

#ifdef __cplusplus
}
#endif
#endif /* MCBA_H */

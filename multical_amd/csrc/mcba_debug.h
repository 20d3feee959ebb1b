/* mcba_debug.h -- test / profiling entry points of libmcba.so.  NOT part of the drop-in boundary (include/mcba.h): nothing
 * in multical binds these; tests/ and the profiling scripts use them to validate and time individual kernels.          */
#ifndef MCBA_DEBUG_H
#define MCBA_DEBUG_H
#include "../../include/mcba.h"
#ifdef __cplusplus
extern "C" {
#endif

/* 1 = accumulate V^T V and the Schur SYRK with v_mfma_f64_16x16x4_f64 (default); 0 = identical data flow with plain
 * FMAs (validation build of the same kernels).  The environment variable MCBA_NO_MFMA=1 sets the initial value.  */
int32_t mcba_set_mfma(mcba_handle h, int32_t on);
/* number of persistent k_linearize workgroups; 0 = automatic (profiling aid)                                        */
int32_t mcba_debug_set_lin_grid(mcba_handle h, int32_t grid);
/* experiment: views of a frame bound to nw waves of "its" workgroup (load-balance bound of a frame-level linearisation);
 * nw = 0 restores the largest-first list of views                                                                     */
int32_t mcba_debug_set_frame_groups(mcba_handle h, int32_t nw);
/* FP64 VALU vs FP64 MFMA pipe-sharing probe: ms_out[3] = all-FMA, all-MFMA, half / half        */
int32_t mcba_debug_pipe_probe(int32_t iters, double* ms_out);
/* [sum a b | sum a^2 | sum b^2] by the single-workgroup k_dot and by its 16-CU form (k_dot3_part / k_dot3_fin): bit-identical by design */
int32_t mcba_debug_dot3(const double* a, const double* b, int64_t n, double* out_single, double* out_wide);
/* workgroup dispatch rate: launches `blocks` workgroups of `threads` threads with `lds_bytes` of LDS, each running `spin`
 * dependent FMAs; out[blocks][2] = 100 MHz wall-clock ticks at the start / end of every workgroup                   */
int32_t mcba_debug_dispatch_probe(int32_t blocks, int32_t threads, int32_t lds_bytes, int32_t spin, long long* out);
int32_t mcba_debug_xcd_probe(int32_t n, long long* out);
/* regularised Gauss-Newton direction (H_h + reg I)^-1 g_h in the column-scaled space, computed by the Schur /
 * Cholesky kernels after a preceding mcba_normal_equations at the same x; g_h and scale_inv may be NULL.           */
int32_t mcba_debug_gn_step(mcba_handle h, double reg, double* gn_h, double* g_h, double* scale_inv);
/* per-view s_memtime stamps of the k_linearize phases: out[views][8] = {setup, rows, stage+mfma, epilogue, count,
 * start, end, 0} in shader cycles (profiling aid for DESIGN.md section 5), followed by ceil(views / 8) rows of k_tmat
 * workgroup stamps (zero unless the library was built with -DMCBA_EXP_TMAT_PROF)                                      */
int32_t mcba_debug_linearize_profile(mcba_handle h, const double* x, long long* out);
/* (S + reg I) p = rhs with the device Cholesky kernels; blocked != 0 forces the multi-workgroup path              */
int32_t mcba_debug_chol(mcba_handle h, int32_t ns, const double* S, const double* rhs, double reg, int32_t blocked,
                        double* p_out);
/* one v_mfma_f64_16x16x4_f64 on V = [A | B] (4 x 32, row-major): out[16][16] = A^T B (operand-layout self-test)     */
int32_t mcba_debug_mfma_probe(const double* V, double* out);

/* the matrix-free Jacobian products of the lsmr mode (k_lsmr_jv / k_lsmr_jtu / k_lsmr_gather) at x, unscaled columns, linear
 * loss: jv_out[m] = J(x) v (reference residual order), jtu_out[n] = J(x)^T u; either pair may be NULL (tests compare them with
 * mcba_jacobian)                                                                                                      */
int32_t mcba_debug_lsmr_products(mcba_handle h, const double* x, const double* v, const double* u, double* jv_out, double* jtu_out);
/* experiment / path-forcing switches (formerly MCBA_* environment variables: the product library no longer reads those, a
 * MCBA_BUILD_VARIANT build does): process-wide, once per name, before the first mcba_create                               */
int32_t mcba_debug_set_switch(const char* name, const char* value);
/* LSMR iteration: -1 (default) = automatic: 3 on static / hand-eye rigs, 2 with rolling shutter or boards=True; 3 = two launches with the
 * per-observation state (A, X_start, X_end, t) stored by the first iteration of a solve and streamed back by the others; 2 = two launches (k_lsmr_fused2: both Jacobian products from one evaluation of the rows + the scalar
 * recurrence / vector update of the previous step in its head; k_lsmr_gather3), 1 = three launches, 0 = the six-launch form    */
int32_t mcba_debug_set_lsmr_fused(mcba_handle h, int32_t on);
/* ... and through the kernels the default solver iterates with (k_lsmr_fused2 + k_lsmr_gather3, one Golub-Kahan step with alpha = 0):
 * jv_out[m] = J(x) v, jtjv_out[n] = J(x)^T J(x) v                                                                               */
int32_t mcba_debug_lsmr_fused_products(mcba_handle h, const double* x, const double* v, double* jv_out, double* jtjv_out);
/* LSMR iterations taken by the last mcba_solve with tr_solver = MCBA_TR_LSMR on this handle                           */
int32_t mcba_debug_lsmr_info(mcba_handle h, int64_t* lsmr_iterations);

/* ONE call of the device's LSMR solve (lsmr_solve: scipy.sparse.linalg.lsmr(J_h, f, damp, atol = btol = 1e-6) of trf.py:481) on the
 * linearisation at x with scipy's Jacobian scaling of a first iterate (or scale_in[n], the scaling of a later iterate; maxiter > 0: scipy's `maxiter`): gn_h_out[n] = the solution, scale_out[n] (may be NULL) = d with
 * J_h = J diag(d), out[8] = {istop, itn, normr, normar, normA, condA, normx, normb} -- compared with scipy's own lsmr on
 * mcba_jacobian's matrix by tests/test_gpu_lsmr.py::test_device_lsmr_call_matches_scipy                                       */
int32_t mcba_debug_lsmr_solve(mcba_handle h, const double* x, const mcba_options* opt, double damp, const double* scale_in, int32_t maxiter,
                              double* gn_h_out, double* scale_out, double* out);
/* the LSMR calls of the last lsmr-mode mcba_solve: rows[cap][10] = {trust-region iteration, damp, Delta, istop, itn, normr, normar,
 * normA, condA, normx} (the last five NaN unless mcba_debug_set_lsmr_trace(h, 1) preceded the solve)                            */
int32_t mcba_debug_lsmr_trace(mcba_handle h, int32_t cap, double* rows, int32_t* n_rows);
int32_t mcba_debug_set_lsmr_trace(mcba_handle h, int32_t scalars);
/* 1: the product kernel of the LSMR iteration reads the frame-major tables (masks compacted per view) on every rig instead of the
 * compacted observation tables (A/B runs, tests)                                                                                  */
int32_t mcba_debug_set_lsmr_masks_form(mcba_handle h, int32_t on);
/* number of collective sizes mcba_allreduce_stats records per handle (default 4096; a whole sharded lsmr solve issues more)       */
int32_t mcba_debug_set_allreduce_trace(mcba_handle h, int32_t cap);
/* persistent workgroups of the LSMR product kernels (default 2048): summation-order experiments                                 */
int32_t mcba_debug_set_lsmr_grid(mcba_handle h, int32_t grid);

#ifdef __cplusplus
}
#endif
#endif /* MCBA_DEBUG_H */

/*
 * graphlearn_amd.h -- C ABI of the MI355X-native graph-learning hot path.
 *
 * The reference (LTS4/graph-learning) is pure Python and has no FFI layer; the
 * entry points below are what a ctypes binding for its hot path binds to.  Each
 * one cites the reference interface it replaces (paths relative to the
 * reference repository root).  See INTEGRATION.md for the reference-side stub.
 *
 * Conventions
 *   - every function returns an int status: 0 = GL_OK, < 0 = error (below);
 *   - all host buffers are caller-owned, C-contiguous, float64 unless stated,
 *     valid for the duration of the call only; nothing returned is
 *     library-allocated except the opaque handles;
 *   - edge vectors use SciPy "condensed" order: edge (i,j), i<j, at
 *     i*(2n-i-1)/2 + (j-i-1)   (graph_learn/operators.py:39-41);
 *   - a gl_ctx is bound to one GPU and one HIP stream and is not thread-safe;
 *     distinct contexts are independent (one problem per GPU, one thread or
 *     process per context);
 *   - there is NO CPU fallback: without a gfx950 device gl_ctx_create fails.
 */
#ifndef GRAPHLEARN_AMD_H
#define GRAPHLEARN_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GL_VERSION 100 /* 0.1.0 */

enum {
    GL_OK = 0,
    GL_E_INVALID = -1,   /* bad argument            -> Python ValueError   */
    GL_E_NOMEM = -2,     /* device/host allocation  -> Python MemoryError  */
    GL_E_HIP = -3,       /* HIP runtime error       -> Python RuntimeError */
    GL_E_NONFINITE = -4, /* NaN/Inf met where the algorithm cannot go on   */
    GL_E_NODEVICE = -5,  /* no usable gfx950 device                         */
    GL_E_COMM = -6,      /* RCCL error                                      */
    GL_E_STATE = -7      /* call sequence error (e.g. iterate before init)  */
};

typedef struct gl_ctx gl_ctx;
typedef struct gl_logdeg gl_logdeg;
typedef struct gl_comm gl_comm;

/* ---- library / context ------------------------------------------------- */
int gl_version(void);
const char *gl_strerror(int code);
int gl_device_count(int *count);
int gl_ctx_create(int device, gl_ctx **out);
int gl_ctx_destroy(gl_ctx *ctx);
const char *gl_last_error(gl_ctx *ctx);
int gl_ctx_device_name(gl_ctx *ctx, char *buf, int buflen);
int gl_ctx_sync(gl_ctx *ctx);
/* Gives back the device memory the context retains between calls for speed: the pool of released session
 * blocks, the batched solver's stand-in session with its ring of iterates, the scratch of the batched entry
 * points.  The library does this itself before it reports GL_E_NOMEM from any of its own allocations; call it
 * on a shared device when a long-lived context should shrink.  No reference counterpart (NumPy arrays are
 * garbage-collected, smooth_learning.py:286-300 keeps only theta_ / weights_). */
int gl_ctx_trim(gl_ctx *ctx);
/* HIP-event timer on the context's stream (bench.py's timed region). */
int gl_ctx_timer_start(gl_ctx *ctx);
int gl_ctx_timer_stop(gl_ctx *ctx, double *elapsed_ms);

/* ---- log-degree solver session (LogModel hot path) ----------------------
 * Replaces the body of gsp_learn_graph_log_degrees
 * (graph_learn/smooth_learning.py:47-174) and LogModel._initialize / fit
 * (:277-300), split into stages so that inputs can be made resident in HBM
 * before a timed region:
 *   create -> set_signals | set_distances -> init -> iterate* -> get_weights_*
 */
int gl_logdeg_create(gl_ctx *ctx, int64_t n_nodes, gl_logdeg **out);
int gl_logdeg_destroy(gl_logdeg *s);

/* A1+A2: squared pairwise distances between the COLUMNS of x (d x n,
 * row-major) via an fp64 MFMA Gram contraction, then theta for `avg_degree`
 * (get_theta, smooth_learning.py:201-219; avg_degree <= 0 means theta = 1,
 * :279-280), then z*theta is laid out in the solver's tiled HBM layout.
 * theta_out may be NULL: theta is then fetched later with gl_logdeg_theta (for n <= 128 it never
 * leaves the device before the distances are scaled, so a batch of windows is set up without a
 * host round trip each). */
int gl_logdeg_set_signals(gl_logdeg *s, const double *x, int64_t d, int64_t n_nodes,
                          int avg_degree, double *theta_out);
/* theta of the last set_signals (1 after set_distances). */
int gl_logdeg_theta(gl_logdeg *s, double *theta_out);
/* Distances given directly (condensed, already scaled), as the free function
 * of the reference takes them (smooth_learning.py:48). */
int gl_logdeg_set_distances(gl_logdeg *s, const double *z_condensed);
/* Download what set_signals computed: condensed squared distances BEFORE the
 * theta scaling (pdist(x.T)**2, smooth_learning.py:278). */
int gl_logdeg_get_distances(gl_logdeg *s, double *z_condensed);

/* A4+A5: gamma = step_size/(2 beta + sqrt(2(n-1))) (:135-137); w = edge_init
 * (condensed, may be NULL = zeros) and dvar = S w (:140-144). */
int gl_logdeg_init(gl_logdeg *s, double alpha, double beta, double step_size,
                   const double *edge_init);

/* A6-A9: run up to `maxit` further iterations of the loop (:146-167); stops
 * early exactly like the reference when both relative changes < tol (never at
 * the session's first iteration).  n_iter_out = iterations done by THIS call;
 * converged_out = 1 if the stopping rule fired; rel_out[2] = last relative
 * primal/dual change.  Any out pointer may be NULL. */
int gl_logdeg_iterate(gl_logdeg *s, int64_t maxit, double tol, int64_t *n_iter_out,
                      int *converged_out, double *rel_out);
/* Same loop, enqueued without any host synchronisation or convergence poll
 * (tol = 0 semantics): used by bench.py inside its event-timed region. */
int gl_logdeg_iterate_async(gl_logdeg *s, int64_t n_iter);

/* A10: w[w < edge_tol] = 0 (:169) applied on the way out (the resident
 * iterate is not modified); condensed (E) or symmetric square (n x n, zero
 * diagonal, = squareform(w), :171-172).  edge_tol = -inf disables it.
 * get_weights_condensed(w_out = NULL) only stages the vector on the device. */
int gl_logdeg_get_weights_condensed(gl_logdeg *s, double edge_tol, double *w_out);
/* The same for B sessions (same context, same n_nodes) with one copy and one synchronisation: w_out is
 * B x E, row b = session b (the reference's edge_w.T of the batched branch, smooth_learning.py:174).  Thetas that
 * set_signals left on the device (theta_out = NULL) come back with the same synchronisation: gl_logdeg_theta
 * of these sessions returns at once afterwards. */
int gl_logdeg_batch_get_weights(gl_logdeg **sessions, int n_sessions, double edge_tol, double *w_out);
int gl_logdeg_get_weights_square(gl_logdeg *s, double edge_tol, double *w_out);
/* Dual variable dvar (n) -- for checkpoint/resume and the parity tests. */
int gl_logdeg_get_dual(gl_logdeg *s, double *d_out);
int64_t gl_logdeg_total_iterations(gl_logdeg *s);

/* Measurement facility (bench.py): average duration in microseconds of the edge pass alone,
 * HIP events on the solver's stream over `reps` back-to-back launches; session state unchanged. */
int gl_logdeg_time_edge_pass(gl_logdeg *s, int reps, double *avg_us_out);

/* Which iteration engine serves long gl_logdeg_iterate / _async calls of this session (after
 * gl_logdeg_init): 0 = streaming (one edge-pass + one node-pass launch per iteration, iterate in
 * HBM; n >~ 11 400 and calls shorter than 4 iterations), 1 = resident (one persistent launch per
 * call: the tile engine -- iterate in registers, distances in LDS, tagged hand-offs between
 * workgroups instead of grid barriers, rounds of streamed tiles past n = 4096 -- for
 * 129 <= n <~ 11 400; round 1's grid-barrier engine only on request), 2 = small (n <= 128: one
 * workgroup, the whole problem in LDS / registers, one launch per call).
 * GRAPHLEARN_AMD_ENGINE = stream | tile | resident (= tile where it applies, else round 1's
 * engine) | resident1 (round 1's engine) | small forces one.  Same iterates up to summation order. */
int gl_logdeg_engine(gl_logdeg *s);

/* Batched (2-D) path of gsp_learn_graph_log_degrees (smooth_learning.py:128-129,158-159):
 * with (B, E) input the reference's norms are spectral norms of (E x B) / (n x B) matrices and
 * all B graphs stop together.  This returns the four B x B Gram matrices of the last iteration
 * over B sessions that advanced in lockstep (out: 4 x B x B =
 * [<dw_a,dw_b>, <w_old_a,w_old_b>, <dd_a,dd_b>, <d_old_a,d_old_b>]); the caller takes
 * sqrt(lambda_max) of each for the stopping rule. */
int gl_logdeg_batch_gram(gl_logdeg **sessions, int B, double *out);
/* The whole coupled loop of the 2-D branch (smooth_learning.py:146-167 with :158-159 spectral norms):
 * B initialised sessions (same context, same n, same iteration count) advance up to maxit iterations in
 * lockstep and stop together; rel_out[2] = the two spectral-norm ratios of the last iteration. */
int gl_logdeg_batch_iterate(gl_logdeg **sessions, int B, int64_t maxit, double tol, int64_t *n_iter_out,
                            int *converged_out, double *rel_out);

/* One-shot forms (what the reference-side binding calls). */
/* gsp_learn_graph_log_degrees(distances, alpha, beta, edge_init, maxit, tol,
 * step_size, edge_tol) for 1-D input -> condensed weights (E). */
int gl_log_degrees_fit(gl_ctx *ctx, const double *z_condensed, int64_t n_nodes, double alpha,
                       double beta, const double *edge_init, int64_t maxit, double tol,
                       double step_size, double edge_tol, double *w_out_condensed,
                       int64_t *n_iter_out, double *rel_out);
/* LogModel(avg_degree, edge_init, maxit, tol, step_size, edge_tol).fit(x):
 * x is (d, n) row-major; weights_out is n x n; alpha = beta = 1 (:289-298). */
int gl_logmodel_fit_signals(gl_ctx *ctx, const double *x, int64_t d, int64_t n_nodes,
                            int avg_degree, const double *edge_init, int64_t maxit, double tol,
                            double step_size, double edge_tol, double *weights_out,
                            double *theta_out, int64_t *n_iter_out, double *rel_out);

/* ---- Laplacian <-> edge-vector operators (graph_learn/operators.py) ------
 * Batched over the leading axis B; host buffers in and out. */
/* square_to_vec (:39-41): (B,n,n) -> (B,E) strict upper triangle. */
int gl_square_to_vec(gl_ctx *ctx, const double *x, int64_t B, int64_t n_nodes, double *out);
/* laplacian_squareform_vec (:58-70) / laplacian_squareform (:44-55):
 * (B,E) edge weights -> (B,n,n) Laplacians L = D - W. */
int gl_laplacian_from_vec(gl_ctx *ctx, const double *w, int64_t B, int64_t n_nodes, double *out);
/* Adjoint (L* Y)_(i,j) = Y_ii + Y_jj - Y_ij - Y_ji, (B,n,n) -> (B,E).
 * order 0: summation order of laplacian_squareform_adj(_vec) (:73-94);
 * order 1: summation order of op_adj_dual (:176-180). */
int gl_laplacian_adjoint(gl_ctx *ctx, const double *y, int64_t B, int64_t n_nodes, int order, double *out);
/* squared_pdiffs (:224-226): (T,n) signals -> (T,E), (x_ti - x_tj)^2. */
int gl_squared_pdiffs(gl_ctx *ctx, const double *x, int64_t T, int64_t n_nodes, double *out);
/* get_theta, blocks=None (graph_learn/smooth_learning.py:201-219) on a host
 * (n,n) matrix of squared distances (zero diagonal included in the ranking). */
int gl_theta_from_distances(gl_ctx *ctx, const double *zsq, int64_t n_nodes, int avg_degree, double *theta_out);
/* The same on a rows x cols matrix: the blocks branch of get_theta (:220-235) applies the plain
 * branch to rectangular sub-matrices sq_pdists[block_a, block_b]; needs 1 <= avg_degree < cols. */
int gl_theta_from_matrix(gl_ctx *ctx, const double *zsq, int64_t rows, int64_t cols, int avg_degree,
                         double *theta_out);

/* ---- batch of graphs across GPUs (BASELINE config 4) ----------------------
 * One independent problem per GPU/rank; RCCL over xGMI is used only as a barrier,
 * a scalar max-reduction (timing) and the final gather of the learned adjacencies.
 * The reference has no counterpart (it has no communication layer).  Bootstrap:
 * rank 0 calls gl_comm_unique_id and ships the 128 bytes to the other ranks by any
 * host channel; then every rank calls gl_comm_create. */
int gl_comm_unique_id(void *out128);
int gl_comm_create(gl_ctx *ctx, int rank, int nranks, const void *unique_id128, gl_comm **out);
/* A collective completes only when every rank enters it.  So that a failing rank cannot hang the others (and the
 * node): every wait of the calls below is bounded by the wall clock (`seconds`, default 300, or the environment
 * variable GRAPHLEARN_AMD_COMM_TIMEOUT_S at creation) and returns GL_E_COMM when a peer never arrives; a rank
 * that meets a local error on its way INTO a collective (allocation, copy, the solver's staging) aborts the
 * communicator before it returns that error, as gl_comm_abort does for a caller that cannot go on.  After
 * either, every later call on this communicator fails at once with GL_E_COMM: the process should exit non-zero. */
int gl_comm_set_timeout(gl_comm *c, double seconds);
int gl_comm_abort(gl_comm *c);
int gl_comm_destroy(gl_comm *c);
int gl_comm_barrier(gl_comm *c);
int gl_comm_allreduce_max(gl_comm *c, double *value);
/* All-gather of the condensed, thresholded weights (E doubles per rank) straight
 * from device memory; host_out is (nranks, E) or NULL. */
int gl_comm_allgather_weights(gl_comm *c, gl_logdeg *s, double edge_tol, double *host_out);
/* All-gather of a small host-resident fp64 vector: count doubles per rank in, (nranks, count) out. */
int gl_comm_allgather_f64(gl_comm *c, const double *host_in, int64_t count, double *host_out);
/* The gather payload: one self-describing record per rank -- the 64-byte header of
 * graph_learning_amd/wire.py (magic, version, n_nodes, n_edges, problem id, iterations, theta, edge_tol,
 * flags) followed by the E condensed, thresholded weights -- gathered device-to-device; host_out is
 * (nranks, 8 + E) doubles or NULL.  flags: bit 0 = stopped on the tolerance, bit 1 = this rank FAILED to
 * solve its problem (it still takes part, so nobody waits for it, and every rank reads the failure from the
 * record).  The reference keeps results as ndarray attributes (smooth_learning.py:299-300: theta_,
 * weights_); this is their wire form. */
int gl_comm_allgather_records(gl_comm *c, gl_logdeg *s, double edge_tol, int64_t problem_id, int64_t n_iter,
                              double theta, int flags, double *host_out);

/* ---- Laplacian-constrained GMRF (graph_learn/lgmrf.py, graph_learn/evaluation.py) ---- */
/* relative_error(y_true, y_pred) (evaluation.py:8-25): ||pred-true||_F / ||true||_F, or the
 * plain norm when y_true is allclose to 0. */
int gl_relative_error(gl_ctx *ctx, const double *y_true, const double *y_pred, int64_t count, double *out);
/* nonnegative_qp_solver(mat, vec, step, tol, max_iter) (lgmrf.py:19-71): ADMM for
 * min 1/2 b'Qb - b'p, b >= 0; mat is m x m, z_out is m; iters_out = ADMM iterations run. */
int gl_nonneg_qp_solve(gl_ctx *ctx, const double *mat, const double *vec, int64_t m, double step, double tol,
                       int max_iter, double *z_out, int *iters_out);
/* LGMRF(norm_par, prob_tol, max_cycle, qp_step, qp_tol, regularization_type, adj_mask).fit(x) for the
 * generalized Laplacian set (lgmrf.py:109-235).  x is n x n (covariance / kernel, not modified);
 * adj_mask is n x n or NULL (fully connected); outputs: laplacian_, inv_laplacian_ (n x n each),
 * converged_ (cycle index or -1) and the number of cycles run.
 * path: 0 automatic, 1 general (explicit inverse per node), 2 fast (needs an all-ones mask). */
int gl_lgmrf_fit(gl_ctx *ctx, const double *x, int64_t n_nodes, double norm_par, double prob_tol, int max_cycle,
                 double qp_step, double qp_tol, int regularization_type, const double *adj_mask, int path,
                 double *laplacian_out, double *inv_laplacian_out, int *converged_out, int *cycles_out);
/* Measurement only (no reference counterpart; bench.py's LGMRF leg): seconds the node updates first_node ..
 * first_node + count - 1 of the FIRST block-coordinate cycle (lgmrf.py:175-219) take on the device, between HIP
 * events on the context's stream.  Nothing is returned but the time; gl_lgmrf_fit itself cannot be truncated. */
int gl_lgmrf_time_node_updates(gl_ctx *ctx, const double *x, int64_t n_nodes, double norm_par, double qp_step,
                               double qp_tol, int regularization_type, const double *adj_mask, int path,
                               int64_t first_node, int64_t count, double *seconds_out);

/* ---- graph dictionary with log-barrier degrees (graph_learn/dictionary/log_model.py:14-50 over
 * dictionary/base_model.py:355-676) -- SURVEY.md section 8 row f-1 ---------------------------
 * K atoms (graphs, condensed weights K x E), coefficients K x T (constant over windows of
 * window_size samples), one dual vector per window (T/window x n).  A session owns the device
 * state; the (T x E) matrix of squared pairwise differences of the reference is never formed.
 * n_samples is the padded sample count (a multiple of window_size, base_model.py:603-608);
 * n_atoms <= 16. */
typedef struct gl_graphdict gl_graphdict;
int gl_graphdict_create(gl_ctx *ctx, int64_t n_nodes, int64_t n_samples, int n_atoms, int window_size,
                        gl_graphdict **out);
void gl_graphdict_destroy(gl_graphdict *s);
/* x: (n_samples, n_nodes) row-major. */
int gl_graphdict_set_signals(gl_graphdict *s, const double *x);
/* coefficients_ (K x T), weights_ (K x E), dual_ (T/window x n, NULL = zeros) as the reference
 * holds them after _initialize (base_model.py:313-333) or after any step. */
int gl_graphdict_set_state(gl_graphdict *s, const double *coefficients, const double *weights, const double *dual);
int gl_graphdict_get_state(gl_graphdict *s, double *coefficients, double *weights, double *dual);
/* l1_w, ortho_w, l1_c, log_c, l1_diff_c, step_c, step_w, step_dual (base_model.py:95-117). */
int gl_graphdict_set_params(gl_graphdict *s, double l1_w, double ortho_w, double l1_c, double log_c,
                            double l1_diff_c, double step_c, double step_w, double step_dual);
/* Precision of the two O(K T E) contractions of a step (`self.weights_ @ sq_pdiffs.T`, base_model.py:375, and
 * `self.coefficients_ @ sq_pdiffs`, :447): 64 (default, what the reference computes) or 32 (BASELINE configs[4]:
 * fp32 operands and partial sums, packed two per instruction; folds, projections, dual prox and state stay fp64). */
int gl_graphdict_set_precision(gl_graphdict *s, int bits);
/* current (step_c, step_w, step_dual): reduce_step_on_plateau shrinks them (base_model.py:571-573) */
int gl_graphdict_get_steps(gl_graphdict *s, double *steps3_out);
/* Which form fit / predict will take for this session: 1 = the whole loop in one launch of one workgroup
 * (state in LDS; small problems such as the reference's tests/test_models.py shape), 0 = one captured graph
 * of kernels per step.  GRAPHLEARN_AMD_GD_ENGINE = launch | small overrides the choice. */
int gl_graphdict_engine(gl_graphdict *s);
/* The loop of _single_fit (base_model.py:552-581): up to max_iter steps of _fit_step (:485-527);
 * history_out is (max_iter, 2) = relative change of coefficients / weights per step (-1 where not
 * run); converged_out = converged_ (step index or -1; the loop stops only at an index > 0, :578);
 * flags_out: bit 0 "All coefficients dropped to 0" (:406-408), bit 1 step sizes were reduced. */
int gl_graphdict_fit(gl_graphdict *s, int64_t max_iter, double tol, int reduce_step_on_plateau,
                     double *history_out, int64_t *steps_run_out, int64_t *converged_out, int *flags_out);
/* predict (base_model.py:647-676): weights fixed, coefficients start from the session's state,
 * dual from zero; coefficients_out is (K x T). */
int gl_graphdict_predict(gl_graphdict *s, int64_t max_iter, double tol, double *coefficients_out,
                         int64_t *steps_run_out, int *flags_out);
/* Measurement facility: average microseconds of the coefficient-gradient and the weight-gradient
 * kernel alone (HIP events on the session's stream, `reps` launches each); state unchanged. */
int gl_graphdict_time_gradients(gl_graphdict *s, int reps, double *grad_c_us_out, double *grad_w_us_out);

#ifdef __cplusplus
}
#endif
#endif /* GRAPHLEARN_AMD_H */

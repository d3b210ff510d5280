// Internal: the log-degree solver session and the kernels' parameter block.
#pragma once
#include "gl_common.h"

// Device-side view of one problem (passed by value to the kernels; nothing in it
// changes from one iteration to the next, so a captured hipGraph can be replayed).
struct LogdegDev {
    double *w[2];          // edge iterate, tiled layout, ping-pong: iterate `it` lives in w[it & 1]
    const double *z;       // scaled squared distances, tiled layout
    double *dvar;          // dual variable d            (n_pad)
    double *dprev;         // d of the previous iteration (for the coupled stopping rule of the batched path)
    double *ybar;          // forward dual point  ybar   (n_pad)
    double *pbar;          // prox dual point     pbar   (n_pad)
    double *rowslab_p;     // row-degree partials of p:     [J][node i],  J >= I(i)      (nb x n_pad)
    double *rowslab_w;     // row-degree partials of w_new
    double *colslab;       // column-degree partials, one entry per (wave, tile) segment: [n_segments][2][64],
                           // entries of one column block J are contiguous
    const int *col_ptr;    // column block J -> [col_ptr[J], col_ptr[J+1]) colslab entries   (nb + 1)
    const int *seg_dest;   // (wave, segment) -> colslab entry                                (G * kmax)
    double *tnorm;         // per-wave {sum (q-y)^2, sum w^2}               [G][2]
    double *bnorm;         // per-node-block norm partials                  [2][nb][4]
    double *dstat;         // {rel_primal, rel_dual, tol}
    long long *ctr;        // {done, final_iter, it_edge, it_node}
    unsigned long long *dbg;  // development only (tools/edge_ablate.hip): per-wave cycle stamps, else null
    int64_t n, nb, n_pad, ntiles;
    int64_t npairs;        // ntiles * 32 row pairs = the flat work list
    int G;                 // persistent edge grid (single-wave workgroups)
    int kmax;              // segments reserved per wave in seg_dest
    double gamma;          // scaled step                      (smooth_learning.py:135-137)
    double c;              // 1 - gamma*2*beta
    double g2;             // 2*gamma
    double a4g;            // 4*alpha*gamma
};

struct gl_resident;
struct gl_tile;

struct gl_logdeg {
    gl_ctx *ctx = nullptr;
    LogdegDev dev{};
    int64_t n = 0, n_edges = 0;
    void *arena = nullptr;        // the one allocation every fixed-size buffer of the session is carved from
    void *sig_arena = nullptr;    // zsq, theta_terms and xdev (set_signals only), one allocation too
    size_t arena_bytes = 0, sig_bytes = 0;  // (both blocks go back to the context's pool: gl_ctx_release)
    double *zsq = nullptr;        // n x n squared distances (set_signals only)
    double *stage = nullptr;      // device staging for condensed host transfers (E doubles)
    double *stage_rec = nullptr;  // = stage - 8 doubles: room for the 64-byte record header of the gather payload
    double *xdev = nullptr;       // signals (d x n) on device, followed by their centred copy (another d x n)
    int64_t xdev_elems = 0;
    double *theta_terms = nullptr;  // 2 x n per-row terms of get_theta
    double *gram_diag = nullptr;    // n Gram diagonal entries (scratch of the distance construction)
    double *square_out = nullptr;   // n x n scratch of get_weights_square (allocated at its first use, kept)
    int *col_ptr = nullptr, *seg_dest = nullptr;
    int n_segments = 0;
    bool have_z = false, inited = false;
    bool have_zsq = false;
    int64_t it = 0;               // iterations completed (iterate lives in w[it & 1])
    int64_t init_version = 0;     // bumped by every gl_logdeg_init (engines that cache what depends on z and gamma compare it)
    double theta = 1.0;
    bool theta_pending = false;   // set_signals left theta on the device (dstat[4]): gl_logdeg_theta fetches it
    // captured batch of iterations (launch-bound inner loop -> one graph launch)
    hipGraphExec_t graph = nullptr;
    int graph_iters = 0;
    int wpc = 8;                  // edge waves per CU
    gl_resident *res = nullptr;   // first resident engine (gl_resident.hip: grid barriers), null when not applicable
    gl_tile *tile = nullptr;      // tile engine (gl_tile.hip: tagged hand-offs), preferred when it applies
    bool engines_ready = false;   // gl_resident_prepare / gl_tile_prepare have run (at the first own iterate call)
    bool res_unchecked = false;   // an asynchronous resident launch has not had its error word read yet
    int64_t res_pending = 0;      // ... and this many iterations were asked of the unchecked launches together
    double tol_dev = 0.0;         // the tolerance word the streaming kernels see (dstat[2]) ...
    bool tol_set = false;         // ... once it has been uploaded
};

// gl_gram.hip
int gl_gram_distances(gl_ctx *ctx, const double *x_dev, double *xc_dev, int64_t d, int64_t n, double *zsq_dev,
                      double *diag_dev);
// gl_theta.hip
int gl_theta_from_square(gl_ctx *ctx, const double *zsq_dev, int64_t n, int k, double *terms_dev,
                         double *theta_out);
int gl_theta_from_square_dev(gl_ctx *ctx, const double *zsq_dev, int64_t n, int k, double *terms_dev, double *theta_dev);
int gl_theta_from_rows(gl_ctx *ctx, const double *zsq_dev, int64_t rows, int64_t cols, int k, double *terms_dev,
                       double *theta_out);
// gl_resident.hip
int gl_resident_prepare(gl_logdeg *s);
void gl_resident_free(gl_logdeg *s);
int gl_resident_launch(gl_logdeg *s, int64_t iters, double tol);
int gl_resident_finish(gl_logdeg *s, int64_t *iters_done, bool *stopped, bool *not_resident, bool *gated_out = nullptr);
// gl_tile.hip
constexpr int GL_TILE_MIN_CALL = 4;  // shortest call (iterations) the tile engine takes; shorter ones stream
int gl_tile_prepare(gl_logdeg *s);
int gl_tile_prepare_batch(gl_logdeg *s, int B);  // s stands for B graphs of dev.n nodes (gl_logdeg.hip: the batch's cruise)
int gl_tile_batch_make_zg(gl_logdeg *s, const LogdegDev *devs_dev, int B);
void gl_tile_batch_set_ring(gl_logdeg *s, double *ring, int64_t ring_stride, double *dring, int64_t dring_stride);
void gl_tile_free(gl_logdeg *s);
int gl_tile_launch(gl_logdeg *s, int64_t iters, double tol, bool norms, int final_parity, double gate = 0.0);
bool gl_tile_writes_final_parity(const gl_logdeg *s);
void gl_tile_dump_stamps(gl_logdeg *s);
// gl_small.hip
bool gl_small_applicable(const gl_logdeg *s);
int gl_small_launch(gl_logdeg *s, int64_t iters, double tol);
size_t gl_small_batch_scratch_bytes(int B, int n);
int gl_small_batch_run(gl_logdeg **sessions, int B, int64_t iters, double tol, void *dev_scratch, int64_t *ran_out,
                       bool *need_exact_out, bool *stopped_out);
// gl_logdeg.hip: layout conversions of an edge vector (tiled <-> condensed, device to device)
int gl_logdeg_tiled_to_condensed(gl_logdeg *s, const double *src_tiled, double *dst_condensed);
int gl_logdeg_condensed_to_tiled(gl_logdeg *s, const double *src_condensed, double *dst_tiled);

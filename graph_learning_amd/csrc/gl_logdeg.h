// Internal: the log-degree solver session and the kernels' parameter block.
#pragma once
#include "gl_common.h"

// Device-side view of one problem (passed by value to the kernels).
struct LogdegDev {
    double *w[2];        // edge iterate, tiled layout, ping-pong: iterate `it` lives in w[it & 1]
    const double *z;     // scaled squared distances, tiled layout
    double *dvar;        // dual variable d            (n_pad)
    double *ybar;        // forward dual point  ybar   (n_pad)
    double *pbar;        // prox dual point     pbar   (n_pad)
    double *part_p;      // per-tile degree partials of p      [nb][n_pad]
    double *part_w;      // per-tile degree partials of w_new  [nb][n_pad]
    double *tnorm;       // per-tile {sum (q-y)^2, sum w^2}    [ntiles][2]
    double *bnorm;       // per-node-block norm partials       [2][nb][4]
    double *status;      // {done, final_iter, rel_primal, rel_dual}
    int64_t n, nb, n_pad, ntiles;
    double gamma;        // scaled step                      (smooth_learning.py:135-137)
    double c;            // 1 - gamma*2*beta
    double g2;           // 2*gamma
    double a4g;          // 4*alpha*gamma
    double tol;
};

struct gl_logdeg {
    gl_ctx *ctx = nullptr;
    LogdegDev dev{};
    int64_t n = 0, n_edges = 0;
    double *zsq = nullptr;        // n x n squared distances (set_signals only)
    double *stage = nullptr;      // device staging for condensed host transfers (E doubles)
    double *xdev = nullptr;       // signals (d x n) on device
    int64_t xdev_elems = 0;
    double *theta_terms = nullptr;  // 2 x n per-row terms of get_theta
    bool have_z = false, inited = false;
    bool have_zsq = false;
    int64_t it = 0;               // iterations completed (iterate lives in w[it & 1])
    double theta = 1.0;
};

// gl_gram.hip
int gl_gram_distances(gl_ctx *ctx, const double *x_dev, int64_t d, int64_t n, double *zsq_dev);
// gl_theta.hip
int gl_theta_from_square(gl_ctx *ctx, const double *zsq_dev, int64_t n, int k, double *terms_dev,
                         double *theta_out);

// Internal: the graph-dictionary session and the kernels' parameter block (gl_graphdict.hip,
// gl_graphdict_small.hip).
#pragma once
#include "gl_common.h"

#include <vector>

constexpr int GD_MAXK = 16;
constexpr int GD_SPL = 4;        // samples per lane in k_gd_grad_c
constexpr int GD_SB = 64 * GD_SPL;  // samples per wave
inline int gd_rows_c(int KP) { return KP <= 8 ? 2 : 1; }  // rows per strip in k_gd_grad_c (SGPR budget)
constexpr int GD_NORM_SLOTS = 8;

struct GdDev {
    int n, n_pad, T, Tp, Tw, W, K, KP;
    int NG;     // strip groups of k_gd_grad_c (slabs of SC)
    int TS;     // sample splits of k_gd_grad_w (slabs of GS)
    int MS;     // sample splits of k_gd_m
    double *X;        // Tp x n_pad
    double *XT;       // n_pad x Tp
    double *WS;       // n_pad x n_pad x KP
    double *C;        // K x T: coefficients_, reference layout (repeated over windows)
    double *CT;       // Tp x KP: the same, sample-major, zero padded
    float *WSf, *CTf; // fp32 mirrors of WS / CT while the contractions run in fp32 (else NULL): every writer of WS / CT keeps them
    double *Cw;       // KP x Tw: new window coefficients of the running step
    double *dual;     // Tw x n
    double *deg;      // KP x n_pad: S W' of the current weights
    double *deg_new;  // KP x n_pad: of the updated weights
    double *SC;       // NG x KP x Tp: partial coefficient gradients
    double *SCR;      // KP x Tp: their sum over the strip groups
    double *cnorm;    // 256 x 3: per-workgroup partials of the coefficient change
    double *GS;       // TS x n_pad x (n_pad / 64) x KP/2 x 64 x 2: partial weight gradients, a column block's atom pair contiguous
    double *MSL;      // MS x KP x n_pad: partial sums of C @ dual
    double *M;        // KP x n_pad
    double *cpart;    // Tw: max |c_new| per window
    double *wpart;    // (n * nbj) x 3: sum (w_new - w)^2, sum w^2, max |w| per (row, column block)
    int2 *wtasks;     // (row strip, column block) tasks of k_gd_grad_w
    int nwtasks;
    double *stat;     // [0] rel change C, [1] rel change W, [2] flag "all coefficients dropped to 0", [3] ||dC||,
                      // [4] steps done, [5] done (later kernels fall through), [6] tol, [7] converged_
    double *hist;     // history of the running fit: 2 doubles per step
    long long hist_cap;
    // parameters of the step
    double l1_w, ortho_w, l1_c, log_c, l1_diff_c, step_c, step_w, step_dual;
};

struct gl_graphdict {
    gl_ctx *ctx = nullptr;
    GdDev dev{};
    double *stage = nullptr;  // device staging for host <-> device transfers of W (K x E) / x (T x n)
    size_t stage_elems = 0;
    bool have_x = false, have_state = false;
    // fp32 form of the two gradient contractions (gl_graphdict_set_precision): fp32 copies of their operands
    bool f32 = false;
    int rc32 = 2;            // rows per strip of the fp32 k_gd_grad_c (gl_graphdict_create)
    int ng64 = 1, ng32 = 1;  // strip groups of k_gd_grad_c in either precision (dev.NG is the current one)
    float *Xf = nullptr, *XTf = nullptr, *WSf = nullptr, *CTf = nullptr;
    // one step (memset + 14 kernels + copy of the status words) captured per parity of the degree
    // ping-pong and replayed: one graph launch per step instead of 16 enqueues (matters below ~1 ms/step)
    hipGraphExec_t step_graph[4] = {nullptr, nullptr, nullptr, nullptr};  // [2 * fit_mode + parity]
    int step_parity = 0;  // which of (deg, deg_new) is current: flips with every step
    double *dbgbuf = nullptr;  // tuning aid (GD_EXP & 8): 4 doubles per wave of k_gd_grad_w
    // side branch of a step: the folds that only need the coefficient gradient's partials and the dual variable run beside
    // the weight gradient (GRAPHLEARN_AMD_GD_FORK=1; measured without gain, off by default)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // small engine (gl_graphdict_small.hip): result words + history of a run, on the device
    double *small_hist = nullptr;
    int64_t small_hist_steps = 0;
    bool small_attr_set = false;
    std::vector<void *> allocs;
};

// gl_graphdict_small.hip: the whole loop of small problems in one launch of one workgroup
bool gd_small_applicable(const gl_graphdict *s);
int gd_small_fit(gl_graphdict *s, int64_t max_iter, double tol, int reduce_step_on_plateau, double *history_out,
                 int64_t *steps_run_out, int64_t *converged_out, int *flags_out);
int gd_small_predict(gl_graphdict *s, int64_t max_iter, double tol, int64_t *steps_run_out, int *flags_out);

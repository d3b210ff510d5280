// Graph-dictionary model with log-barrier degrees (GraphDictLog), the "next" row f-1 of SURVEY.md
// section 8.  Reference: graph_learn/dictionary/base_model.py:355-527 (_update_coefficients,
// _update_weights, _fit_step, _single_fit), dictionary/log_model.py:27-50 (adjoints, dual prox).
//
// T samples on n nodes, K atoms.  The reference materialises the (T x E) matrix of squared
// pairwise differences D[t,e] = (x_ti - x_tj)^2 (17 GB at n = 1024, T = 4096) and runs two GEMMs
// per step against it:  W @ D' (K x T, coefficient gradient) and C @ D (K x E, weight gradient).
// With K <= 16 these are not GEMM-shaped work: forming D[t,e] on the fly costs 2 flops against
// 2 K flops of use, and nothing has to be stored.  Two kernels do that:
//
//   k_gd_grad_c   lane <-> sample (two per lane), wave walks pairs (i, j): x_i, x_j come from the
//                 transposed signal matrix XT (n x T, coalesced), the K weights of the pair are
//                 wave-uniform -> scalar loads from the pair-major weight array WS[i][j][K].
//   k_gd_grad_w   lane <-> column j, wave holds a strip of rows i: x_j from X (T x n, coalesced),
//                 x_i and the K coefficients of sample t are wave-uniform -> scalar loads.
//
// Both are bound by the fp64 vector rate (2 + 2K lane-ops per pair-sample and kernel), not by
// HBM: X / XT (33 MB at the BASELINE size) stay in L2 / Infinity Cache.  Everything else of a
// step (adjoint terms through the closed form of S / S', projections, dual prox, relative
// changes) is O(K E + K T + T n) and runs in a handful of small kernels.
//
// Weights live in WS: n_pad x n_pad x KP doubles, entry (i, j, k) for i < j < n, zero elsewhere
// (the zero lower triangle doubles as the mask of the pair loops).  KP = K rounded up to 4/8/16.
#include "gl_common.h"

#include <algorithm>
#include <limits>
#include <new>
#include <vector>

namespace {

constexpr int GD_MAXK = 16;
constexpr int GD_SB = 128;       // samples per wave in k_gd_grad_c (two per lane)
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
    double *Cw;       // KP x Tw: new window coefficients of the running step
    double *dual;     // Tw x n
    double *deg;      // KP x n_pad: S W' of the current weights
    double *deg_new;  // KP x n_pad: of the updated weights
    double *SC;       // NG x KP x Tp: partial coefficient gradients
    double *GS;       // TS x n_pad x n_pad x KP: partial weight gradients
    double *MSL;      // MS x KP x n_pad: partial sums of C @ dual
    double *M;        // KP x n_pad
    double *cpart;    // Tw: max |c_new| per window
    double *wpart;    // (n * nbj) x 3: sum (w_new - w)^2, sum w^2, max |w| per (row, column block)
    double *stat;     // [0] rel change C, [1] rel change W, [2] flag "all coefficients dropped to 0"
    // parameters of the step
    double l1_w, ortho_w, l1_c, log_c, l1_diff_c, step_c, step_w, step_dual;
};

__device__ __forceinline__ double gd_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double gd_wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// ------------------------------------------------------------------------------------------
// coefficient gradient: SC[g][k][t] = sum over the pairs of strip group g of W[k,(i,j)] (x_ti - x_tj)^2
// (base_model.py:375 `self.weights_ @ sq_pdiffs.T`).  grid (Tp / 128, NG), one wave per workgroup.
template <int KP>
__global__ __launch_bounds__(64) void k_gd_grad_c(const double *__restrict__ XT, const double *__restrict__ WS,
                                                  double *__restrict__ SC, int n, int n_pad, int Tp, int NG) {
    const int lane = threadIdx.x;
    const int g = blockIdx.y;
    const int64_t t0 = (int64_t)blockIdx.x * GD_SB + lane;
    double acc0[KP], acc1[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) acc0[k] = acc1[k] = 0.0;
    const int nstrips = (n + 3) / 4;
    for (int s = g; s < nstrips; s += NG) {
        const int i0 = 4 * s;
        double xi0[4], xi1[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            xi0[a] = XT[(int64_t)(i0 + a) * Tp + t0];
            xi1[a] = XT[(int64_t)(i0 + a) * Tp + t0 + 64];
        }
        for (int j = i0 + 1; j < n; ++j) {
            const double xj0 = XT[(int64_t)j * Tp + t0], xj1 = XT[(int64_t)j * Tp + t0 + 64];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                // rows i0 + a >= j carry zeros in WS (lower triangle): no branch needed
                const double *__restrict__ wp = WS + ((int64_t)(i0 + a) * n_pad + j) * KP;  // wave-uniform
                const double d0 = xi0[a] - xj0, d1 = xi1[a] - xj1;
                const double s0 = d0 * d0, s1 = d1 * d1;
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    const double w = wp[k];
                    acc0[k] = __builtin_fma(w, s0, acc0[k]);
                    acc1[k] = __builtin_fma(w, s1, acc1[k]);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        SC[((int64_t)g * KP + k) * Tp + t0] = acc0[k];
        SC[((int64_t)g * KP + k) * Tp + t0 + 64] = acc1[k];
    }
}

// ------------------------------------------------------------------------------------------
// weight gradient: GS[ts][i][j][k] = sum over the samples of split ts of C[k,t] (x_ti - x_tj)^2
// (base_model.py:447 `self.coefficients_ @ sq_pdiffs`).  grid (n_pad / 64, n_pad / SW, TS).
template <int KP, int SW>
__global__ __launch_bounds__(64) void k_gd_grad_w(const double *__restrict__ X, const double *__restrict__ CT,
                                                  double *__restrict__ GS, int n_pad, int Tp, int tchunk) {
    const int lane = threadIdx.x;
    const int jb = blockIdx.x, i0 = blockIdx.y * SW, ts = blockIdx.z;
    if (jb * 64 + 63 <= i0) return;  // no pair with i < j in this block
    const int j = jb * 64 + lane;
    double acc[SW][KP];
#pragma unroll
    for (int a = 0; a < SW; ++a)
#pragma unroll
        for (int k = 0; k < KP; ++k) acc[a][k] = 0.0;
    const int tA = ts * tchunk, tB = min(Tp, tA + tchunk);
#pragma unroll 2
    for (int t = tA; t < tB; ++t) {
        const double xj = X[(int64_t)t * n_pad + j];
        const double *__restrict__ xr = X + (int64_t)t * n_pad + i0;  // wave-uniform
        const double *__restrict__ cr = CT + (int64_t)t * KP;         // wave-uniform
#pragma unroll
        for (int a = 0; a < SW; ++a) {
            const double d = xr[a] - xj;
            const double sq = d * d;
#pragma unroll
            for (int k = 0; k < KP; ++k) acc[a][k] = __builtin_fma(cr[k], sq, acc[a][k]);
        }
    }
#pragma unroll
    for (int a = 0; a < SW; ++a) {
        double *dst = GS + (((int64_t)ts * n_pad + (i0 + a)) * n_pad + j) * KP;
#pragma unroll
        for (int k = 0; k < KP; k += 2) *reinterpret_cast<double2 *>(dst + k) = double2{acc[a][k], acc[a][k + 1]};
    }
}

// ------------------------------------------------------------------------------------------
// log_c reset (base_model.py:385-387): columns of the strided view whose sum is < 1e-8
__global__ void k_gd_logc_reset(GdDev P) {
    const int tw = blockIdx.x * blockDim.x + threadIdx.x;
    if (tw >= P.Tw) return;
    const int64_t t = (int64_t)tw * P.W;
    double s = 0;
    for (int k = 0; k < P.K; ++k) s += P.C[(int64_t)k * P.T + t];
    if (s < 1e-8)
        for (int k = 0; k < P.K; ++k) {
            P.C[(int64_t)k * P.T + t] = P.log_c;
            P.CT[t * P.KP + k] = P.log_c;
        }
}

// ------------------------------------------------------------------------------------------
// coefficient update (base_model.py:373-401), one wave per window
__global__ __launch_bounds__(64) void k_gd_coef_update(GdDev P, int use_sc) {
    const int lane = threadIdx.x, tw = blockIdx.x;
    const int K = P.K;
    // adjoint term  W @ (S' dual')  ==  <S W_k, dual_tw>   (log_model.py:33-34)
    double dt[GD_MAXK];
    for (int k = 0; k < K; ++k) dt[k] = 0.0;
    for (int i = lane; i < P.n; i += 64) {
        const double dv = P.dual[(int64_t)tw * P.n + i];
        for (int k = 0; k < K; ++k) dt[k] += P.deg[(int64_t)k * P.n_pad + i] * dv;
    }
    for (int k = 0; k < K; ++k) dt[k] = gd_wave_sum(dt[k]);
    // smoothness term: partial sums over the strip groups, mean over the window's samples (:375-379)
    double sm[GD_MAXK];
    for (int k = 0; k < K; ++k) {
        double tot = 0.0;
        for (int q = 0; q < P.W; ++q) {
            const int64_t t = (int64_t)tw * P.W + q;
            double v = 0.0;
            for (int g = lane; g < P.NG; g += 64) v += P.SC[((int64_t)g * P.KP + k) * P.Tp + t];
            tot += gd_wave_sum(v);
        }
        sm[k] = P.W > 1 ? tot / (double)P.W : tot;
    }
    (void)use_sc;
    double cmax = 0.0;
    if (lane < K) {
        const int k = lane;
        const double *c = P.C + (int64_t)k * P.T;  // the window view is c[m * W]
        double grad = 0.0, dterm = 0.0;
        for (int q = 0; q < K; ++q)  // select without dynamic register indexing
            if (q == k) {
                grad = sm[q];
                dterm = dt[q];
            }
        grad += P.l1_c * (double)P.T;
        if (P.l1_diff_c > 0) {  // :389-394, diff(sign(diff(c)), prepend=0, append=0)[tw]
            auto sgn = [](double v) { return v > 0 ? 1.0 : (v < 0 ? -1.0 : 0.0); };
            const double right = tw + 1 < P.Tw ? sgn(c[(int64_t)(tw + 1) * P.W] - c[(int64_t)tw * P.W]) : 0.0;
            const double left = tw > 0 ? sgn(c[(int64_t)tw * P.W] - c[(int64_t)(tw - 1) * P.W]) : 0.0;
            grad -= P.l1_diff_c * (double)P.T * (right - left);
        }
        double cn = c[(int64_t)tw * P.W] - P.step_c / (double)P.T * (dterm + grad);  // :400
        cn = cn < 0 ? 0.0 : cn;
        cn = cn > 1 ? 1.0 : cn;
        P.Cw[(int64_t)k * P.Tw + tw] = cn;
        cmax = fabs(cn);
    }
    cmax = gd_wave_max(cmax);
    if (lane == 0) P.cpart[tw] = cmax;
}

// all-zero guard (:406-408) and the relative change of the coefficients (:516); one workgroup
__global__ __launch_bounds__(256) void k_gd_coef_finish(GdDev P) {
    __shared__ double s_a[256], s_b[256], s_c[256];
    const int tid = threadIdx.x;
    double mx = 0.0;
    for (int tw = tid; tw < P.Tw; tw += 256) mx = fmax(mx, P.cpart[tw]);
    s_a[tid] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) s_a[tid] = fmax(s_a[tid], s_a[tid + o]);
        __syncthreads();
    }
    const bool all_zero = s_a[0] <= 1e-8;  // np.allclose(c, 0)
    __syncthreads();
    if (all_zero) {
        for (int64_t q = tid; q < (int64_t)P.K * P.Tw; q += 256) P.Cw[(q / P.Tw) * P.Tw + q % P.Tw] = P.log_c;
        if (tid == 0) P.stat[2] = 1.0;
    }
    __syncthreads();
    // relative_error(coefficients_.ravel(), new.ravel()): ||new - old|| / ||old||, plain norm if old ~ 0
    double num = 0, den = 0, omax = 0;
    for (int64_t q = tid; q < (int64_t)P.K * P.T; q += 256) {
        const int64_t k = q / P.T, t = q % P.T;
        const double o = P.C[q], nw = P.Cw[k * P.Tw + t / P.W];
        num += (nw - o) * (nw - o);
        den += o * o;
        omax = fmax(omax, fabs(o));
    }
    s_a[tid] = num;
    s_b[tid] = den;
    s_c[tid] = omax;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            s_a[tid] += s_a[tid + o];
            s_b[tid] += s_b[tid + o];
            s_c[tid] = fmax(s_c[tid], s_c[tid + o]);
        }
        __syncthreads();
    }
    if (tid == 0) P.stat[0] = s_c[0] <= 1e-8 ? sqrt(s_a[0]) : sqrt(s_a[0]) / sqrt(s_b[0]);
}

// ------------------------------------------------------------------------------------------
// M[k][i] = sum_t C[k,t] dual[t / W, i]  (log_model.py:36-37: C @ (S' dual_rep')' = S'(C @ dual_rep) pairwise)
__global__ __launch_bounds__(64) void k_gd_m(GdDev P, int tchunk) {
    const int i = blockIdx.x * 64 + threadIdx.x, ms = blockIdx.y;
    double acc[GD_MAXK];
    for (int k = 0; k < P.K; ++k) acc[k] = 0.0;
    const int tA = ms * tchunk, tB = min(P.T, tA + tchunk);
    if (i < P.n)
        for (int t = tA; t < tB; ++t) {
            const double dv = P.dual[(int64_t)(t / P.W) * P.n + i];
            for (int k = 0; k < P.K; ++k) acc[k] += P.CT[(int64_t)t * P.KP + k] * dv;
        }
    for (int k = 0; k < P.K; ++k) P.MSL[((int64_t)ms * P.KP + k) * P.n_pad + i] = i < P.n ? acc[k] : 0.0;
}
__global__ void k_gd_m_reduce(GdDev P) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (int64_t)P.KP * P.n_pad) return;
    double v = 0.0;
    for (int ms = 0; ms < P.MS; ++ms) v += P.MSL[(int64_t)ms * P.KP * P.n_pad + q];
    P.M[q] = v;
}

// ------------------------------------------------------------------------------------------
// weight update (base_model.py:444-461): one wave per (row i, column block); in place
__global__ __launch_bounds__(64) void k_gd_weight_update(GdDev P) {
    const int lane = threadIdx.x, jb = blockIdx.x, i = blockIdx.y;
    const int nbj = gridDim.x;
    const int j = jb * 64 + lane;
    double num = 0, den = 0, wmax = 0;
    if (jb * 64 + 63 > i && j > i && j < P.n) {
        double *w = P.WS + ((int64_t)i * P.n_pad + j) * P.KP;
        double wsum = 0.0;
        for (int k = 0; k < P.K; ++k) wsum += w[k];
        const double Tn = (double)P.T;
        for (int k = 0; k < P.K; ++k) {
            double g = 0.0;
            for (int ts = 0; ts < P.TS; ++ts) g += P.GS[(((int64_t)ts * P.n_pad + i) * P.n_pad + j) * P.KP + k];
            g += P.l1_w * Tn;                                                   // :448
            const double wo = w[k];
            if (P.ortho_w > 0) g += P.ortho_w * Tn * (wsum - wo);               // :450-451
            const double dterm = P.M[(int64_t)k * P.n_pad + i] + P.M[(int64_t)k * P.n_pad + j];
            double wn = wo - P.step_w / Tn * (dterm + g);                       // :456
            wn = wn < 0 ? 0.0 : wn;                                             // :459
            w[k] = wn;
            num += (wn - wo) * (wn - wo);
            den += wo * wo;
            wmax = fmax(wmax, fabs(wo));
        }
    }
    num = gd_wave_sum(num);
    den = gd_wave_sum(den);
    wmax = gd_wave_max(wmax);
    if (lane == 0) {
        double *o = P.wpart + ((int64_t)i * nbj + jb) * 3;
        o[0] = num;
        o[1] = den;
        o[2] = wmax;
    }
}

__global__ __launch_bounds__(256) void k_gd_weight_finish(GdDev P, int64_t nparts) {
    __shared__ double s_a[256], s_b[256], s_c[256];
    const int tid = threadIdx.x;
    double num = 0, den = 0, mx = 0;
    for (int64_t q = tid; q < nparts; q += 256) {
        num += P.wpart[3 * q];
        den += P.wpart[3 * q + 1];
        mx = fmax(mx, P.wpart[3 * q + 2]);
    }
    s_a[tid] = num;
    s_b[tid] = den;
    s_c[tid] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            s_a[tid] += s_a[tid + o];
            s_b[tid] += s_b[tid + o];
            s_c[tid] = fmax(s_c[tid], s_c[tid + o]);
        }
        __syncthreads();
    }
    if (tid == 0) P.stat[1] = s_c[0] <= 1e-8 ? sqrt(s_a[0]) : sqrt(s_a[0]) / sqrt(s_b[0]);
}

// degrees S W_k' of the weights in WS: one wave per node
__global__ __launch_bounds__(64) void k_gd_degree(GdDev P, double *__restrict__ out) {
    const int lane = threadIdx.x, v = blockIdx.x;
    double acc[GD_MAXK];
    for (int k = 0; k < P.K; ++k) acc[k] = 0.0;
    if (v < P.n)
        for (int u = lane; u < P.n; u += 64) {
            if (u == v) continue;
            const double *p = u < v ? P.WS + ((int64_t)u * P.n_pad + v) * P.KP : P.WS + ((int64_t)v * P.n_pad + u) * P.KP;
            for (int k = 0; k < P.K; ++k) acc[k] += p[k];
        }
    for (int k = 0; k < P.K; ++k) {
        const double s = gd_wave_sum(acc[k]);
        if (lane == 0) out[(int64_t)k * P.n_pad + v] = s;
    }
}

// ------------------------------------------------------------------------------------------
// dual update (log_model.py:39-50) with the overshoot 2 new - old of weights and coefficients
// (base_model.py:509-514).  fixed_weights: predict() passes the weights themselves (:665-669).
__global__ void k_gd_dual_update(GdDev P, int fixed_weights) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, tw = blockIdx.y;
    if (i >= P.n) return;
    double step = 0.0;
    for (int k = 0; k < P.K; ++k) {
        const double dn = P.deg_new[(int64_t)k * P.n_pad + i], dold = P.deg[(int64_t)k * P.n_pad + i];
        const double dv = fixed_weights ? dold : 2.0 * dn - dold;
        const double cv = 2.0 * P.Cw[(int64_t)k * P.Tw + tw] - P.C[(int64_t)k * P.T + (int64_t)tw * P.W];
        step += dv * cv;
    }
    const double sigma = P.step_dual;
    double d = P.dual[(int64_t)tw * P.n + i] + sigma * step;
    P.dual[(int64_t)tw * P.n + i] = (d - sqrt(d * d + 4.0 * sigma)) / 2.0;
}

// coefficients_ <- repeat(new window coefficients) (base_model.py:495-501,519); predict's stopping
// norm ||new - old|| (:671) goes to stat[3]
__global__ __launch_bounds__(256) void k_gd_coef_commit(GdDev P) {
    __shared__ double s_a[256];
    const int tid = threadIdx.x;
    double num = 0.0;
    for (int64_t q = (int64_t)blockIdx.x * 256 + tid; q < (int64_t)P.K * P.T; q += (int64_t)gridDim.x * 256) {
        const int64_t k = q / P.T, t = q % P.T;
        const double nw = P.Cw[k * P.Tw + t / P.W], o = P.C[q];
        num += (nw - o) * (nw - o);
        P.C[q] = nw;
        P.CT[t * P.KP + k] = nw;
    }
    s_a[tid] = num;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) s_a[tid] += s_a[tid + o];
        __syncthreads();
    }
    if (tid == 0) P.stat[3] = sqrt(s_a[0]);  // single workgroup launch
}

// ---- layout kernels --------------------------------------------------------------------------
__global__ void k_gd_pack_x(const double *__restrict__ x, GdDev P) {  // x: T x n row-major
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (int64_t)P.T * P.n) return;
    const int64_t t = q / P.n, i = q % P.n;
    const double v = x[q];
    P.X[t * P.n_pad + i] = v;
    P.XT[i * P.Tp + t] = v;
}
__global__ void k_gd_pack_w(const double *__restrict__ w, GdDev P) {  // w: K x E condensed
    const int64_t E = (int64_t)P.n * (P.n - 1) / 2;
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (int64_t)P.n * P.n) return;
    const int64_t i = q / P.n, j = q % P.n;
    if (j <= i) return;
    const int64_t e = gl_condensed_index(i, j, P.n);
    for (int k = 0; k < P.K; ++k) P.WS[(i * P.n_pad + j) * P.KP + k] = w[(int64_t)k * E + e];
}
__global__ void k_gd_unpack_w(double *__restrict__ w, GdDev P) {
    const int64_t E = (int64_t)P.n * (P.n - 1) / 2;
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (int64_t)P.n * P.n) return;
    const int64_t i = q / P.n, j = q % P.n;
    if (j <= i) return;
    const int64_t e = gl_condensed_index(i, j, P.n);
    for (int k = 0; k < P.K; ++k) w[(int64_t)k * E + e] = P.WS[(i * P.n_pad + j) * P.KP + k];
}
__global__ void k_gd_pack_c(GdDev P) {  // CT from C
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (int64_t)P.K * P.T) return;
    const int64_t k = q / P.T, t = q % P.T;
    P.CT[t * P.KP + k] = P.C[q];
}

}  // namespace

struct gl_graphdict {
    gl_ctx *ctx = nullptr;
    GdDev dev{};
    double *stage = nullptr;  // device staging for host <-> device transfers of W (K x E) / x (T x n)
    size_t stage_elems = 0;
    bool have_x = false, have_state = false;
    std::vector<void *> allocs;
};

namespace {

int gd_check(gl_graphdict *s) {
    if (!s || !s->ctx) return GL_E_INVALID;
    return GL_OK;
}

template <typename T>
int gd_alloc(gl_graphdict *s, T **p, size_t count) {
    gl_ctx *ctx = s->ctx;
    GL_HIP(ctx, hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(T)));
    s->allocs.push_back(*p);
    GL_HIP(ctx, hipMemsetAsync(*p, 0, std::max<size_t>(count, 1) * sizeof(T), ctx->stream));
    return GL_OK;
}

int gd_launch_grad_c(gl_graphdict *s) {
    const GdDev &P = s->dev;
    dim3 grid((unsigned)(P.Tp / GD_SB), (unsigned)P.NG);
    hipStream_t st = s->ctx->stream;
    switch (P.KP) {
        case 4: k_gd_grad_c<4><<<grid, 64, 0, st>>>(P.XT, P.WS, P.SC, P.n, P.n_pad, P.Tp, P.NG); break;
        case 8: k_gd_grad_c<8><<<grid, 64, 0, st>>>(P.XT, P.WS, P.SC, P.n, P.n_pad, P.Tp, P.NG); break;
        default: k_gd_grad_c<16><<<grid, 64, 0, st>>>(P.XT, P.WS, P.SC, P.n, P.n_pad, P.Tp, P.NG); break;
    }
    GL_HIP(s->ctx, hipGetLastError());
    return GL_OK;
}

int gd_strip_w(int KP) { return KP <= 8 ? 8 : 4; }

int gd_launch_grad_w(gl_graphdict *s) {
    const GdDev &P = s->dev;
    const int tchunk = (P.Tp + P.TS - 1) / P.TS;
    hipStream_t st = s->ctx->stream;
    dim3 grid((unsigned)(P.n_pad / 64), (unsigned)(P.n_pad / gd_strip_w(P.KP)), (unsigned)P.TS);
    switch (P.KP) {
        case 4: k_gd_grad_w<4, 8><<<grid, 64, 0, st>>>(P.X, P.CT, P.GS, P.n_pad, P.Tp, tchunk); break;
        case 8: k_gd_grad_w<8, 8><<<grid, 64, 0, st>>>(P.X, P.CT, P.GS, P.n_pad, P.Tp, tchunk); break;
        default: k_gd_grad_w<16, 4><<<grid, 64, 0, st>>>(P.X, P.CT, P.GS, P.n_pad, P.Tp, tchunk); break;
    }
    GL_HIP(s->ctx, hipGetLastError());
    return GL_OK;
}

int gd_degrees(gl_graphdict *s, double *out) {
    k_gd_degree<<<dim3((unsigned)s->dev.n), 64, 0, s->ctx->stream>>>(s->dev, out);
    GL_HIP(s->ctx, hipGetLastError());
    return GL_OK;
}

// one primal-dual step (base_model.py:485-527); the relative changes land in stat[0..1]
int gd_step(gl_graphdict *s) {
    gl_ctx *ctx = s->ctx;
    GdDev &P = s->dev;
    hipStream_t st = ctx->stream;
    if (P.log_c > 0) k_gd_logc_reset<<<dim3((unsigned)((P.Tw + 255) / 256)), 256, 0, st>>>(P);
    GL_TRY(gd_launch_grad_c(s));
    GL_TRY(gd_launch_grad_w(s));
    k_gd_coef_update<<<dim3((unsigned)P.Tw), 64, 0, st>>>(P, 1);
    k_gd_coef_finish<<<1, 256, 0, st>>>(P);
    const int mchunk = (P.T + P.MS - 1) / P.MS;
    k_gd_m<<<dim3((unsigned)(P.n_pad / 64), (unsigned)P.MS), 64, 0, st>>>(P, mchunk);
    k_gd_m_reduce<<<dim3((unsigned)(((int64_t)P.KP * P.n_pad + 255) / 256)), 256, 0, st>>>(P);
    const int nbj = P.n_pad / 64;
    k_gd_weight_update<<<dim3((unsigned)nbj, (unsigned)P.n), 64, 0, st>>>(P);
    k_gd_weight_finish<<<1, 256, 0, st>>>(P, (int64_t)P.n * nbj);
    GL_TRY(gd_degrees(s, P.deg_new));
    k_gd_dual_update<<<dim3((unsigned)((P.n + 255) / 256), (unsigned)P.Tw), 256, 0, st>>>(P, 0);
    k_gd_coef_commit<<<1, 256, 0, st>>>(P);
    GL_HIP(ctx, hipGetLastError());
    std::swap(P.deg, P.deg_new);
    return GL_OK;
}

}  // namespace

// ==============================================================================================
GL_API int gl_graphdict_create(gl_ctx *ctx, int64_t n_nodes, int64_t n_samples, int n_atoms, int window_size,
                               gl_graphdict **out) {
    if (!ctx) return GL_E_INVALID;
    if (!out) return gl_set_error(ctx, GL_E_INVALID, "out is NULL");
    *out = nullptr;
    if (n_nodes < 2 || n_nodes > 16384) return gl_set_error(ctx, GL_E_INVALID, "n_nodes must be in [2, 16384]");
    if (n_atoms < 1 || n_atoms > GD_MAXK)
        return gl_set_error(ctx, GL_E_INVALID, "n_atoms must be in [1, %d], got %d", GD_MAXK, n_atoms);
    if (window_size < 1 || n_samples < 1 || n_samples % window_size != 0)
        return gl_set_error(ctx, GL_E_INVALID, "n_samples must be a positive multiple of window_size");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    gl_graphdict *s = new (std::nothrow) gl_graphdict();
    if (!s) return GL_E_NOMEM;
    s->ctx = ctx;
    GdDev &P = s->dev;
    P.n = (int)n_nodes;
    P.n_pad = (int)((n_nodes + 63) / 64 * 64);
    P.T = (int)n_samples;
    P.Tp = (int)((n_samples + GD_SB - 1) / GD_SB * GD_SB);
    P.W = window_size;
    P.Tw = P.T / P.W;
    P.K = n_atoms;
    P.KP = n_atoms <= 4 ? 4 : (n_atoms <= 8 ? 8 : 16);
    // enough waves to fill 256 CUs x 4 SIMDs a few times over, slabs kept small
    const int nstrips = (P.n + 3) / 4;
    const int sblocks = P.Tp / GD_SB;
    P.NG = std::max(1, std::min(nstrips, std::min(128, (4096 + sblocks - 1) / sblocks)));
    const int64_t wtasks = (int64_t)(P.n_pad / 64) * (P.n_pad / gd_strip_w(P.KP)) / 2 + 1;
    P.TS = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, (P.Tp + 255) / 256), (4096 + wtasks - 1) / wtasks));
    P.MS = std::max(1, std::min(32, P.T / 64));
    int rc = GL_OK;
    const size_t np2 = (size_t)P.n_pad * P.n_pad;
    if ((rc = gd_alloc(s, &P.X, (size_t)P.Tp * P.n_pad)) || (rc = gd_alloc(s, &P.XT, (size_t)P.n_pad * P.Tp)) ||
        (rc = gd_alloc(s, &P.WS, np2 * P.KP)) || (rc = gd_alloc(s, &P.C, (size_t)P.K * P.T)) ||
        (rc = gd_alloc(s, &P.CT, (size_t)P.Tp * P.KP)) || (rc = gd_alloc(s, &P.Cw, (size_t)P.KP * P.Tw)) ||
        (rc = gd_alloc(s, &P.dual, (size_t)P.Tw * P.n)) || (rc = gd_alloc(s, &P.deg, (size_t)P.KP * P.n_pad)) ||
        (rc = gd_alloc(s, &P.deg_new, (size_t)P.KP * P.n_pad)) || (rc = gd_alloc(s, &P.SC, (size_t)P.NG * P.KP * P.Tp)) ||
        (rc = gd_alloc(s, &P.GS, (size_t)P.TS * np2 * P.KP)) || (rc = gd_alloc(s, &P.MSL, (size_t)P.MS * P.KP * P.n_pad)) ||
        (rc = gd_alloc(s, &P.M, (size_t)P.KP * P.n_pad)) || (rc = gd_alloc(s, &P.cpart, (size_t)P.Tw)) ||
        (rc = gd_alloc(s, &P.wpart, (size_t)P.n * (P.n_pad / 64) * 3)) || (rc = gd_alloc(s, &P.stat, GD_NORM_SLOTS))) {
        for (void *p : s->allocs) (void)hipFree(p);
        delete s;
        return rc;
    }
    s->stage_elems = std::max((size_t)P.K * ((size_t)P.n * (P.n - 1) / 2), (size_t)P.T * P.n);
    if ((rc = gd_alloc(s, &s->stage, s->stage_elems))) {
        for (void *p : s->allocs) (void)hipFree(p);
        delete s;
        return rc;
    }
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = s;
    return GL_OK;
}

GL_API void gl_graphdict_destroy(gl_graphdict *s) {
    if (!s) return;
    if (s->ctx) {
        (void)hipSetDevice(s->ctx->device);
        (void)hipStreamSynchronize(s->ctx->stream);
    }
    for (void *p : s->allocs) (void)hipFree(p);
    delete s;
}

GL_API int gl_graphdict_set_signals(gl_graphdict *s, const double *x) {
    GL_TRY(gd_check(s));
    gl_ctx *ctx = s->ctx;
    if (!x) return gl_set_error(ctx, GL_E_INVALID, "x is NULL");
    GdDev &P = s->dev;
    GL_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cnt = (size_t)P.T * P.n;
    for (size_t q = 0; q < cnt; ++q)
        if (!std::isfinite(x[q])) return gl_set_error(ctx, GL_E_NONFINITE, "x contains NaN or Inf");
    GL_HIP(ctx, hipMemcpyAsync(s->stage, x, cnt * 8, hipMemcpyHostToDevice, ctx->stream));
    k_gd_pack_x<<<dim3((unsigned)((cnt + 255) / 256)), 256, 0, ctx->stream>>>(s->stage, P);
    GL_HIP(ctx, hipGetLastError());
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    s->have_x = true;
    return GL_OK;
}

GL_API int gl_graphdict_set_state(gl_graphdict *s, const double *coefficients, const double *weights, const double *dual) {
    GL_TRY(gd_check(s));
    gl_ctx *ctx = s->ctx;
    GdDev &P = s->dev;
    if (!coefficients || !weights) return gl_set_error(ctx, GL_E_INVALID, "coefficients / weights are NULL");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    const size_t E = (size_t)P.n * (P.n - 1) / 2;
    GL_HIP(ctx, hipMemcpyAsync(P.C, coefficients, (size_t)P.K * P.T * 8, hipMemcpyHostToDevice, ctx->stream));
    GL_HIP(ctx, hipMemsetAsync(P.CT, 0, (size_t)P.Tp * P.KP * 8, ctx->stream));
    k_gd_pack_c<<<dim3((unsigned)(((size_t)P.K * P.T + 255) / 256)), 256, 0, ctx->stream>>>(P);
    GL_HIP(ctx, hipMemcpyAsync(s->stage, weights, (size_t)P.K * E * 8, hipMemcpyHostToDevice, ctx->stream));
    GL_HIP(ctx, hipMemsetAsync(P.WS, 0, (size_t)P.n_pad * P.n_pad * P.KP * 8, ctx->stream));
    k_gd_pack_w<<<dim3((unsigned)(((size_t)P.n * P.n + 255) / 256)), 256, 0, ctx->stream>>>(s->stage, P);
    if (dual)
        GL_HIP(ctx, hipMemcpyAsync(P.dual, dual, (size_t)P.Tw * P.n * 8, hipMemcpyHostToDevice, ctx->stream));
    else
        GL_HIP(ctx, hipMemsetAsync(P.dual, 0, (size_t)P.Tw * P.n * 8, ctx->stream));
    GL_TRY(gd_degrees(s, P.deg));
    GL_HIP(ctx, hipGetLastError());
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    s->have_state = true;
    return GL_OK;
}

GL_API int gl_graphdict_get_state(gl_graphdict *s, double *coefficients, double *weights, double *dual) {
    GL_TRY(gd_check(s));
    gl_ctx *ctx = s->ctx;
    GdDev &P = s->dev;
    if (!s->have_state) return gl_set_error(ctx, GL_E_STATE, "no state set");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    const size_t E = (size_t)P.n * (P.n - 1) / 2;
    if (coefficients)
        GL_HIP(ctx, hipMemcpyAsync(coefficients, P.C, (size_t)P.K * P.T * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (weights) {
        k_gd_unpack_w<<<dim3((unsigned)(((size_t)P.n * P.n + 255) / 256)), 256, 0, ctx->stream>>>(s->stage, P);
        GL_HIP(ctx, hipGetLastError());
        GL_HIP(ctx, hipMemcpyAsync(weights, s->stage, (size_t)P.K * E * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (dual) GL_HIP(ctx, hipMemcpyAsync(dual, P.dual, (size_t)P.Tw * P.n * 8, hipMemcpyDeviceToHost, ctx->stream));
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GL_OK;
}

GL_API int gl_graphdict_set_params(gl_graphdict *s, double l1_w, double ortho_w, double l1_c, double log_c,
                                   double l1_diff_c, double step_c, double step_w, double step_dual) {
    GL_TRY(gd_check(s));
    if (!(step_c > 0) || !(step_w > 0) || !(step_dual > 0))
        return gl_set_error(s->ctx, GL_E_INVALID, "step sizes must be positive");
    GdDev &P = s->dev;
    P.l1_w = l1_w, P.ortho_w = ortho_w, P.l1_c = l1_c, P.log_c = log_c, P.l1_diff_c = l1_diff_c;
    P.step_c = step_c, P.step_w = step_w, P.step_dual = step_dual;
    return GL_OK;
}

GL_API int gl_graphdict_get_steps(gl_graphdict *s, double *steps3_out) {
    GL_TRY(gd_check(s));
    if (!steps3_out) return gl_set_error(s->ctx, GL_E_INVALID, "out is NULL");
    steps3_out[0] = s->dev.step_c, steps3_out[1] = s->dev.step_w, steps3_out[2] = s->dev.step_dual;
    return GL_OK;
}

// The loop of _single_fit (base_model.py:552-581).
GL_API int gl_graphdict_fit(gl_graphdict *s, int64_t max_iter, double tol, int reduce_step_on_plateau,
                            double *history_out, int64_t *steps_run_out, int64_t *converged_out, int *flags_out) {
    GL_TRY(gd_check(s));
    gl_ctx *ctx = s->ctx;
    if (!s->have_x || !s->have_state) return gl_set_error(ctx, GL_E_STATE, "fit needs set_signals and set_state first");
    if (max_iter < 0 || !history_out) return gl_set_error(ctx, GL_E_INVALID, "bad argument");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    GdDev &P = s->dev;
    double *st = (double *)ctx->pinned;
    int64_t converged = -1, i = 0;
    int flags = 0;
    for (int64_t q = 0; q < 2 * max_iter; ++q) history_out[q] = -1.0;
    for (i = 0; i < max_iter; ++i) {
        GL_HIP(ctx, hipMemsetAsync(P.stat + 2, 0, 8, ctx->stream));
        GL_TRY(gd_step(s));
        GL_HIP(ctx, hipMemcpyAsync(st, P.stat, 4 * 8, hipMemcpyDeviceToHost, ctx->stream));
        GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const double dc = st[0], dw = st[1];
        history_out[2 * i] = dc;
        history_out[2 * i + 1] = dw;
        if (st[2] != 0.0) flags |= 1;  // "All coefficients dropped to 0"
        if (dc < tol && dw < tol) converged = i;
        if (reduce_step_on_plateau && i > 100) {  // :563-574
            double mc = 0, mw = 0;
            for (int64_t q = i - 10; q < i; ++q) {
                mc += history_out[2 * q];
                mw += history_out[2 * q + 1];
            }
            if (dc >= mc / 10 && dw >= mw / 10) {
                P.step_c *= 0.7, P.step_w *= 0.7, P.step_dual *= 0.7;
                flags |= 2;  // "Divergence detected ..., reducing step size"
            }
        }
        if (converged > 0) {  // sic: a stop at step 0 is not taken (:578)
            ++i;
            break;
        }
    }
    if (steps_run_out) *steps_run_out = i;
    if (converged_out) *converged_out = converged;
    if (flags_out) *flags_out = flags;
    return GL_OK;
}

// predict (base_model.py:647-676): coefficients and dual only, weights fixed.  The session's
// coefficients are the start (K x T), its dual is reset to zero.  Returns in coefficients_out
// what the reference returns (the update that met the tolerance, or the last iterate).
GL_API int gl_graphdict_predict(gl_graphdict *s, int64_t max_iter, double tol, double *coefficients_out,
                                int64_t *steps_run_out, int *flags_out) {
    GL_TRY(gd_check(s));
    gl_ctx *ctx = s->ctx;
    if (!s->have_x || !s->have_state) return gl_set_error(ctx, GL_E_STATE, "predict needs set_signals and set_state first");
    if (max_iter < 0 || !coefficients_out) return gl_set_error(ctx, GL_E_INVALID, "bad argument");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    GdDev &P = s->dev;
    hipStream_t st_ = ctx->stream;
    double *st = (double *)ctx->pinned;
    GL_HIP(ctx, hipMemsetAsync(P.dual, 0, (size_t)P.Tw * P.n * 8, st_));
    GL_TRY(gd_launch_grad_c(s));  // the weights never change: one pass over the pairs
    int flags = 0;
    int64_t i = 0;
    for (i = 0; i < max_iter; ++i) {
        GL_HIP(ctx, hipMemsetAsync(P.stat + 2, 0, 8, st_));
        if (P.log_c > 0) k_gd_logc_reset<<<dim3((unsigned)((P.Tw + 255) / 256)), 256, 0, st_>>>(P);
        k_gd_coef_update<<<dim3((unsigned)P.Tw), 64, 0, st_>>>(P, 1);
        k_gd_coef_finish<<<1, 256, 0, st_>>>(P);
        k_gd_dual_update<<<dim3((unsigned)((P.n + 255) / 256), (unsigned)P.Tw), 256, 0, st_>>>(P, 1);
        k_gd_coef_commit<<<1, 256, 0, st_>>>(P);
        GL_HIP(ctx, hipGetLastError());
        GL_HIP(ctx, hipMemcpyAsync(st, P.stat, 4 * 8, hipMemcpyDeviceToHost, st_));
        GL_HIP(ctx, hipStreamSynchronize(st_));
        if (st[2] != 0.0) flags |= 1;
        if (st[3] < tol) {
            ++i;
            break;
        }
    }
    GL_HIP(ctx, hipMemcpyAsync(coefficients_out, P.C, (size_t)P.K * P.T * 8, hipMemcpyDeviceToHost, st_));
    GL_HIP(ctx, hipStreamSynchronize(st_));
    if (steps_run_out) *steps_run_out = i;
    if (flags_out) *flags_out = flags;
    return GL_OK;
}

// Measurement facility (tools / bench): average duration in microseconds of the two gradient
// kernels alone, HIP events on the session's stream; state unchanged.
GL_API int gl_graphdict_time_gradients(gl_graphdict *s, int reps, double *grad_c_us_out, double *grad_w_us_out) {
    GL_TRY(gd_check(s));
    gl_ctx *ctx = s->ctx;
    if (!s->have_x || !s->have_state || reps < 1) return gl_set_error(ctx, GL_E_STATE, "needs signals, state, reps >= 1");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    float ms = 0;
    for (int which = 0; which < 2; ++which) {
        GL_TRY(which ? gd_launch_grad_w(s) : gd_launch_grad_c(s));
        GL_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        for (int r = 0; r < reps; ++r) GL_TRY(which ? gd_launch_grad_w(s) : gd_launch_grad_c(s));
        GL_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        GL_HIP(ctx, hipEventSynchronize(ctx->ev1));
        GL_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        double *o = which ? grad_w_us_out : grad_c_us_out;
        if (o) *o = (double)ms * 1e3 / reps;
    }
    return GL_OK;
}

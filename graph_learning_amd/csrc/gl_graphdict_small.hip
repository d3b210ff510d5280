// GraphDictLog, small problems: the whole loop of _single_fit / predict in ONE launch of ONE workgroup.
// Reference: graph_learn/dictionary/base_model.py:355-527 (one step), :552-581 (loop of _single_fit),
// :647-676 (predict); dictionary/log_model.py:27-50 (adjoints, dual prox).
//
// The launch form (gl_graphdict.hip) spends 14 dependent kernels + a host round trip per step: about
// 70 us whatever the size.  The reference's own test shape (7 nodes, 23 samples, tests/test_models.py)
// and anything of that order is pure latency there.  Here the signals, the K x E weights, the K x T
// coefficients, the duals and the degrees live in LDS for the whole run, the phases of a step are
// separated by workgroup barriers, the stopping / plateau rules are evaluated by every thread from the
// same reduced values, and the history goes to a device buffer that is copied once at the end.
//
// Work split (1024 threads): the coefficient gradient over (sample, pair chunk), the weight gradient
// over (pair, sample chunk), C @ dual over (node, sample chunk); the chunk partials are folded in a
// fixed order, so a run is reproducible bit for bit.
#include "gl_common.h"
#include "gl_graphdict.h"
#include "gl_lanes.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int GS_NT = 1024;
constexpr int64_t GS_LDS_DOUBLES = 156 * 1024 / 8;  // dynamic LDS of the kernel (160 KiB per CU minus its static arrays)
// above this many pair-sample-atom products per step the launch form's kernels win (measured crossover,
// tools/graphdict_small_bench.py)
constexpr double GS_MAX_WORK = 4.0e5;

struct GdSmallArgs {
    int mode;  // 0: fit, 1: predict (weights fixed)
    int plateau;
    long long max_iter;
    double tol;
    double *history;  // fit: 2 doubles per step
    double *result;   // [0] steps run, [1] converged_, [2] flags, [3..5] step sizes at the end
    int PC, TC, MC;   // chunk counts of the three products
    int XS;           // row stride of the signals in LDS (odd)
    int oX, oW, oC, oCw, oDual, oDeg, oDegN, oPc, oPw, oPm, oPair;  // LDS offsets, in doubles
};

__device__ __forceinline__ double gs_wave_max(double v) { return wave_max(v); }  // gl_lanes.h (DPP)

// q = tid, tid + NT, ... split as (q / d, q % d) without a division inside the step loop: a runtime integer
// division is ~40 dependent instructions, and a lone workgroup is bound by exactly such chains
struct GsIdx {
    int a0, b0, sa, sb;  // first item of this thread and the step of (quotient, remainder)
};
__device__ __forceinline__ GsIdx gs_idx(int tid, int nt, int d) {
    GsIdx r;
    r.a0 = tid / d;
    r.b0 = tid - r.a0 * d;
    r.sa = nt / d;
    r.sb = nt - r.sa * d;
    return r;
}
#define GS_FOR2(IT, D, COUNT, A_, B_) \
    for (int q = tid, A_ = (IT).a0, B_ = (IT).b0; q < (COUNT); q += NT, A_ += (IT).sa, B_ += (IT).sb, A_ += (B_ >= (D)) ? 1 : 0, B_ -= (B_ >= (D)) ? (D) : 0)

template <int KP>
__global__ __launch_bounds__(GS_NT) void k_gd_small(GdDev P, GdSmallArgs A) {
    extern __shared__ double smem[];
    __shared__ double s_red[6][GS_NT / 64];
    __shared__ double s_hist[10][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NT = blockDim.x, NW = NT >> 6;
    const int n = P.n, T = P.T, W = P.W, Tw = P.Tw, K = P.K, XS = A.XS;
    const int E = n * (n - 1) / 2;
    double *sX = smem + A.oX, *sW = smem + A.oW, *sC = smem + A.oC, *sCw = smem + A.oCw, *sDual = smem + A.oDual;
    double *sDeg = smem + A.oDeg, *sDegN = smem + A.oDegN;
    double *sPc = smem + A.oPc, *sPw = smem + A.oPw, *sPm = smem + A.oPm;
    unsigned short *sI = reinterpret_cast<unsigned short *>(smem + A.oPair), *sJ = sI + E;
    const bool fit = A.mode == 0;
    const double Tn = (double)T;

    // ---- state -> LDS -------------------------------------------------------------------------
    for (int q = tid; q < n * T; q += NT) {
        const int i = q / T, t = q - i * T;
        sX[i * XS + t] = P.XT[(int64_t)i * P.Tp + t];
    }
    for (int e = tid; e < E; e += NT) {
        int i = 0, r = e;
        while (r >= n - 1 - i) {
            r -= n - 1 - i;
            ++i;
        }
        const int j = i + 1 + r;
        sI[e] = (unsigned short)i;
        sJ[e] = (unsigned short)j;
#pragma unroll
        for (int k = 0; k < KP; ++k) sW[e * KP + k] = P.WS[((int64_t)i * P.n_pad + j) * KP + k];
    }
    for (int q = tid; q < K * T; q += NT) sC[q] = P.C[q];
    for (int q = tid; q < Tw * n; q += NT) sDual[q] = P.dual[q];
    for (int q = tid; q < KP * n; q += NT) {
        const int k = q / n, i = q - k * n;
        sDeg[q] = P.deg[(int64_t)k * P.n_pad + i];
        sDegN[q] = 0.0;
    }
    __syncthreads();

    // sums (slots 0, 1, 3, 4) and maxima (slots 2, 5) over the workgroup, fixed order; the first barrier
    // also publishes what the phase before wrote to LDS
    auto reduce6 = [&](double(&v)[6]) {
        // the four sums in one halving butterfly (lane 8 q of the lower half holds sum q), the maxima by DPP
        const double four[4] = {v[0], v[1], v[3], v[4]};
        const double sums = add_xor32(halfwave_rows4(four, lane));
        const double m2 = gs_wave_max(v[2]), m5 = gs_wave_max(v[5]);
        __syncthreads();
        if (lane < 32 && (lane & 7) == 0) {
            const int q = lane >> 3;
            s_red[q < 2 ? q : q + 1][wave] = sums;
        }
        if (lane == 0) {
            s_red[2][wave] = m2;
            s_red[5][wave] = m5;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double r = s_red[c][0];
            for (int w = 1; w < NW; ++w) r = (c % 3 == 2) ? fmax(r, s_red[c][w]) : r + s_red[c][w];
            v[c] = r;
        }
    };

    const GsIdx ixT = gs_idx(tid, NT, T), ixE = gs_idx(tid, NT, E > 0 ? E : 1), ixN = gs_idx(tid, NT, n),
                ixTw = gs_idx(tid, NT, Tw);
    double step_c = P.step_c, step_w = P.step_w, step_dual = P.step_dual;
    int flags = 0;
    long long converged = -1, it = 0;
    for (it = 0; it < A.max_iter; ++it) {
        // log_c reset of the window view (base_model.py:385-387)
        if (P.log_c > 0) {
            for (int tw = tid; tw < Tw; tw += NT) {
                double s = 0;
                for (int k = 0; k < K; ++k) s += sC[k * T + tw * W];
                if (s < 1e-8)
                    for (int k = 0; k < K; ++k) sC[k * T + tw * W] = P.log_c;
            }
            __syncthreads();
        }
        // ---- the three products, chunk partials ---------------------------------------------------
        if (fit || it == 0) {  // W @ sq_pdiffs' (:375); predict: the weights never change
            GS_FOR2(ixT, T, T * A.PC, pc, t) {
                double acc[KP];
#pragma unroll
                for (int k = 0; k < KP; ++k) acc[k] = 0.0;
                for (int e = pc; e < E; e += A.PC) {
                    const double d = sX[sI[e] * XS + t] - sX[sJ[e] * XS + t];
                    const double sq = d * d;
#pragma unroll
                    for (int k = 0; k < KP; ++k) acc[k] = __builtin_fma(sW[e * KP + k], sq, acc[k]);  // rows k >= K are zero
                }
#pragma unroll
                for (int k = 0; k < KP; ++k) sPc[(pc * KP + k) * T + t] = acc[k];
            }
        }
        if (fit) {
            GS_FOR2(ixE, E, E * A.TC, tc, e) {  // C @ sq_pdiffs (:447)
                const double *xi = sX + sI[e] * XS, *xj = sX + sJ[e] * XS;
                double acc[KP];
#pragma unroll
                for (int k = 0; k < KP; ++k) acc[k] = 0.0;
                for (int t = tc; t < T; t += A.TC) {
                    const double d = xi[t] - xj[t];
                    const double sq = d * d;
#pragma unroll
                    for (int k = 0; k < KP; ++k)
                        if (k < K) acc[k] = __builtin_fma(sC[k * T + t], sq, acc[k]);
                }
#pragma unroll
                for (int k = 0; k < KP; ++k) sPw[(tc * E + e) * KP + k] = acc[k];
            }
            GS_FOR2(ixN, n, n * A.MC, mc, i) {  // C @ dual_rep (log_model.py:36-37)
                double acc[KP];
#pragma unroll
                for (int k = 0; k < KP; ++k) acc[k] = 0.0;
                for (int t = mc; t < T; t += A.MC) {
                    const double dv = sDual[(t / W) * n + i];
#pragma unroll
                    for (int k = 0; k < KP; ++k)
                        if (k < K) acc[k] = __builtin_fma(sC[k * T + t], dv, acc[k]);
                }
#pragma unroll
                for (int k = 0; k < KP; ++k) sPm[(mc * KP + k) * n + i] = acc[k];
            }
        }
        __syncthreads();
        // fold the chunks into chunk 0
        if (fit || it == 0)
            for (int q = tid; q < KP * T; q += NT) {
                double v = sPc[q];
                for (int c = 1; c < A.PC; ++c) v += sPc[c * KP * T + q];
                sPc[q] = v;
            }
        if (fit) {
            for (int q = tid; q < E * KP; q += NT) {
                double v = sPw[q];
                for (int c = 1; c < A.TC; ++c) v += sPw[c * E * KP + q];
                sPw[q] = v;
            }
            for (int q = tid; q < KP * n; q += NT) {
                double v = sPm[q];
                for (int c = 1; c < A.MC; ++c) v += sPm[c * KP * n + q];
                sPm[q] = v;
            }
        }
        __syncthreads();
        // ---- coefficient update (:373-401) and weight update (:444-461) ----------------------------
        double r[6] = {0, 0, 0, 0, 0, 0};  // [2] max |c_new|; [3..5] sum (w_new - w)^2, sum w^2, max |w|
        GS_FOR2(ixTw, Tw, K * Tw, k, tw) {
            double dterm = 0.0;  // W @ (S' dual') == <S W_k, dual_tw>  (log_model.py:33-34)
            for (int i = 0; i < n; ++i) dterm += sDeg[k * n + i] * sDual[tw * n + i];
            double tot = 0.0;
            for (int qq = 0; qq < W; ++qq) tot += sPc[k * T + tw * W + qq];
            double grad = W > 1 ? tot / (double)W : tot;
            grad += P.l1_c * Tn;
            const double *c = sC + k * T;  // the window view is c[m * W]
            if (P.l1_diff_c > 0) {  // :389-394
                auto sgn = [](double v) { return v > 0 ? 1.0 : (v < 0 ? -1.0 : 0.0); };
                const double right = tw + 1 < Tw ? sgn(c[(tw + 1) * W] - c[tw * W]) : 0.0;
                const double left = tw > 0 ? sgn(c[tw * W] - c[(tw - 1) * W]) : 0.0;
                grad -= P.l1_diff_c * Tn * (right - left);
            }
            double cn = c[tw * W] - step_c / Tn * (dterm + grad);
            cn = cn < 0 ? 0.0 : cn;
            cn = cn > 1 ? 1.0 : cn;
            sCw[q] = cn;
            r[2] = fmax(r[2], fabs(cn));
        }
        if (fit)
            for (int e = tid; e < E; e += NT) {
                const int i = sI[e], j = sJ[e];
                double wo[KP], wsum = 0.0;
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    wo[k] = sW[e * KP + k];
                    wsum += k < K ? wo[k] : 0.0;
                }
#pragma unroll
                for (int k = 0; k < KP; ++k)
                    if (k < K) {
                        double gk = sPw[e * KP + k] + P.l1_w * Tn;
                        if (P.ortho_w > 0) gk += P.ortho_w * Tn * (wsum - wo[k]);
                        const double dterm = sPm[k * n + i] + sPm[k * n + j];
                        double v = wo[k] - step_w / Tn * (dterm + gk);
                        v = v < 0 ? 0.0 : v;
                        r[3] += (v - wo[k]) * (v - wo[k]);
                        r[4] += wo[k] * wo[k];
                        r[5] = fmax(r[5], fabs(wo[k]));
                        sW[e * KP + k] = v;
                    }
            }
        reduce6(r);
        const double dw = r[5] <= 1e-8 ? sqrt(r[3]) : sqrt(r[3]) / sqrt(r[4]);  // evaluation.py:21-23
        if (r[2] <= 1e-8) {  // "All coefficients dropped to 0" (:406-408)
            for (int q = tid; q < K * Tw; q += NT) sCw[q] = P.log_c;
            flags |= 1;
        }
        if (fit)  // degrees of the new weights
            GS_FOR2(ixN, n, K * n, k, i) {
                double s = 0.0;
                for (int u = 0; u < n; ++u) {
                    if (u == i) continue;
                    const int a = min(i, u), b = max(i, u);
                    s += sW[(a * n - a * (a + 1) / 2 + (b - a - 1)) * KP + k];
                }
                sDegN[q] = s;
            }
        __syncthreads();
        // ---- dual update with the overshoot 2 new - old (log_model.py:39-50, base_model.py:509-514) ----
        GS_FOR2(ixN, n, Tw * n, tw, i) {
            double stp = 0.0;
            for (int k = 0; k < K; ++k) {
                const double dold = sDeg[k * n + i];
                const double dv = fit ? 2.0 * sDegN[k * n + i] - dold : dold;
                stp += dv * (2.0 * sCw[k * Tw + tw] - sC[k * T + tw * W]);
            }
            const double d = sDual[q] + step_dual * stp;
            sDual[q] = (d - sqrt(d * d + 4.0 * step_dual)) / 2.0;
        }
        __syncthreads();
        // ---- commit the coefficients, relative changes (:516-519) ------------------------------------
        double c6[6] = {0, 0, 0, 0, 0, 0};
        GS_FOR2(ixT, T, K * T, k, t) {
            const double nw = sCw[k * Tw + (W == 1 ? t : t / W)], o = sC[q];
            c6[0] += (nw - o) * (nw - o);
            c6[1] += o * o;
            c6[2] = fmax(c6[2], fabs(o));
            sC[q] = nw;
        }
        reduce6(c6);
        const double dnorm = sqrt(c6[0]);
        const double dc = c6[2] <= 1e-8 ? dnorm : dnorm / sqrt(c6[1]);
        if (fit) {
            double *tmp = sDeg;
            sDeg = sDegN;
            sDegN = tmp;
            if (tid == 0) {
                A.history[2 * it] = dc;
                A.history[2 * it + 1] = dw;
            }
            if (dc < A.tol && dw < A.tol) converged = it;
            if (A.plateau) {  // :563-574, every thread takes the same decision from the same ten entries
                if (it > 100) {
                    double mc = 0, mw = 0;
                    for (long long qq = it - 10; qq < it; ++qq) {
                        mc += s_hist[qq % 10][0];
                        mw += s_hist[qq % 10][1];
                    }
                    if (dc >= mc / 10 && dw >= mw / 10) {
                        step_c *= 0.7, step_w *= 0.7, step_dual *= 0.7;
                        flags |= 2;
                    }
                }
                __syncthreads();
                if (tid == 0) s_hist[it % 10][0] = dc, s_hist[it % 10][1] = dw;
            }
            if (converged > 0) {  // sic: a stop at step 0 is not taken (:578)
                ++it;
                break;
            }
        } else if (dnorm < A.tol) {  // predict (:671)
            ++it;
            break;
        }
    }
    __syncthreads();
    // ---- LDS -> state ---------------------------------------------------------------------------
    if (fit) {
        for (int e = tid; e < E; e += NT) {
            const int i = sI[e], j = sJ[e];
#pragma unroll
            for (int k = 0; k < KP; ++k) P.WS[((int64_t)i * P.n_pad + j) * KP + k] = sW[e * KP + k];
        }
        for (int q = tid; q < KP * n; q += NT) {
            const int k = q / n, i = q - k * n;
            P.deg[(int64_t)k * P.n_pad + i] = sDeg[q];
        }
    }
    for (int q = tid; q < K * T; q += NT) {
        const int k = q / T, t = q - k * T;
        P.C[q] = sC[q];
        P.CT[(int64_t)t * KP + k] = sC[q];
    }
    for (int q = tid; q < Tw * n; q += NT) P.dual[q] = sDual[q];
    if (tid == 0) {
        A.result[0] = (double)it;
        A.result[1] = (double)converged;
        A.result[2] = (double)flags;
        A.result[3] = step_c;
        A.result[4] = step_w;
        A.result[5] = step_dual;
    }
}

// LDS layout and chunk counts; false when the problem does not fit one workgroup
bool gs_layout(const GdDev &P, GdSmallArgs &A, int nt) {
    const int64_t n = P.n, T = P.T, Tw = P.Tw, K = P.K, KP = P.KP;
    const int64_t E = n * (n - 1) / 2;
    if (n > 512 || E > 16384 || T > 16384) return false;
    A.XS = (int)(T | 1);
    int64_t o = 0;
    auto take = [&](int64_t cnt) {
        const int64_t at = o;
        o += cnt;
        return (int)at;
    };
    A.oX = take(n * A.XS);
    A.oW = take(E * KP);
    A.oC = take(K * T);
    A.oCw = take(KP * Tw);
    A.oDual = take(Tw * n);
    A.oDeg = take(KP * n);
    A.oDegN = take(KP * n);
    A.oPair = take((E * 4 + 7) / 8);
    const int64_t budget = GS_LDS_DOUBLES - o;
    const int64_t uc = KP * T, uw = E * KP, um = KP * n;  // one chunk of each product
    if (budget < uc + uw + um) return false;
    int64_t pc = std::max<int64_t>(1, std::min<int64_t>(E, nt / T));
    int64_t tc = std::max<int64_t>(1, std::min<int64_t>(T, nt / std::max<int64_t>(E, 1)));
    int64_t mc = std::max<int64_t>(1, std::min<int64_t>(T, nt / n));
    mc = std::min<int64_t>(mc, 16);
    while (pc * uc + tc * uw + mc * um > budget) {  // halve the largest region that still has chunks to give
        int64_t *best = nullptr, bsz = 0;
        if (pc > 1 && pc * uc > bsz) best = &pc, bsz = pc * uc;
        if (tc > 1 && tc * uw > bsz) best = &tc, bsz = tc * uw;
        if (mc > 1 && mc * um > bsz) best = &mc, bsz = mc * um;
        if (!best) return false;
        *best = (*best + 1) / 2;
    }
    A.PC = (int)pc, A.TC = (int)tc, A.MC = (int)mc;
    A.oPc = take(pc * uc);
    A.oPw = take(tc * uw);
    A.oPm = take(mc * um);
    return o <= GS_LDS_DOUBLES;
}

// threads of the workgroup: barriers and the cross-wave folds cost per wave, so tiny problems take fewer
int gs_threads(const GdDev &P) {
    if (const char *e = getenv("GRAPHLEARN_AMD_GD_SMALL_NT")) {  // tuning aid
        const int v = atoi(e);
        if (v == 64 || v == 128 || v == 256 || v == 512 || v == 1024) return v;
    }
    const double work = 0.5 * P.n * (P.n - 1.0) * P.T * (P.K + 2.0);
    return work <= 2.0e4 ? 256 : (work <= 1.0e5 ? 512 : GS_NT);
}

// one launch; afterwards host_out holds the 8 result words followed by the history of the run (fit)
int gs_run(gl_graphdict *s, int mode, int64_t max_iter, double tol, int plateau, std::vector<double> &host_out) {
    gl_ctx *ctx = s->ctx;
    GdDev &P = s->dev;
    GdSmallArgs A{};
    const int nt = gs_threads(P);
    if (!gs_layout(P, A, nt)) return gl_set_error(ctx, GL_E_STATE, "problem does not fit the single-workgroup engine");
    A.mode = mode, A.plateau = plateau, A.max_iter = max_iter, A.tol = tol;
    const int64_t hist_steps = mode == 0 ? max_iter : 0;
    if (!s->small_hist || s->small_hist_steps < hist_steps) {  // [8 result words][2 per step]
        double *h = nullptr;
        GL_MALLOC(ctx, (void **)&h, (size_t)(8 + 2 * hist_steps) * sizeof(double), "graphdict");
        if (s->small_hist) {
            auto itp = std::find(s->allocs.begin(), s->allocs.end(), (void *)s->small_hist);
            if (itp != s->allocs.end()) s->allocs.erase(itp);
            (void)hipFree(s->small_hist);
        }
        s->small_hist = h;
        s->small_hist_steps = hist_steps;
        s->allocs.push_back(h);
    }
    A.result = s->small_hist;
    A.history = s->small_hist + 8;
    const int64_t total = (int64_t)A.oPm + (int64_t)A.MC * P.KP * P.n;
    const size_t lds = (size_t)total * sizeof(double);
    const void *fn = P.KP == 4 ? (const void *)k_gd_small<4> : (P.KP == 8 ? (const void *)k_gd_small<8> : (const void *)k_gd_small<16>);
    if (!s->small_attr_set) {
        GL_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(GS_LDS_DOUBLES * 8)));
        s->small_attr_set = true;
    }
    switch (P.KP) {
        case 4: k_gd_small<4><<<1, nt, lds, ctx->stream>>>(P, A); break;
        case 8: k_gd_small<8><<<1, nt, lds, ctx->stream>>>(P, A); break;
        default: k_gd_small<16><<<1, nt, lds, ctx->stream>>>(P, A); break;
    }
    GL_HIP(ctx, hipGetLastError());
    host_out.resize((size_t)(8 + 2 * hist_steps));
    GL_TRY(gl_d2h(ctx, host_out.data(), s->small_hist, host_out.size() * sizeof(double)));
    return GL_OK;
}

}  // namespace

// GRAPHLEARN_AMD_GD_ENGINE = launch | small forces the form (small: whenever the problem fits LDS)
bool gd_small_applicable(const gl_graphdict *s) {
    const GdDev &P = s->dev;
    const char *e = getenv("GRAPHLEARN_AMD_GD_ENGINE");
    if (e && !strcmp(e, "launch")) return false;
    GdSmallArgs A{};
    if (!gs_layout(P, A, gs_threads(P))) return false;
    if (e && !strcmp(e, "small")) return true;
    const double E = 0.5 * P.n * (P.n - 1.0);
    return E * P.T * (P.K + 2.0) <= GS_MAX_WORK;
}

int gd_small_fit(gl_graphdict *s, int64_t max_iter, double tol, int reduce_step_on_plateau, double *history_out,
                 int64_t *steps_run_out, int64_t *converged_out, int *flags_out) {
    GdDev &P = s->dev;
    for (int64_t q = 0; q < 2 * max_iter; ++q) history_out[q] = -1.0;
    int64_t ran = 0, converged = -1;
    int flags = 0;
    if (max_iter > 0) {
        std::vector<double> out;
        GL_TRY(gs_run(s, 0, max_iter, tol, reduce_step_on_plateau, out));
        ran = (int64_t)out[0], converged = (int64_t)out[1], flags = (int)out[2];
        P.step_c = out[3], P.step_w = out[4], P.step_dual = out[5];  // plateau rule
        std::copy(out.begin() + 8, out.begin() + 8 + 2 * ran, history_out);
    }
    if (steps_run_out) *steps_run_out = ran;
    if (converged_out) *converged_out = converged;
    if (flags_out) *flags_out = flags;
    return GL_OK;
}

int gd_small_predict(gl_graphdict *s, int64_t max_iter, double tol, int64_t *steps_run_out, int *flags_out) {
    int64_t ran = 0;
    int flags = 0;
    if (max_iter > 0) {
        std::vector<double> out;
        GL_TRY(gs_run(s, 1, max_iter, tol, 0, out));
        ran = (int64_t)out[0], flags = (int)out[2];
    }
    if (steps_run_out) *steps_run_out = ran;
    if (flags_out) *flags_out = flags;
    return GL_OK;
}

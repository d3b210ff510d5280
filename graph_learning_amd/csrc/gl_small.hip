// Small-graph engine of the log-degree loop (n <= 128): ALL iterations of a call in one launch of ONE
// workgroup, the whole problem on one CU.
//
// Reference path: graph_learn/smooth_learning.py:146-167 (same arithmetic as gl_logdeg.hip).  Why:
// below a few hundred nodes an iteration of the streaming engine is two dependent kernel launches
// (~9 us) for a few microseconds of data; the reference's own CPU-runnable case (BASELINE configs[0],
// n = 64) and the windows / clusters of its batched callers live in this regime.  Here the edge
// iterate w and the prox point p sit in LDS in condensed order (E <= 8128 doubles each), z and the
// edge's (i, j) in registers (<= 8 edges per thread), the node vectors in LDS; the two grid-wide
// dependencies of an iteration are workgroup barriers, and the stopping rule is evaluated in place,
// exactly when the reference evaluates it (no deferral).  Deterministic: every sum has a fixed order.
//
// The session keeps its tiled layout: the call converts w (and once z) to condensed scratch buffers
// and back, so every other entry point sees a consistent state afterwards.
#include <algorithm>

#include "gl_lanes.h"
#include "gl_logdeg.h"

namespace {

constexpr int SMALL_MAX_N = 128;
constexpr int SMALL_EPT = 8;   // edges per thread: 8128 / 1024 at n = 128, 2016 / 256 at n = 64

struct SmallDev {
    double *wc;         // condensed iterate (E), in / out
    const double *zc;   // condensed scaled distances (E)
    double *dvar, *ybar, *pbar, *dprev, *bnorm, *dstat;
    long long *ctr;
    int n, E, nb;
    double gamma, c, g2, a4g, tol;
    long long it0;
    int maxit;
};

// T threads: 1024 (n <= 128, 8 threads per node in the node phase) or 256 (n <= 64, 4 per node: fewer
// waves to synchronise and to reduce over when the graph is tiny)
template <int T>
__global__ __launch_bounds__(T) void k_small(SmallDev P) {
    constexpr int ST = T, NW = T / 64, PARTS = T / 128 >= 8 ? 8 : 4;
    extern __shared__ double sh[];
    const int n = P.n, E = P.E;
    double *w_l = sh;               // E
    double *p_l = sh + E;           // E
    double *d_l = p_l + E;          // n
    double *yb_l = d_l + n;         // n
    double *pb_l = yb_l + n;        // n
    double *dp_l = pb_l + n;        // n: d of the previous iteration
    double *red = dp_l + n;         // [4][NW] norm partials of the waves
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double c = P.c, g = P.gamma, g2 = P.g2;

    // my edges: e = tid + ST k; z and the packed end points stay in registers
    double z[SMALL_EPT], wr[SMALL_EPT];  // z and my edges' w stay in registers; LDS holds the copy the node phase sums
    int ij[SMALL_EPT];
#pragma unroll
    for (int k = 0; k < SMALL_EPT; ++k) {
        const int e = tid + ST * k;
        z[k] = 0.0;
        wr[k] = 0.0;
        ij[k] = -1;
        if (e < E) {
            // inverse of the condensed index e = i (2n - i - 1) / 2 + (j - i - 1)
            const double b = 2.0 * n - 1.0;
            int i = (int)((b - sqrt(b * b - 8.0 * (double)e)) * 0.5);
            i = max(0, min(i, n - 2));
            while (i > 0 && i * (2 * n - i - 1) / 2 > e) --i;
            while (i + 1 < n - 1 && (i + 1) * (2 * n - i - 2) / 2 <= e) ++i;
            const int j = e - i * (2 * n - i - 1) / 2 + i + 1;
            ij[k] = i | (j << 16);
            z[k] = P.zc[e];
            wr[k] = P.wc[e];
        }
    }
    if (tid < n) {
        d_l[tid] = P.dvar[tid];
        yb_l[tid] = P.ybar[tid];
        pb_l[tid] = P.pbar[tid];
        dp_l[tid] = P.dprev[tid];
    }
    __syncthreads();

    const int node = tid / PARTS, part = tid % PARTS;  // node phase: PARTS threads per node
    int ran = 0, stopped = 0;
    double rp = 0, rd = 0, tot[4] = {0, 0, 0, 0};
    for (int iter = 0; iter < P.maxit; ++iter) {
        // ---- edge phase (:147-153,163): y, p, q, w <- (w - y) + q, norms of q - y and w
        double nqy = 0, nw = 0;
#pragma unroll
        for (int k = 0; k < SMALL_EPT; ++k) {
            if (ij[k] < 0) continue;
            const int e = tid + ST * k, i = ij[k] & 0xffff, j = ij[k] >> 16;
            const double w = wr[k];
            const double y = c * w - g * (d_l[i] + d_l[j]);
            const double tt = y - g2 * z[k];
            const double pr = tt > 0 ? tt : 0.0;
            const double q = c * pr - g * (pb_l[i] + pb_l[j]);
            const double dq = q - y;
            nqy += dq * dq;
            nw += w * w;
            wr[k] = (w - y) + q;
            p_l[e] = pr;
            w_l[e] = wr[k];
        }
        __syncthreads();
        // ---- node phase: S p and S w_new in fixed order (8 interleaved partials per node, combined
        //      inside the 8-lane group), then the dual update (:154,159,164) and the next ybar / pbar (:148,151)
        double ndq = 0, nd = 0;
        if (node < n) {
            double sp = 0, sw = 0;
            for (int k = part; k < n - 1; k += PARTS) {
                const int o = k < node ? k : k + 1;
                const int a = min(node, o), b = max(node, o);
                const int e = a * (2 * n - a - 1) / 2 + (b - a - 1);
                sp += p_l[e];
                sw += w_l[e];
            }
            sp += dpp_mov<DPP_QUAD_XOR1>(sp);
            sw += dpp_mov<DPP_QUAD_XOR1>(sw);
            sp += dpp_mov<DPP_QUAD_XOR2>(sp);
            sw += dpp_mov<DPP_QUAD_XOR2>(sw);
            if (PARTS == 8) {
                sp += dpp_mov<DPP_HALF_MIRROR>(sp);
                sw += dpp_mov<DPP_HALF_MIRROR>(sw);
            }
            if (part == 0) {
                const double d_old = d_l[node], yb = yb_l[node], pb = pb_l[node];
                const double qb = pb + g * sp;
                const double dq = qb - yb;
                const double dnew = (d_old - yb) + qb;
                const double ybn = dnew + g * sw;
                ndq = dq * dq;
                nd = d_old * d_old;
                dp_l[node] = d_old;
                d_l[node] = dnew;
                yb_l[node] = ybn;
                pb_l[node] = (ybn - sqrt(ybn * ybn + P.a4g)) / 2;
            }
        }
        // ---- norms: waves in fixed order
        nqy = wave_sum(nqy);
        nw = wave_sum(nw);
        ndq = wave_sum(ndq);
        nd = wave_sum(nd);
        if (lane == 0) {
            red[0 * NW + wave] = nqy;
            red[1 * NW + wave] = nw;
            red[2 * NW + wave] = ndq;
            red[3 * NW + wave] = nd;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double t = 0;
#pragma unroll
            for (int v = 0; v < NW; ++v) t += red[q * NW + v];
            tot[q] = t;
        }
        ran = iter + 1;
        // the reference's test sqrt(a)/sqrt(b) < tol (:156-161,166; never at iteration 0) costs four
        // square roots and two divisions in fp64; a factor-4 screen on the squares skips them while
        // the ratios are clearly above tol (every thread takes the same branch: same LDS values)
        const double t2 = 4.0 * P.tol * P.tol;
        if (P.it0 + iter >= 1 && tot[0] < t2 * tot[1] && tot[2] < t2 * tot[3]) {
            rp = sqrt(tot[0]) / sqrt(tot[1]);
            rd = sqrt(tot[2]) / sqrt(tot[3]);
            if (rp < P.tol && rd < P.tol) {
                stopped = 1;
                break;
            }
        }
        __syncthreads();  // red is rewritten next iteration
    }
    __syncthreads();
    rp = sqrt(tot[0]) / sqrt(tot[1]);  // the values of the last iteration, for the caller
    rd = sqrt(tot[2]) / sqrt(tot[3]);
    // ---- state back to the session's arrays
#pragma unroll
    for (int k = 0; k < SMALL_EPT; ++k) {
        const int e = tid + ST * k;
        if (e < E) P.wc[e] = wr[k];
    }
    if (tid < n) {
        P.dvar[tid] = d_l[tid];
        P.ybar[tid] = yb_l[tid];
        P.pbar[tid] = pb_l[tid];
        P.dprev[tid] = dp_l[tid];
    }
    if (tid == 0) {
        const long long itf = P.it0 + ran;
        if (ran > 0 && itf - 1 >= 1) {
            P.dstat[0] = rp;
            P.dstat[1] = rd;
        }
        if (ran > 0) {  // norm partials of the last iteration where the streaming engine's deferred rule looks
            double *bn = P.bnorm + ((itf - 1) & 1) * 4 * P.nb;
            for (int b = 0; b < P.nb; ++b)
                for (int q = 0; q < 4; ++q) bn[4 * b + q] = b == 0 ? tot[q] : 0.0;
        }
        P.ctr[0] = stopped;
        P.ctr[1] = itf;
        P.ctr[2] = itf;
        P.ctr[3] = itf > 0 ? itf - 1 : 0;
    }
}

}  // namespace

struct gl_small {
    double *wc = nullptr, *zc = nullptr;
    bool have_z = false;
    size_t lds = 0;
};

bool gl_small_applicable(const gl_logdeg *s) {
    if (s->n > SMALL_MAX_N) return false;
    if (const char *e = getenv("GRAPHLEARN_AMD_ENGINE"))
        if (!strcmp(e, "stream") || !strcmp(e, "resident")) return false;  // another engine is asked for
    return true;
}

void gl_small_free(gl_logdeg *s) {
    if (!s || !s->small) return;
    if (s->small->wc) (void)hipFree(s->small->wc);
    if (s->small->zc) (void)hipFree(s->small->zc);
    delete s->small;
    s->small = nullptr;
}

void gl_small_invalidate_z(gl_logdeg *s) {
    if (s && s->small) s->small->have_z = false;
}

// Runs up to `iters` iterations from the session's current state (asynchronously on the context's
// stream).  The session's tiled iterate is refreshed in BOTH ping-pong buffers afterwards, so the
// parity of the final iteration count does not matter; s->it is NOT advanced here.
int gl_small_launch(gl_logdeg *s, int64_t iters, double tol) {
    gl_ctx *ctx = s->ctx;
    const LogdegDev &D = s->dev;
    const int n = (int)s->n, E = (int)s->n_edges;
    if (!s->small) {
        s->small = new (std::nothrow) gl_small();
        if (!s->small) return GL_E_NOMEM;
        GL_HIP(ctx, hipMalloc((void **)&s->small->wc, (size_t)std::max(E, 1) * 8));
        GL_HIP(ctx, hipMalloc((void **)&s->small->zc, (size_t)std::max(E, 1) * 8));
        s->small->lds = ((size_t)2 * E + 4 * n + 64) * 8;
        GL_HIP(ctx, hipFuncSetAttribute((const void *)k_small<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->small->lds));
    }
    gl_small *m = s->small;
    if (!m->have_z) {
        GL_TRY(gl_logdeg_tiled_to_condensed(s, D.z, m->zc));
        m->have_z = true;
    }
    GL_TRY(gl_logdeg_tiled_to_condensed(s, D.w[s->it & 1], m->wc));
    SmallDev P{};
    P.wc = m->wc;
    P.zc = m->zc;
    P.dvar = D.dvar;
    P.ybar = D.ybar;
    P.pbar = D.pbar;
    P.dprev = D.dprev;
    P.bnorm = D.bnorm;
    P.dstat = D.dstat;
    P.ctr = D.ctr;
    P.n = n;
    P.E = E;
    P.nb = (int)D.nb;
    P.gamma = D.gamma;
    P.c = D.c;
    P.g2 = D.g2;
    P.a4g = D.a4g;
    P.tol = tol;
    P.it0 = s->it;
    P.maxit = (int)std::min<int64_t>(iters, 1 << 30);
    if (n <= 64)
        k_small<256><<<1, 256, m->lds, ctx->stream>>>(P);
    else
        k_small<1024><<<1, 1024, m->lds, ctx->stream>>>(P);
    GL_HIP(ctx, hipGetLastError());
    GL_TRY(gl_logdeg_condensed_to_tiled(s, m->wc, D.w[0]));
    GL_TRY(gl_logdeg_condensed_to_tiled(s, m->wc, D.w[1]));
    return GL_OK;
}

// Small-graph engine of the log-degree loop (n <= 128): ALL iterations of a call in one launch, ONE
// workgroup per graph, the whole problem on one CU.
//
// Reference path: graph_learn/smooth_learning.py:146-167 (same arithmetic as gl_logdeg.hip).  Why:
// below a few hundred nodes an iteration of the streaming engine is two dependent kernel launches
// (~9 us) for a few microseconds of data; the reference's own CPU-runnable case (BASELINE configs[0],
// n = 64) and the windows / clusters of its batched callers live in this regime.  Here the prox point p
// and a copy of w sit in LDS in condensed order (E <= 8128 doubles each), z, the edge's (i, j) and its
// own w in registers (<= 8 edges per thread), the node vectors in LDS; the two grid-wide dependencies of
// an iteration are workgroup barriers, and the stopping rule is evaluated in place, exactly when the
// reference evaluates it (no deferral).  Deterministic: every sum has a fixed order.
//
// Batched form (the 2-D branch, :128-129,158-159): one workgroup per graph, a grid barrier per
// iteration to exchange the four squared norms of every graph.  They are the diagonals of the Gram
// matrices whose largest eigenvalues decide the coupled stopping rule, and lambda_max lies in
// [max diagonal, trace]: while those bounds keep the ratios above tol the loop goes on; when they
// straddle tol (or allow a stop) all workgroups leave with their state saved and the host evaluates
// the exact rule (gl_logdeg_batch_iterate), then re-enters.
//
// The kernels read and write the session's tiled layout directly (edge (i, j) -> tile offset), so every
// other entry point sees a consistent state afterwards.
#include <algorithm>
#include <vector>

#include "gl_gridbar.h"
#include "gl_lanes.h"
#include "gl_logdeg.h"

namespace {

constexpr int SMALL_MAX_N = 128;
constexpr int SMALL_EPT = 8;  // edges per thread: 8128 / 1024 at n = 128, 2016 / 256 at n = 64

struct SmallDev {
    double *w[2];       // tiled ping-pong iterate of the session: current iterate in w[it0 & 1]
    const double *z;    // tiled scaled distances
    double *dvar, *ybar, *pbar, *dprev, *bnorm, *dstat;
    long long *ctr;
    int n, E, nb;
    double gamma, c, g2, a4g, tol;
    long long it0;
    int maxit;
};

struct SmallBatch {
    const SmallDev *devs;  // one per graph (device array)
    double *diag;          // [B][4] squared norms of the last iteration
    long long *status;     // [0] iterations run by this launch, [1] 0 = maxit reached, 1 = the host must decide, 2 = barrier lost
    unsigned *bar;         // grid barrier words (gl_gridbar.h)
    int B;
    int rendezvous;        // workgroups expected at the start-up rendezvous (= B; B + 1 in the fallback test)
};

__device__ __forceinline__ int64_t tiled_offset(int i, int j, int nb) {  // i < j
    const int I = i >> 6, J = j >> 6;
    const int64_t t = (int64_t)I * nb - (int64_t)I * (I - 1) / 2 + (J - I);
    return t * GL_TILE_ELEMS + (int64_t)(i & 63) * GL_TB + (j & 63);
}

// T threads: 1024 (n <= 128, 8 threads per node in the node phase) or 256 (n <= 64, 4 per node: fewer
// waves to synchronise and to reduce over when the graph is tiny).  BATCH: see the header comment.
template <int T, bool BATCH>
__global__ __launch_bounds__(T) void k_small(SmallDev P1, SmallBatch Q) {
    constexpr int ST = T, NW = T / 64, PARTS = T / 128 >= 8 ? 8 : 4;
    extern __shared__ double sh[];
    __shared__ int s_ok;
    const SmallDev P = BATCH ? Q.devs[blockIdx.x] : P1;
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

    // my edges: e = tid + ST k; z, the packed end points, w and the previous w stay in registers
    double z[SMALL_EPT], wr[SMALL_EPT], wo[SMALL_EPT];
    int ij[SMALL_EPT];
    const double *w_in = P.w[P.it0 & 1];
#pragma unroll
    for (int k = 0; k < SMALL_EPT; ++k) {
        const int e = tid + ST * k;
        z[k] = wr[k] = wo[k] = 0.0;
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
            const int64_t off = tiled_offset(i, j, P.nb);
            z[k] = P.z[off];
            wr[k] = w_in[off];
            wo[k] = wr[k];
        }
    }
    if (tid < n) {
        d_l[tid] = P.dvar[tid];
        yb_l[tid] = P.ybar[tid];
        pb_l[tid] = P.pbar[tid];
        dp_l[tid] = P.dprev[tid];
    }
    const GlGridBar gb{BATCH ? Q.bar : nullptr, BATCH ? Q.rendezvous : 0};
    GlBarState bs{};
    bool alive = true;
    if (BATCH) {
        if (tid == 0) s_ok = gl_barrier_setup(gb, bs) ? 1 : 0;
        __syncthreads();
        alive = s_ok != 0;
    }
    __syncthreads();

    const int node = tid / PARTS, part = tid % PARTS;  // node phase: PARTS threads per node
    int ran = 0, stopped = 0, reason = 0;
    double tot[4] = {0, 0, 0, 0};
    for (int iter = 0; alive && iter < P.maxit; ++iter) {
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
            wo[k] = w;
            wr[k] = (w - y) + q;
            p_l[e] = pr;
            w_l[e] = wr[k];
        }
        __syncthreads();
        // ---- node phase: S p and S w_new in fixed order (PARTS interleaved partials per node, combined
        //      inside the lane group), then the dual update (:154,159,164) and the next ybar / pbar (:148,151)
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
        if (!BATCH) {
            // the reference's test sqrt(a)/sqrt(b) < tol (:156-161,166; never at iteration 0) costs four
            // square roots and two divisions in fp64; a factor-4 screen on the squares skips them while
            // the ratios are clearly above tol (every thread takes the same branch: same LDS values)
            const double t2 = 4.0 * P.tol * P.tol;
            if (P.it0 + iter >= 1 && tot[0] < t2 * tot[1] && tot[2] < t2 * tot[3]) {
                const double rp = sqrt(tot[0]) / sqrt(tot[1]), rd = sqrt(tot[2]) / sqrt(tot[3]);
                if (rp < P.tol && rd < P.tol) {
                    stopped = 1;
                    break;
                }
            }
            __syncthreads();  // red is rewritten next iteration
        } else {
            // coupled rule: publish my four squared norms, meet, read everybody's
            if (tid == 0)
                for (int q = 0; q < 4; ++q)
                    __hip_atomic_store((unsigned long long *)(Q.diag + 4 * blockIdx.x + q),
                                       (unsigned long long)__double_as_longlong(tot[q]), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            if (!gl_grid_barrier(gb, bs, &s_ok)) {
                reason = 2;  // barrier lost: the host reports it
                break;
            }
            if (P.it0 + iter >= 1) {
                double mx[4] = {0, 0, 0, 0}, tr[4] = {0, 0, 0, 0};
                for (int b = 0; b < Q.B; ++b)  // same data, same order in every workgroup
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double v = __longlong_as_double((long long)__hip_atomic_load(
                            (const unsigned long long *)(Q.diag + 4 * b + q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        mx[q] = fmax(mx[q], v);
                        tr[q] += v;
                    }
                // rel_p^2 in [mx0 / tr1, tr0 / mx1], rel_d^2 in [mx2 / tr3, tr2 / mx3]
                const double t2 = P.tol * P.tol * (1 + 1e-9);
                const bool degenerate = !(mx[1] > 0) || !(mx[3] > 0);
                const bool surely_not = !degenerate && (mx[0] > t2 * tr[1] || mx[2] > t2 * tr[3]);
                if (!surely_not) {
                    reason = 1;  // the exact spectral norms are needed: the host decides
                    break;
                }
            }
            // the barrier of the next iteration separates these reads from the next round of writes
        }
    }
    __syncthreads();
    // ---- state back to the session's arrays: the iterate into the buffer of its parity, the previous one
    //      into the other (the batched rule's exact evaluation and the streaming engine read both)
    {
        const long long itf = P.it0 + ran;
        double *w_cur = P.w[itf & 1], *w_prev = P.w[(itf + 1) & 1];
#pragma unroll
        for (int k = 0; k < SMALL_EPT; ++k)
            if (ij[k] >= 0) {
                const int64_t off = tiled_offset(ij[k] & 0xffff, ij[k] >> 16, P.nb);
                w_cur[off] = wr[k];
                w_prev[off] = wo[k];
            }
        if (tid < n) {
            P.dvar[tid] = d_l[tid];
            P.ybar[tid] = yb_l[tid];
            P.pbar[tid] = pb_l[tid];
            P.dprev[tid] = dp_l[tid];
        }
        if (tid == 0) {
            if (ran > 0 && itf - 1 >= 1) {  // the values of the last iteration, for the caller
                P.dstat[0] = sqrt(tot[0]) / sqrt(tot[1]);
                P.dstat[1] = sqrt(tot[2]) / sqrt(tot[3]);
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
            if (BATCH && blockIdx.x == 0) {
                Q.status[0] = ran;
                Q.status[1] = alive ? reason : 2;
            }
        }
    }
}

SmallDev small_dev_of(const gl_logdeg *s, double tol, int64_t iters) {
    const LogdegDev &D = s->dev;
    SmallDev P{};
    P.w[0] = D.w[0];
    P.w[1] = D.w[1];
    P.z = D.z;
    P.dvar = D.dvar;
    P.ybar = D.ybar;
    P.pbar = D.pbar;
    P.dprev = D.dprev;
    P.bnorm = D.bnorm;
    P.dstat = D.dstat;
    P.ctr = D.ctr;
    P.n = (int)s->n;
    P.E = (int)s->n_edges;
    P.nb = (int)D.nb;
    P.gamma = D.gamma;
    P.c = D.c;
    P.g2 = D.g2;
    P.a4g = D.a4g;
    P.tol = tol;
    P.it0 = s->it;
    P.maxit = (int)std::min<int64_t>(iters, 1 << 30);
    return P;
}

size_t small_lds_bytes(int n, int E) { return ((size_t)2 * E + 4 * n + 64) * 8; }

int small_set_attrs(gl_ctx *ctx) {  // per context: one process may drive several GPUs
    if (ctx->small_attrs_set) return GL_OK;
    const int most = (int)small_lds_bytes(SMALL_MAX_N, SMALL_MAX_N * (SMALL_MAX_N - 1) / 2);
    GL_HIP(ctx, hipFuncSetAttribute((const void *)k_small<1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize, most));
    GL_HIP(ctx, hipFuncSetAttribute((const void *)k_small<1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, most));
    ctx->small_attrs_set = true;
    return GL_OK;
}

}  // namespace

bool gl_small_applicable(const gl_logdeg *s) {
    if (s->n > SMALL_MAX_N) return false;
    if (const char *e = getenv("GRAPHLEARN_AMD_ENGINE"))
        if (!strcmp(e, "stream") || !strcmp(e, "resident")) return false;  // another engine is asked for
    return true;
}

// Runs up to `iters` iterations from the session's current state (asynchronously on the context's
// stream); s->it is NOT advanced here (the kernel leaves the final count in ctr[1]).
int gl_small_launch(gl_logdeg *s, int64_t iters, double tol) {
    gl_ctx *ctx = s->ctx;
    const SmallDev P = small_dev_of(s, tol, iters);
    const size_t lds = small_lds_bytes(P.n, P.E);
    GL_TRY(small_set_attrs(ctx));
    const SmallBatch none{};
    if (P.n <= 64)
        k_small<256, false><<<1, 256, lds, ctx->stream>>>(P, none);
    else
        k_small<1024, false><<<1, 1024, lds, ctx->stream>>>(P, none);
    GL_HIP(ctx, hipGetLastError());
    return GL_OK;
}

// Batched form: B sessions (same context, same n <= 128, same iteration count) advance together for up
// to `iters` iterations or until the norm bounds can no longer exclude a stop.  Synchronous; advances
// every session's iteration count.  need_exact: the host has to evaluate the exact rule now.
// Device scratch layout: SmallDev[B] | diag[4 B] | status[2] | bar[GL_BAR_WORDS].
size_t gl_small_batch_scratch_bytes(int B) {
    return (size_t)B * sizeof(SmallDev) + (size_t)4 * B * 8 + 2 * 8 + GL_BAR_WORDS * 4 + 64;
}

int gl_small_batch_run(gl_logdeg **sessions, int B, int64_t iters, double tol, void *dev_scratch, int64_t *ran_out,
                       bool *need_exact_out) {
    gl_ctx *ctx = sessions[0]->ctx;
    char *base = (char *)dev_scratch;
    SmallDev *devs = (SmallDev *)base;
    double *diag = (double *)(base + (size_t)B * sizeof(SmallDev));
    long long *status = (long long *)(diag + 4 * B);
    unsigned *bar = (unsigned *)(status + 2);
    std::vector<SmallDev> host((size_t)B);
    for (int a = 0; a < B; ++a) host[(size_t)a] = small_dev_of(sessions[a], tol, iters);
    GL_HIP(ctx, hipMemcpyAsync(devs, host.data(), host.size() * sizeof(SmallDev), hipMemcpyHostToDevice, ctx->stream));
    GL_HIP(ctx, hipMemsetAsync(status, 0, 2 * 8 + GL_BAR_WORDS * 4, ctx->stream));
    GL_TRY(small_set_attrs(ctx));
    const int n = host[0].n;
    const size_t lds = small_lds_bytes(n, host[0].E);
    const SmallBatch Q{devs, diag, status, bar, B, B + (getenv("GRAPHLEARN_AMD_RES_TEST_NOT_RESIDENT") ? 1 : 0)};
    if (n <= 64)
        k_small<256, true><<<B, 256, lds, ctx->stream>>>(host[0], Q);
    else
        k_small<1024, true><<<B, 1024, lds, ctx->stream>>>(host[0], Q);
    GL_HIP(ctx, hipGetLastError());
    long long *st = (long long *)((double *)ctx->pinned + 64);
    GL_HIP(ctx, hipMemcpyAsync(st, status, 2 * 8, hipMemcpyDeviceToHost, ctx->stream));
    for (int a = 0; a < B; ++a)  // the done / final-iteration words belong to the per-graph rule: re-arm
        GL_HIP(ctx, hipMemsetAsync(sessions[a]->dev.ctr, 0, 2 * 8, ctx->stream));
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int a = 0; a < B; ++a) sessions[a]->it += st[0];
    *ran_out = st[0];
    *need_exact_out = st[1] == 1;
    if (st[1] == 2) {
        // the workgroups did not all meet (GPU shared with another process).  At the start-up rendezvous
        // nothing has been modified and the caller falls back to per-iteration launches; later it is an error.
        if (st[0] != 0)
            return gl_set_error(ctx, GL_E_HIP, "small batched engine: grid barrier timed out in mid-run");
        *need_exact_out = false;
        return GL_E_STATE;  // not an error message: "not co-resident", handled by gl_logdeg_batch_iterate
    }
    return GL_OK;
}

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
// Batched form (the 2-D branch, :128-129,158-159): one workgroup per graph; after every iteration the
// four squared norms of every graph go round as tagged records (one memory round trip, no grid barrier:
// see the exchange in the kernel).  They are the diagonals of the Gram
// matrices whose largest eigenvalues decide the coupled stopping rule, and lambda_max lies in
// [max diagonal, trace]: while those bounds keep the ratios above tol the loop goes on; when they
// straddle tol (or allow a stop) the workgroups tighten them in place (B <= 32 and LDS permitting): every graph publishes
// its update and its previous iterate, workgroup b forms row b of the four B x B Gram matrices, and four
// waves bracket lambda_max of one matrix each by repeated squaring (A <- A^2 / tr A^2: the bracket
// [tr A_m^2, sqrt(tr A_m^2)] of the m-th power contracts by a square root per level on the way back, to
// 1e-9 after 30 levels whatever the spectrum) -- every workgroup from the same numbers, so they agree.
// Only a genuine tie (|rel / tol - 1| < 1e-9), a degenerate norm or B > 32 still goes to the host
// (gl_logdeg_batch_iterate: full Gram matrices, Jacobi eigenvalues), which then re-enters.
//
// The kernels read and write the session's tiled layout directly (edge (i, j) -> tile offset), so every
// other entry point sees a consistent state afterwards.
#include <algorithm>
#include <vector>

#include "gl_bracket.h"
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

constexpr int SMALL_GRAM_MAXB = 32;  // the in-kernel brackets of lambda_max: B^2 entries over one wave (<= 16 per lane)

struct SmallBatch {
    const SmallDev *devs;  // one per graph (device array)
    unsigned long long *drec;  // [2][B][4][2]: the four squared norms of every graph as tagged records, by iteration parity
    long long *status;     // [0] iterations run by this launch, [1] 0 = maxit reached, 1 = the host must decide,
                           //     2 = barrier lost, 3 = the coupled rule is met (decided on the device)
    unsigned *bar;         // grid barrier words (gl_gridbar.h)
    int B;
    int rendezvous;        // workgroups expected at the start-up rendezvous (= B; B + 1 in the fallback test)
    double *vec;           // [B][vstride]: w_new - w_old (E), w_old (E), d_new - d_old (n), d_old (n) of every graph
    double *gram;          // [4][B][B]
    int vstride;
    int use_gram;          // B <= SMALL_GRAM_MAXB
};

__device__ __forceinline__ double sm_ld(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ double sm_wave_max(double v) { return wave_max(v); }  // gl_lanes.h (DPP)
__device__ __forceinline__ void sm_st(double *p, double v) {
    __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The in-kernel tightening of the coupled rule's brackets (see the header).  Kept out of line and fed from LDS
// only (w_new in w_l, w_old in wo_l, d in d_l, the previous d in dp_l): the iteration loop around it keeps
// its register allocation.  All threads of every workgroup call it in the same iteration.
// Returns reason (0 go on, 1 host decides, 2 barrier lost, 3 rule met) | barrier epoch << 32.
template <int ST>
__device__ __noinline__ unsigned long long sm_tighten(SmallBatch Q, GlBarState bs, const double *w_l, const double *wo_l,
                                                      const double *d_l, const double *dp_l, double *gred, double *gA,
                                                      int *s_ok, int n, int E, double tol) {
    constexpr int NW = ST / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = Q.B;
    const GlGridBar gb{Q.bar, Q.rendezvous};
    int reason = 0;
    // (a) everybody publishes update and previous iterate
    double *mv = Q.vec + (size_t)blockIdx.x * Q.vstride;
    for (int e = tid; e < E; e += ST) {
        sm_st(mv + e, w_l[e] - wo_l[e]);
        sm_st(mv + E + e, wo_l[e]);
    }
    if (tid < n) {
        sm_st(mv + 2 * E + tid, d_l[tid] - dp_l[tid]);
        sm_st(mv + 2 * E + n + tid, dp_l[tid]);
    }
    if (!gl_grid_barrier(gb, bs, s_ok)) return 2ull | ((unsigned long long)bs.epoch << 32);
    // (b) my row of the four Gram matrices, one column at a time, fixed summation order
    for (int cg = 0; cg < B; ++cg) {
        const double *ov = Q.vec + (size_t)cg * Q.vstride;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (int e = tid; e < E; e += ST) {
            a0 += (w_l[e] - wo_l[e]) * sm_ld(ov + e);
            a1 += wo_l[e] * sm_ld(ov + E + e);
        }
        if (tid < n) {
            a2 = (d_l[tid] - dp_l[tid]) * sm_ld(ov + 2 * E + tid);
            a3 = dp_l[tid] * sm_ld(ov + 2 * E + n + tid);
        }
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
        a2 = wave_sum(a2);
        a3 = wave_sum(a3);
        if (lane == 0) {
            gred[0 * NW + wave] = a0;
            gred[1 * NW + wave] = a1;
            gred[2 * NW + wave] = a2;
            gred[3 * NW + wave] = a3;
        }
        __syncthreads();
        if (tid < 4) {
            double t = 0.0;
            for (int v = 0; v < NW; ++v) t += gred[tid * NW + v];
            sm_st(Q.gram + ((size_t)tid * B + blockIdx.x) * B + cg, t);
        }
        __syncthreads();
    }
    if (!gl_grid_barrier(gb, bs, s_ok)) return 2ull | ((unsigned long long)bs.epoch << 32);
    // (c) brackets of the four largest eigenvalues: wave q takes matrix q
    for (int idx = tid; idx < 4 * B * B; idx += ST) gA[idx] = sm_ld(Q.gram + idx);
    __syncthreads();
    if (wave < 4) {
        double lo, hi;
        if (B <= 16)
            sm_lambda_bracket<4>(gA + wave * B * B, B, lane, lo, hi);
        else
            sm_lambda_bracket<16>(gA + wave * B * B, B, lane, lo, hi);
        if (lane == 0) {
            gA[4 * B * B + 2 * wave] = lo;
            gA[4 * B * B + 2 * wave + 1] = hi;
        }
    }
    __syncthreads();
    const double *br = gA + 4 * B * B;  // {lo, hi} x {update, iterate, dual update, dual}
    const double tp = tol * tol * (1 + 1e-9), tm = tol * tol * (1 - 1e-9);
    const bool not_yet = br[0] > tp * br[3] || br[4] > tp * br[7];
    const bool met = br[2] > 0 && br[6] > 0 && br[1] < tm * br[2] && br[5] < tm * br[6];
    __syncthreads();  // gA is rewritten at the next tightening
    if (met)
        reason = 3;  // :166-167 holds for every admissible value of the four norms
    else if (!not_yet)
        reason = 1;  // a tie to 1e-9: the host's exact evaluation decides
    return (unsigned long long)reason | ((unsigned long long)bs.epoch << 32);
}

__device__ __forceinline__ int64_t tiled_offset(int i, int j, int nb) {  // i < j
    const int I = i >> 6, J = j >> 6;
    const int64_t t = (int64_t)I * nb - (int64_t)I * (I - 1) / 2 + (J - I);
    return t * GL_TILE_ELEMS + (int64_t)(i & 63) * GL_TB + (j & 63);
}

// T threads: 1024 (n <= 128, 8 threads per node in the node phase) or 256 (n <= 64, 4 per node: fewer
// waves to synchronise and to reduce over when the graph is tiny).  BATCH: see the header comment.
// TIGHT compiles the in-kernel tightening in.  At 1024 threads it has a price: 8 edges x (z, w, w_old, end
// points) plus the batch state already fill the 128 registers per thread, and what lives across the call
// is spilled inside the loop as well (bare iteration 14 -> 24 us at n = 128); the host therefore picks the
// variant by batch size (from B = 12 on the host's Gram + Jacobi evaluation costs more than that).
template <int T, bool BATCH, bool TIGHT = false>
__global__ __launch_bounds__(T) void k_small(SmallDev P1, SmallBatch Q) {
    constexpr int ST = T, NW = T / 64, PARTS = T / 128 >= 8 ? 8 : 4;
    constexpr int EPT = T == 512 ? 2 * SMALL_EPT : SMALL_EPT;  // edges per thread: 8128 / 512 in the batched form at n > 64
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
    double *gred = red + 64;        // [4][NW] partials of a column of Gram entries (batched form)
    double *gA = gred + 256;        // [4][B][B] Gram matrices, then [8] their brackets (batched form)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double c = P.c, g = P.gamma, g2 = P.g2;

    // my edges: e = tid + ST k; z, the packed end points, w and the previous w stay in registers
    double z[EPT], wr[EPT], wo[EPT];
    int ij[EPT];
    const double *w_in = P.w[P.it0 & 1];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
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
        for (int k = 0; k < EPT; ++k) {
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
        {  // the four sums of the wave in one halving butterfly (7 exchanges instead of 24): lane 8 q holds sum q
            const double four[4] = {nqy, nw, ndq, nd};
            const double c = add_xor32(halfwave_rows4(four, lane));
            if (lane < 32 && (lane & 7) == 0) red[(lane >> 3) * NW + wave] = c;
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
            // coupled rule: every graph publishes its four squared norms as self-validating records {low word |
            // tag}, {high word | tag}, tag = iteration + 1 (zeroed by the host before the launch), and wave q
            // re-reads quantity q of all graphs until every tag matches: one memory round trip, no grid barrier
            // ("the data is the flag", as in the LGMRF ADMM kernel).  Two record sets by iteration parity: a
            // graph publishes iteration k + 1 only after it has read everybody's iteration k.
            if (tid < 4) {
                const double v = tid == 0 ? tot[0] : (tid == 1 ? tot[1] : (tid == 2 ? tot[2] : tot[3]));
                const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
                const unsigned long long tg = (unsigned long long)(unsigned)(iter + 1) << 32;
                unsigned long long *r = Q.drec + (((size_t)(iter & 1) * Q.B + blockIdx.x) * 4 + tid) * 2;
                __hip_atomic_store(r, (bits & 0xffffffffull) | tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(r + 1, (bits >> 32) | tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (wave < 4) {
                double m = 0.0, t = 0.0;
                const unsigned want = (unsigned)(iter + 1);
                const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
                bool good = true;
                for (int b0 = 0; good && b0 < Q.B; b0 += 64) {
                    const int b = b0 + lane;
                    const unsigned long long *r = Q.drec + (((size_t)(iter & 1) * Q.B + (b < Q.B ? b : 0)) * 4 + wave) * 2;
                    unsigned long long lo = 0, hi = 0;
                    for (unsigned spins = 0;; ++spins) {
                        lo = __hip_atomic_load(r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        hi = __hip_atomic_load(r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const bool ok = b >= Q.B || ((unsigned)(lo >> 32) == want && (unsigned)(hi >> 32) == want);
                        if (__all(ok)) break;
                        __builtin_amdgcn_s_sleep(2);
                        if ((spins & 63u) == 63u && __builtin_amdgcn_s_memrealtime() - t_start > 100000000ull) {
                            good = false;  // 1 s: a graph's workgroup is gone
                            break;
                        }
                    }
                    if (good && b < Q.B) {
                        const double v = __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffull)));
                        m = fmax(m, v);
                        t += v;
                    }
                }
                m = sm_wave_max(m);
                t = wave_sum(t);
                if (lane == 0) {
                    gred[wave] = m;
                    gred[4 + wave] = t;
                    if (!good) s_ok = 0;
                }
            }
            __syncthreads();
            if (!s_ok) {
                reason = 2;  // hand-off lost: the host reports it
                break;
            }
            if (P.it0 + iter >= 1) {
                double mx[4], tr[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    mx[q] = gred[q];
                    tr[q] = gred[4 + q];
                }
                // rel_p^2 in [mx0 / tr1, tr0 / mx1], rel_d^2 in [mx2 / tr3, tr2 / mx3]
                const double t2 = P.tol * P.tol * (1 + 1e-9);
                const bool degenerate = !(mx[1] > 0) || !(mx[3] > 0);
                const bool surely_not = !degenerate && (mx[0] > t2 * tr[1] || mx[2] > t2 * tr[3]);
                if (!surely_not) {
                    if (!TIGHT || !Q.use_gram || degenerate) {
                        reason = 1;  // the exact spectral norms are needed: the host decides
                        break;
                    }
                    if (TIGHT) {
                        // ---- tighten the brackets in place: the previous iterate joins the new one in LDS (the
                        //      prox point is not needed any more in this iteration)
#pragma unroll
                        for (int k = 0; k < EPT; ++k)
                            if (ij[k] >= 0) p_l[tid + ST * k] = wo[k];
                        __syncthreads();
                        const unsigned long long r = sm_tighten<ST>(Q, bs, w_l, p_l, d_l, dp_l, gred, gA, &s_ok, n, E, P.tol);
                        bs.epoch = (unsigned)(r >> 32);
                        if ((int)(r & 0xff) != 0) {
                            reason = (int)(r & 0xff);
                            break;
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- state back to the session's arrays: the iterate into the buffer of its parity, the previous one
    //      into the other (the batched rule's exact evaluation and the streaming engine read both)
    {
        const long long itf = P.it0 + ran;
        double *w_cur = P.w[itf & 1], *w_prev = P.w[(itf + 1) & 1];
#pragma unroll
        for (int k = 0; k < EPT; ++k)
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

// gram_b: batch size whose four Gram matrices the kernel holds (0: none)
size_t small_lds_bytes(int n, int E, int gram_b = 0) { return ((size_t)2 * E + 4 * n + 64 + 256 + 4 * (size_t)gram_b * gram_b + 8) * 8; }
constexpr size_t SMALL_LDS_MAX = 158 * 1024;

int small_set_attrs(gl_ctx *ctx) {  // per context: one process may drive several GPUs
    if (ctx->small_attrs_set) return GL_OK;
    const int most = (int)SMALL_LDS_MAX;
    GL_HIP(ctx, hipFuncSetAttribute((const void *)k_small<1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize, most));
    GL_HIP(ctx, hipFuncSetAttribute((const void *)k_small<1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, most));
    GL_HIP(ctx, hipFuncSetAttribute((const void *)k_small<1024, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, most));
    GL_HIP(ctx, hipFuncSetAttribute((const void *)k_small<256, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, most));
    ctx->small_attrs_set = true;
    return GL_OK;
}

}  // namespace

bool gl_small_applicable(const gl_logdeg *s) {
    if (s->n > SMALL_MAX_N) return false;
    if (const char *e = getenv("GRAPHLEARN_AMD_ENGINE"))
        if (!strcmp(e, "stream") || !strcmp(e, "resident") || !strcmp(e, "resident1") || !strcmp(e, "tile"))
            return false;  // another engine is asked for
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
static size_t small_vstride(int n) { return ((size_t)n * (n - 1) + 2 * (size_t)n + 15) / 16 * 16; }  // 2 E + 2 n, padded
// Device scratch layout: SmallDev[B] | status[2] | bar[GL_BAR_WORDS] | drec[2][B][4][2] | pad | vec[B][vstride] |
// gram[4][B][B] (B <= 32).  status .. drec are zeroed before every launch.
// stopped: the coupled rule was met and decided on the device (no exact evaluation needed for the decision).
size_t gl_small_batch_scratch_bytes(int B, int n) {
    return (size_t)B * sizeof(SmallDev) + 2 * 8 + GL_BAR_WORDS * 4 + (size_t)B * 128 + 192 +
           (B <= SMALL_GRAM_MAXB ? ((size_t)B * small_vstride(n) + (size_t)4 * B * B) * 8 : 0);
}

int gl_small_batch_run(gl_logdeg **sessions, int B, int64_t iters, double tol, void *dev_scratch, int64_t *ran_out,
                       bool *need_exact_out, bool *stopped_out) {
    gl_ctx *ctx = sessions[0]->ctx;
    char *base = (char *)dev_scratch;
    SmallDev *devs = (SmallDev *)base;
    long long *status = (long long *)(base + (size_t)B * sizeof(SmallDev));
    unsigned *bar = (unsigned *)(status + 2);
    unsigned long long *drec = (unsigned long long *)(bar + GL_BAR_WORDS);
    const size_t zero_bytes = 2 * 8 + GL_BAR_WORDS * 4 + (size_t)B * 128;
    std::vector<SmallDev> host((size_t)B);
    for (int a = 0; a < B; ++a) host[(size_t)a] = small_dev_of(sessions[a], tol, iters);
    GL_HIP(ctx, hipMemcpyAsync(devs, host.data(), host.size() * sizeof(SmallDev), hipMemcpyHostToDevice, ctx->stream));
    GL_HIP(ctx, hipMemsetAsync(status, 0, zero_bytes, ctx->stream));
    GL_TRY(small_set_attrs(ctx));
    const int n = host[0].n;
    // the env: tests / comparison
    const bool use_gram = B <= SMALL_GRAM_MAXB && small_lds_bytes(n, host[0].E, B) <= SMALL_LDS_MAX && (n <= 64 || B >= 12) &&
                          !getenv("GRAPHLEARN_AMD_BATCH_HOST_RULE");
    const size_t lds = small_lds_bytes(n, host[0].E, use_gram ? B : 0);
    double *vec = (double *)(((uintptr_t)(drec + (size_t)B * 16) + 63) & ~(uintptr_t)63);
    const int vstride = (int)small_vstride(n);
    double *gram = vec + (B <= SMALL_GRAM_MAXB ? (size_t)B * vstride : 0);
    const SmallBatch Q{devs, drec, status, bar, B, B + (getenv("GRAPHLEARN_AMD_RES_TEST_NOT_RESIDENT") ? 1 : 0),
                       vec, gram, vstride, use_gram ? 1 : 0};
    if (n <= 64)
        k_small<256, true, true><<<B, 256, lds, ctx->stream>>>(host[0], Q);
    else if (use_gram)
        k_small<1024, true, true><<<B, 1024, lds, ctx->stream>>>(host[0], Q);
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
    *stopped_out = st[1] == 3;
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

// Log-degree graph learning (Kalofolias) -- the primal-dual loop on MI355X.
//
// Reference path: graph_learn/smooth_learning.py:47-174 (gsp_learn_graph_log_degrees).
// One iteration of the reference's forward-backward-forward loop (:146-167) is
//   edge pass  k_edge  : y, p, q and the new edge iterate, per edge, plus the two
//                        degree sums S p and S w_new and the primal norm partials;
//   node pass  k_node  : qbar, the new dual variable, ybar/pbar of the NEXT
//                        iteration, the dual norm partials and the stopping rule.
// The S / S^T operators of the reference (sum_squareform, :13-44) are never
// built: (S^T d)_(i,j) = d_i + d_j and (S w)_i = sum_j W_ij are evaluated from
// the tile coordinates.  Arithmetic is fp64 with contraction disabled so that the
// per-edge results equal NumPy's bit for bit; only the order of the degree sums
// and norm sums differs (fixed, deterministic order: no atomics anywhere).
//
// Work decomposition of the edge pass (HBM/Infinity-Cache bound, 24 B per edge):
// the tiled lattice is a flat list of "row pairs" (2 rows x 64 columns = 1 KiB, one
// wave-wide 16-B load); a persistent grid of G = 8 x #CU single-wave workgroups
// takes equal contiguous ranges of it, so every CU streams the same number of
// bytes and the launch has no tail.
#include "gl_bracket.h"
#include "gl_lanes.h"
#include "gl_logdeg.h"

#include <algorithm>
#include <chrono>
#include <limits>
#include <type_traits>
#include <vector>

namespace {

constexpr int TB = GL_TB;            // 64 x 64 tiles
#ifndef GL_ROWS_PER_LANE
#define GL_ROWS_PER_LANE 8
#endif
#ifndef GL_EDGE_WAVES_PER_SIMD
#define GL_EDGE_WAVES_PER_SIMD 2
#endif
constexpr int ROWS_PER_LANE = GL_ROWS_PER_LANE;  // lane: 2 columns x this many row pairs per step
constexpr int PAIRS_PER_TILE = TB / 2;
constexpr int LAYOUT_ROWS = 8;        // layout kernels: lane = 2 columns x 8 rows (256 threads per tile)
constexpr int ITER_BATCH = 64;       // iterations per captured graph / per convergence poll

template <int R>
struct RowReduce;
template <>
struct RowReduce<8> {
    static __device__ __forceinline__ double run(const double (&v)[8], int lane) { return halfwave_rows8(v, lane); }
    static __device__ __forceinline__ int row_of(int lane) { return (lane >> 2) & 7; }
    static __device__ __forceinline__ bool writer(int lane) { return (lane & 3) == 0; }
};
template <>
struct RowReduce<4> {
    static __device__ __forceinline__ double run(const double (&v)[4], int lane) { return halfwave_rows4(v, lane); }
    static __device__ __forceinline__ int row_of(int lane) { return (lane >> 3) & 3; }
    static __device__ __forceinline__ bool writer(int lane) { return (lane & 7) == 0; }
};


// ---------------------------------------------------------------------------
// Edge pass.  Every WAVE is an autonomous worker (workgroup = one wave, no LDS, no
// barrier): it owns a contiguous range of the flat row-pair list and walks it in
// steps of <= 8 row pairs inside one tile.  Loads of the next step are issued
// slot by slot right after the slot's registers are consumed, so the next step's
// data is already on its way while the row / column reductions of this step run.
// Row sums are complete inside the wave (half-wave butterfly); column partials stay
// in registers until the wave leaves the tile, then go to its colslab entry.
// INIT = true computes only the degree sums of the current iterate (dvar = S w, :144).
// ABL: development-only ablation mask (tools/edge_ablate.hip); the product uses ABL = 0.
//   1 no node-vector gathers, 2 no row sums, 4 no column sums, 8 no w store, 16 no norms, 32 no z load
// development (tools/build_variant.sh): non-temporal hints on the streams of the edge pass -- 1: z, 2: w loads, 4: w stores
#ifndef GL_EDGE_NT
#define GL_EDGE_NT 0
#endif
template <bool NT>
__device__ __forceinline__ double2 edge_ld(const double *p) {
    if (NT) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(p));
        return double2{v.x, v.y};
    }
    return *reinterpret_cast<const double2 *>(p);
}
template <bool NT>
__device__ __forceinline__ void edge_st(double *p, double2 v) {
    if (NT) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store((v2d){v.x, v.y}, reinterpret_cast<v2d *>(p));
    } else {
        *reinterpret_cast<double2 *>(p) = v;
    }
}
struct EdgeStep {
    int64_t t;   // tile
    int a;       // first row pair of the step inside the tile
    int cnt;     // row pairs in the step (0 = none)
};

__device__ __forceinline__ EdgeStep edge_step_at(int64_t p, int64_t pend) {
    EdgeStep s;
    s.t = p >> 5;
    s.a = (int)(p & (PAIRS_PER_TILE - 1));
    const int64_t left = pend - p;
    int cnt = PAIRS_PER_TILE - s.a;
    cnt = cnt < ROWS_PER_LANE ? cnt : ROWS_PER_LANE;
    s.cnt = left <= 0 ? 0 : (left < cnt ? (int)left : cnt);
    return s;
}

// dvar / pbar come in as separate noalias arguments: a slot's two row entries have wave-uniform
// addresses and are fetched with SCALAR loads (s_load, counted by lgkmcnt).  Vector loads for
// them would sit in the in-order vector-memory queue behind the prefetch just issued by the
// previous slot, and waiting for them would cost a full memory round trip per slot.
// par >= 0 (the batched launches): the parity of the iteration index comes with the launch, so the first loads do not wait
// for a round trip to the counter; the done flag is looked at once they are on their way, before anything is stored.
template <bool INIT, int ABL = 0>
__device__ __forceinline__ void edge_body(const LogdegDev &P, const double *__restrict__ dvar,
                                          const double *__restrict__ pbar, const int b, const int par = -1) {
    if (!INIT && par < 0 && P.ctr[0] != 0) return;  // converged earlier: leave everything as is
    const long long it = (INIT || par >= 0) ? 0 : P.ctr[2];

    const int lane = threadIdx.x;
    const int half = lane >> 5;
    const int c0 = 2 * (lane & 31);
    if (!INIT && par < 0 && b == 0 && lane == 0) P.ctr[3] = it;  // hand the index to the node pass

    const int pin = par >= 0 ? par : (int)(it & 1);
    const double *__restrict__ wsrc = P.w[pin];
    double *__restrict__ wdst = P.w[pin ^ 1];
    const double c = P.c, g = P.gamma, g2 = P.g2;

    int64_t p = ((int64_t)b * P.npairs) / P.G;
    const int64_t pend = ((int64_t)(b + 1) * P.npairs) / P.G;
    const int lofs = half * TB + c0;  // lane offset inside a row pair

    double2 wv[ROWS_PER_LANE], zv[ROWS_PER_LANE];
    EdgeStep cur = edge_step_at(p, pend);
#pragma unroll
    for (int k = 0; k < ROWS_PER_LANE; ++k) {
        wv[k] = double2{0.0, 0.0};
        zv[k] = double2{0.0, 0.0};
        if (k < cur.cnt) {
            const int64_t off = cur.t * GL_TILE_ELEMS + (int64_t)(cur.a + k) * (2 * TB) + lofs;
            wv[k] = edge_ld<(GL_EDGE_NT & 2) != 0>(wsrc + off);
            if (!INIT && !(ABL & 32)) zv[k] = edge_ld<(GL_EDGE_NT & 1) != 0>(P.z + off);
        }
    }

    int seg = 0, seg_steps = 0;
    int I = 0, J = 0;
    bool full = false;
    double dj[2] = {0, 0}, pj[2] = {0, 0};
    double colp[2] = {0, 0}, colw[2] = {0, 0};
    double nqy = 0, nw = 0;

    // One step (<= 8 row pairs of one tile).  FULL = the tile has no masked entries: the
    // common case gets its own branch-free arithmetic; diagonal / ragged tiles take the
    // masked variant.  Everything per-step (base pointers, node rows) is hoisted so that a
    // slot costs its arithmetic plus immediate-offset loads and stores.
    auto step = [&](const EdgeStep &cs, const EdgeStep &ns, auto full_tag, double (&rowp)[ROWS_PER_LANE],
                    double (&roww)[ROWS_PER_LANE]) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int64_t sbase = cs.t * GL_TILE_ELEMS + (int64_t)cs.a * (2 * TB) + lofs;
        const int64_t nbase = ns.t * GL_TILE_ELEMS + (int64_t)ns.a * (2 * TB) + lofs;
        double *__restrict__ wq = wdst + sbase;
        const double *__restrict__ wp = wsrc + nbase;
        const double *__restrict__ zp = P.z + nbase;
        const int iu0 = __builtin_amdgcn_readfirstlane(I * TB + 2 * cs.a);  // wave-uniform: rows iu0 + 2k (+1)
        const int irow0 = iu0 + half;                                        // node of slot 0 (this lane's half)
        const double *__restrict__ dr = dvar + iu0;
        const double *__restrict__ pr_ = pbar + iu0;
        const int jcol0 = J * TB + c0;
#pragma unroll
        for (int k = 0; k < ROWS_PER_LANE; ++k) {
            rowp[k] = 0.0;
            roww[k] = 0.0;
            if (k < cs.cnt) {
                double di = 0, pi = 0;
                if (!INIT && !(ABL & 1)) {
                    const double d0 = dr[2 * k], d1 = dr[2 * k + 1];
                    const double q0 = pr_[2 * k], q1 = pr_[2 * k + 1];
                    di = half ? d1 : d0;
                    pi = half ? q1 : q0;
                }
                double wn[2], pp[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const double w = e ? wv[k].y : wv[k].x;
                    bool valid = true;
                    if (!FULL) {
                        const int i = irow0 + 2 * k, j = jcol0 + e;
                        valid = i < j && j < P.n;
                    }
                    if (INIT) {
                        pp[e] = 0.0;
                        wn[e] = (FULL || valid) ? w : 0.0;
                    } else {
                        const double z = e ? zv[k].y : zv[k].x;
                        double y = c * w - g * (di + dj[e]);    // :147
                        const double tt = y - g2 * z;                 // :150
                        double pr = tt > 0 ? tt : 0.0;
                        double q = c * pr - g * (pi + pj[e]);   // :153
                        double wnew = (w - y) + q;              // :163
                        if (ABL & 128) {  // diagnostic: no arithmetic at all, same memory traffic
                            wnew = w + z;
                            pr = z;
                            q = w;
                            y = z;
                        }
                        const double dq = q - y;                      // == -y + q (:158)
                        if (FULL) {
                            pp[e] = pr;
                            wn[e] = wnew;
                            if (!(ABL & 16)) {
                                nqy += dq * dq;
                                nw += w * w;
                            }
                        } else {  // entries outside i < j < n stay exactly 0 and count for nothing
                            pp[e] = valid ? pr : 0.0;
                            wn[e] = valid ? wnew : 0.0;
                            if (!(ABL & 16)) {
                                nqy += valid ? dq * dq : 0.0;
                                nw += valid ? w * w : 0.0;
                            }
                        }
                    }
                    colp[e] += pp[e];
                    colw[e] += wn[e];
                }
                rowp[k] = pp[0] + pp[1];
                roww[k] = wn[0] + wn[1];
                if (!INIT && !(ABL & 8)) edge_st<(GL_EDGE_NT & 4) != 0>(wq + k * (2 * TB), double2{wn[0], wn[1]});
                if (ABL & 8) nqy += wn[0] + wn[1];
            }
            // prefetch: this slot's registers are free again
            if (k < ns.cnt) {
                wv[k] = edge_ld<(GL_EDGE_NT & 2) != 0>(wp + k * (2 * TB));
                if (!INIT && !(ABL & 32)) zv[k] = edge_ld<(GL_EDGE_NT & 1) != 0>(zp + k * (2 * TB));
            }
        }
    };

    // column entries (dvar / pbar of this lane's two columns) of the current tile; those of the
    // next tile are requested one step ahead, BEFORE that step's ring refills, so that they are
    // older in the in-order vector-memory queue than anything the next tile's first step waits for
    double djn[2] = {0, 0}, pjn[2] = {0, 0};
    int In = 0, Jn = 0;
    auto fetch_tile = [&](int64_t t, int &Io, int &Jo, double (&dd)[2], double (&pp)[2]) {
        gl_tile_decode(t, P.nb, Io, Jo);
        Io = __builtin_amdgcn_readfirstlane(Io);
        Jo = __builtin_amdgcn_readfirstlane(Jo);
        if (!INIT && !(ABL & 1)) {
            const int64_t j0 = (int64_t)Jo * TB + c0;
            dd[0] = dvar[j0];
            dd[1] = dvar[j0 + 1];
            pp[0] = pbar[j0];
            pp[1] = pbar[j0 + 1];
        }
    };
    if (cur.cnt > 0) {
        fetch_tile(cur.t, I, J, dj, pj);
        full = (I < J) && ((int64_t)J * TB + TB <= P.n);
    }
    if (!INIT && par >= 0) {
        if (P.ctr[0] != 0) return;                  // (a halted batch: nothing has been stored yet)
        if (b == 0 && lane == 0) P.ctr[3] = P.ctr[2];  // the index for the node pass (which moves ctr[2] itself)
    }

    while (cur.cnt > 0) {
        const EdgeStep nxt = edge_step_at(p + cur.cnt, pend);
        const bool tile_change = nxt.cnt > 0 && nxt.t != cur.t;
        if (tile_change) fetch_tile(nxt.t, In, Jn, djn, pjn);
        const int irow0 = I * TB + 2 * cur.a + half;  // node of slot 0

        unsigned long long st0 = 0, st1 = 0, st2 = 0;
        if (ABL & 64) {  // diagnostic build only: wait for this step's data, then stamp
            st0 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0);
            st1 = __builtin_amdgcn_s_memtime();
        }
        double rowp[ROWS_PER_LANE], roww[ROWS_PER_LANE];
        if (full)
            step(cur, nxt, std::true_type{}, rowp, roww);
        else
            step(cur, nxt, std::false_type{}, rowp, roww);
        if (ABL & 64) st2 = __builtin_amdgcn_s_memtime();

        // ---- row degree partials of this step: complete inside the wave
        if (ABL & 2) {
#pragma unroll
            for (int k = 0; k < ROWS_PER_LANE; ++k) nw += rowp[k] + roww[k];
        } else {
            const double rw = RowReduce<ROWS_PER_LANE>::run(roww, lane);
            double rp_ = 0;
            if (!INIT) rp_ = RowReduce<ROWS_PER_LANE>::run(rowp, lane);
            const int k = RowReduce<ROWS_PER_LANE>::row_of(lane);
            if (RowReduce<ROWS_PER_LANE>::writer(lane) && k < cur.cnt) {
                const int64_t idx = (int64_t)J * P.n_pad + irow0 + 2 * k;
                if (!INIT) P.rowslab_p[idx] = rp_;
                P.rowslab_w[idx] = rw;
            }
        }

        // ---- leaving the tile (or the range): flush the column partials
        if (nxt.cnt == 0 || nxt.t != cur.t) {
            if (ABL & 4) {
                nw += (colp[0] + colp[1]) + (colw[0] + colw[1]);
            } else {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    colp[e] = add_xor32(colp[e]);
                    colw[e] = add_xor32(colw[e]);
                }
                if (half == 0) {
                    const int64_t ent = P.seg_dest[(int64_t)b * P.kmax + seg];
                    double *dst = P.colslab + ent * (2 * TB) + c0;
                    if (!INIT) *reinterpret_cast<double2 *>(dst) = double2{colp[0], colp[1]};
                    *reinterpret_cast<double2 *>(dst + TB) = double2{colw[0], colw[1]};
                }
            }
            colp[0] = colp[1] = colw[0] = colw[1] = 0.0;
            ++seg;
        }
        if ((ABL & 64) && P.dbg && lane == 0 && seg_steps < 8) {
            const unsigned long long st3 = __builtin_amdgcn_s_memtime();
            unsigned long long *o = P.dbg + ((size_t)b * 8 + seg_steps) * 4;
            o[0] = st0;
            o[1] = st1;
            o[2] = st2;
            o[3] = st3;
        }
        ++seg_steps;
        p += cur.cnt;
        cur = nxt;
        if (tile_change) {
            I = In;
            J = Jn;
            dj[0] = djn[0];
            dj[1] = djn[1];
            pj[0] = pjn[0];
            pj[1] = pjn[1];
            full = (I < J) && ((int64_t)J * TB + TB <= P.n);
        }
    }

    if (!INIT) {
        nqy = wave_sum(nqy);
        nw = wave_sum(nw);
        if (lane == 0) {
            P.tnorm[2 * b + 0] = nqy;
            P.tnorm[2 * b + 1] = nw;
        }
    }
}

template <bool INIT, int ABL = 0>
__global__ __launch_bounds__(64, GL_EDGE_WAVES_PER_SIMD) void k_edge(LogdegDev P, const double *__restrict__ dvar,
                                                                     const double *__restrict__ pbar) {
    edge_body<INIT, ABL>(P, dvar, pbar, (int)blockIdx.x);
}
// the same pass for a batch of problems of one size in ONE launch (blockIdx.y = problem): small
// graphs then run side by side instead of one latency-bound launch after the other
__global__ __launch_bounds__(64, GL_EDGE_WAVES_PER_SIMD) void k_edge_batch(const LogdegDev *__restrict__ devs, const int par) {
    const LogdegDev P = devs[blockIdx.y];
    edge_body<false, 0>(P, P.dvar, P.pbar, (int)blockIdx.x, par);
}

// ---------------------------------------------------------------------------
// Node pass.  One workgroup (16 waves) per block of 64 nodes: lanes are nodes,
// wave v sums every 16th slab entry, so all slab loads of a lane are in flight
// together; order of every sum is fixed.  It also evaluates the stopping rule
// (:156-161,166) of the PREVIOUS iteration -- deferred by one pass so that no
// extra launch or grid-wide wait sits on the iteration path; the ping-pong edge
// buffers keep the iterate the rule refers to intact.
constexpr int NODE_WAVES = 16;

// par >= 0 (the batched launches): as in edge_body; the single-graph rule is not evaluated (the batch has its own), and
// false comes back when the batch was halted (nothing written).
template <bool INIT>
__device__ __forceinline__ bool node_body(const LogdegDev &P, const int64_t B, const int par = -1) {
    __shared__ double s_sum[2][NODE_WAVES][TB];
    __shared__ double s_red[2][NODE_WAVES];
    __shared__ int s_stop;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long it = 0;
    if (!INIT && par < 0) {
        if (P.ctr[0] != 0) return false;
        it = P.ctr[3];
    }
    const bool chk = !INIT && par < 0 && it >= 2;  // rule of iteration it-1 >= 1 (the reference skips i == 0)
    const int64_t i = B * TB + lane;

    // loads first: stopping-rule partials, my node's state, then the slabs
    double b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    if (chk && wave == 1) {
        const double *bn = P.bnorm + ((it - 1) & 1) * 4 * P.nb;
        for (int64_t bb = lane; bb < P.nb; bb += 64) {
            b0 += bn[4 * bb + 0];
            b1 += bn[4 * bb + 1];
            b2 += bn[4 * bb + 2];
            b3 += bn[4 * bb + 3];
        }
    }
    double d_old = 0, yb_old = 0, pb_old = 0;
    if (!INIT && wave == 0) {
        d_old = P.dvar[i];
        yb_old = P.ybar[i];
        pb_old = P.pbar[i];
    }

    // degree sums of my 64 nodes: row partials of tiles (B, J >= B), column
    // partials of every segment of the tiles (I <= B, B) (contiguous in colslab)
    double sp = 0, sw = 0;
#pragma unroll 4
    for (int64_t J = B + wave; J < P.nb; J += NODE_WAVES) {
        if (!INIT) sp += P.rowslab_p[J * P.n_pad + i];
        sw += P.rowslab_w[J * P.n_pad + i];
    }
    const int64_t e1 = P.col_ptr[B + 1];
#pragma unroll 4
    for (int64_t e = P.col_ptr[B] + wave; e < e1; e += NODE_WAVES) {
        if (!INIT) sp += P.colslab[e * (2 * TB) + lane];
        sw += P.colslab[e * (2 * TB) + TB + lane];
    }
    s_sum[0][wave][lane] = sp;
    s_sum[1][wave][lane] = sw;

    // my share of the per-wave primal norm partials of the edge pass
    if (!INIT) {
        double t0 = 0, t1 = 0;
        for (int64_t gq = B + (int64_t)tid * P.nb; gq < P.G; gq += (int64_t)NODE_WAVES * 64 * P.nb) {
            t0 += P.tnorm[2 * gq + 0];
            t1 += P.tnorm[2 * gq + 1];
        }
        t0 = wave_sum(t0);
        t1 = wave_sum(t1);
        if (lane == 0) {
            s_red[0][wave] = t0;
            s_red[1][wave] = t1;
        }
    }
    if (tid == 0) s_stop = 0;
    if (!INIT && par >= 0) {  // (the loads above are on their way; every thread reads the same two words)
        if (P.ctr[0] != 0) return false;
        it = P.ctr[3];  // (handed over by the edge pass)
    }
    __syncthreads();
    if (chk && wave == 1) {
        b0 = wave_sum(b0);
        b1 = wave_sum(b1);
        b2 = wave_sum(b2);
        b3 = wave_sum(b3);
        const double rp = sqrt(b0) / sqrt(b1);  // ||q - y|| / ||w_prev||        (:158)
        const double rd = sqrt(b2) / sqrt(b3);  // ||qbar - ybar|| / ||d_prev||  (:159)
        const int stop = (rp < P.dstat[2] && rd < P.dstat[2]) ? 1 : 0;
        if (lane == 0) {
            if (stop) s_stop = 1;
            if (B == 0) {
                P.dstat[0] = rp;
                P.dstat[1] = rd;
                if (stop) {
                    P.ctr[1] = it;  // iterations completed
                    P.ctr[0] = 1;
                }
            }
        }
    }
    if (chk) __syncthreads();
    if (s_stop) return false;

    if (wave == 0) {
        sp = 0;
        sw = 0;
#pragma unroll
        for (int v = 0; v < NODE_WAVES; ++v) {
            sp += s_sum[0][v][lane];
            sw += s_sum[1][v][lane];
        }
        const bool real = i < P.n;
        double dnew, dq2 = 0, d2 = 0;
        if (INIT) {
            dnew = sw;  // :144  dvar = S w
            if (B == 0 && lane == 0) {  // no relative change yet, no tolerance yet
                P.dstat[0] = INFINITY;
                P.dstat[1] = INFINITY;
                P.dstat[2] = 0.0;
            }
        } else {
            const double qb = pb_old + P.gamma * sp;  // :154
            const double dq = qb - yb_old;            // == -ybar + qbar (:159)
            dq2 = real ? dq * dq : 0.0;
            d2 = real ? d_old * d_old : 0.0;
            dnew = (d_old - yb_old) + qb;             // :164
        }
        // next iteration's forward + prox on the dual side (:148, :151)
        const double ybn = dnew + P.gamma * sw;
        const double pbn = (ybn - sqrt(ybn * ybn + P.a4g)) / 2;
        if (!INIT) P.dprev[i] = d_old;
        P.dvar[i] = real ? dnew : 0.0;
        P.ybar[i] = real ? ybn : 0.0;
        P.pbar[i] = real ? pbn : 0.0;
        if (!INIT) {
            dq2 = wave_sum(dq2);
            d2 = wave_sum(d2);
            if (lane == 0) {
                double r0 = 0, r1 = 0;
#pragma unroll
                for (int v = 0; v < NODE_WAVES; ++v) {
                    r0 += s_red[0][v];
                    r1 += s_red[1][v];
                }
                // (write-through: the batched node pass hands these to the launch's last workgroup inside the launch)
                double *bn = P.bnorm + (it & 1) * 4 * P.nb + 4 * B;
                __hip_atomic_store(bn + 0, r0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(bn + 1, r1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(bn + 2, dq2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(bn + 3, d2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (B == 0) P.ctr[2] = it + 1;  // the next edge pass runs iteration it + 1
            }
        }
    }
    return true;
}

template <bool INIT>
__global__ __launch_bounds__(NODE_WAVES * 64) void k_node(LogdegDev P) {
    node_body<INIT>(P, (int64_t)blockIdx.x);
}
// State of a batched (2-D) call on the device: the coupled rule's screen runs there, so that the host looks once per
// chunk of iterations instead of once per iteration.
struct BatchState {
    int halt;             // 0 running; 1: the screen cannot rule out that the coupled rule holds (host: exact spectral norms);
                          // 2: the call's iteration budget is used up
    int ran;              // iterations completed in this call
    int first;            // 1 until the call's first iteration is done (the rule is not evaluated for it, :156)
    int remaining;        // iterations this call may still run
    unsigned done_blocks; // workgroups of the current node pass that have finished ("last block" pattern)
    int decide_here;      // 1: an iterate the screen cannot rule out is judged on the device (k_batch_gram
                          // behind the node pass) instead of halting for the host
    int want_brackets;    // ... set by the screen for the kernels behind it, cleared by the last of them
    unsigned gram_blocks; // workgroups of the current k_batch_gram that have finished
    double diag[4];       // scratch of the deciding workgroup: max and trace of the four diagonals are formed from LDS
};
// The node pass of a batch (blockIdx.y = problem).  With `st`: the LAST workgroup of the launch to finish -- whichever it
// is: what it computes does not depend on that -- sums every problem's norm partials of the iteration just completed (the
// four squared norms ||dW_a||^2, ||W_a||^2, ||dD_a||^2, ||D_a||^2 = the diagonals of the four B x B Gram matrices) and
// screens the coupled rule of smooth_learning.py:158-159,166-167: lambda_max of a Gram matrix lies in [max diagonal,
// trace], so rel_p^2 >= max_a ||dW_a||^2 / sum_a ||W_a||^2 (and likewise rel_d): above tol^2 the rule cannot hold and the
// batch goes on; otherwise every problem's done flag is raised, the launches queued behind this one fall through, and the
// host forms the exact spectral norms for exactly this iterate.
__global__ __launch_bounds__(NODE_WAVES * 64) void k_node_batch(const LogdegDev *__restrict__ devs, BatchState *st, int nprob, double t2,
                                                                const int par) {
    __shared__ int s_last;
    __shared__ double s_mx[4][NODE_WAVES], s_tr[4][NODE_WAVES];
    const LogdegDev P = devs[blockIdx.y];
    if (!node_body<false>(P, (int64_t)blockIdx.x, par)) return;  // halted: nothing moves (every problem carries the same flag)
    if (!st) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Hand-off of the norm partials inside the launch (CDNA4 guide, Guideline 16: sc1 stores, the storing lane drains, ONE
    // agent-scope add per workgroup, the workgroup whose add came last reads with sc1 loads behind a workgroup barrier).
    // Thread 0 is the lane that stored this workgroup's partials (node_body).
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned arrived = __hip_atomic_fetch_add(&st->done_blocks, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        s_last = arrived == gridDim.x * gridDim.y ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    double mx[4] = {0, 0, 0, 0}, tr[4] = {0, 0, 0, 0};
    for (int a = wave; a < nprob; a += NODE_WAVES) {
        const LogdegDev Q = devs[a];
        const double *bn = Q.bnorm + par * 4 * Q.nb;  // (the iteration this pass completed has the launch's parity)
        double b[4] = {0, 0, 0, 0};
        for (int64_t bb = lane; bb < Q.nb; bb += 64)
#pragma unroll
            for (int q = 0; q < 4; ++q) b[q] += __hip_atomic_load(bn + 4 * bb + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double v = wave_sum(b[q]);
            mx[q] = fmax(mx[q], v);
            tr[q] += v;
        }
    }
    if (lane == 0)
#pragma unroll
        for (int q = 0; q < 4; ++q) s_mx[q][wave] = mx[q], s_tr[q][wave] = tr[q];
    __syncthreads();
    if (tid == 0) {
        double m[4], t[4];
        for (int q = 0; q < 4; ++q) {
            m[q] = s_mx[q][0], t[q] = s_tr[q][0];
            for (int v = 1; v < NODE_WAVES; ++v) m[q] = fmax(m[q], s_mx[q][v]), t[q] += s_tr[q][v];
        }
        const bool degenerate = !(m[1] > 0) || !(m[3] > 0);
        // rel_p^2 in [m0 / t1, t0 / m1], rel_d^2 in [m2 / t3, t2 / m3]; a margin for rounding
        const bool surely_not = !degenerate && (m[0] / t[1] > t2 * (1 + 1e-9) || m[2] / t[3] > t2 * (1 + 1e-9));
        int halt = 0;
        if (!st->first && !surely_not) {
            if (st->decide_here) st->want_brackets = 1;
            else halt = 1;
        }
        st->first = 0;
        st->ran += 1;
        if (--st->remaining <= 0 && !halt) halt = 2;  // (with a bracket decision pending too: the host looks at this iterate itself)
        if (halt) st->want_brackets = 0;
        st->done_blocks = 0u;
        if (halt) {
            for (int a = 0; a < nprob; ++a) devs[a].ctr[0] = 1;
            st->halt = halt;
        }
    }
}
// after the host has looked: the problems' done flags belong to the batch's screen, not to them
__global__ void k_batch_rearm(const LogdegDev *__restrict__ devs, int nprob, BatchState *st) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a < nprob) devs[a].ctr[0] = 0, devs[a].ctr[1] = 0;
    if (a == 0) st->halt = 0;
}

// After a batch of launches: rel. changes of the last completed iteration, and
// the stopping rule for it (the deferred check has not seen it yet).  One wave.
__global__ void k_status(LogdegDev P) {
    if (P.ctr[0] != 0) return;
    const long long it = P.ctr[2];
    if (it < 2) return;  // iteration 0 never stops (:156-161)
    const int lane = threadIdx.x;
    const double *bn = P.bnorm + ((it - 1) & 1) * 4 * P.nb;
    double b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    for (int64_t bb = lane; bb < P.nb; bb += 64) {
        b0 += bn[4 * bb + 0];
        b1 += bn[4 * bb + 1];
        b2 += bn[4 * bb + 2];
        b3 += bn[4 * bb + 3];
    }
    b0 = wave_sum(b0);
    b1 = wave_sum(b1);
    b2 = wave_sum(b2);
    b3 = wave_sum(b3);
    const double rp = sqrt(b0) / sqrt(b1), rd = sqrt(b2) / sqrt(b3);
    if (lane == 0) {
        P.dstat[0] = rp;
        P.dstat[1] = rd;
        if (rp < P.dstat[2] && rd < P.dstat[2]) {
            P.ctr[1] = it;
            P.ctr[0] = 1;
        }
    }
}

// ---------------------------------------------------------------------------
// Layout conversion kernels (one workgroup per tile, same lane map as k_edge).
template <int MODE>  // 0: from condensed, 1: from square (upper triangle)
__global__ __launch_bounds__(256) void k_to_tiled(const double *__restrict__ src, double *__restrict__ dst,
                                                  int64_t n, int64_t nb, double scale_arg,
                                                  const double *__restrict__ scale_dev = nullptr) {
    const double scale = scale_dev ? scale_dev[0] : scale_arg;  // theta left on the device by set_signals
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int I, J;
    gl_tile_decode((int64_t)blockIdx.x, nb, I, J);
    const int c0 = 2 * (lane & 31);
    const int r0 = wave * 16 + (lane >> 5);
    const int64_t base = (int64_t)blockIdx.x * GL_TILE_ELEMS + (int64_t)r0 * TB + c0;
#pragma unroll
    for (int k = 0; k < LAYOUT_ROWS; ++k) {
        const int64_t i = (int64_t)I * TB + r0 + 2 * k;
        double2 o;
        double v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int64_t j = (int64_t)J * TB + c0 + e;
            v[e] = 0.0;
            if (i < j && j < n) {
                const double s = MODE == 0 ? src[gl_condensed_index(i, j, n)] : src[i * n + j];
                v[e] = s * scale;  // the reference's  sq_pdists * theta  (:290), one rounding
            }
        }
        o.x = v[0];
        o.y = v[1];
        *reinterpret_cast<double2 *>(dst + base + (int64_t)k * 2 * TB) = o;
    }
}

// tiled -> condensed, with the final threshold  w[w < edge_tol] = 0  (:169)
__global__ __launch_bounds__(256) void k_to_condensed(const double *__restrict__ src, double *__restrict__ dst,
                                                      int64_t n, int64_t nb, double edge_tol) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int I, J;
    gl_tile_decode((int64_t)blockIdx.x, nb, I, J);
    const int c0 = 2 * (lane & 31);
    const int r0 = wave * 16 + (lane >> 5);
    const int64_t base = (int64_t)blockIdx.x * GL_TILE_ELEMS + (int64_t)r0 * TB + c0;
#pragma unroll
    for (int k = 0; k < LAYOUT_ROWS; ++k) {
        const int64_t i = (int64_t)I * TB + r0 + 2 * k;
        const double2 v = *reinterpret_cast<const double2 *>(src + base + (int64_t)k * 2 * TB);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int64_t j = (int64_t)J * TB + c0 + e;
            if (i < j && j < n) {
                const double x = e ? v.y : v.x;
                dst[gl_condensed_index(i, j, n)] = (x < edge_tol) ? 0.0 : x;
            }
        }
    }
}

// tiled -> symmetric square with zero diagonal (= squareform(w), :171-172).
// The tile goes through LDS so that both the (i,j) and the mirrored (j,i) rows
// are written as contiguous runs.
__global__ __launch_bounds__(256) void k_to_square(const double *__restrict__ src, double *__restrict__ dst,
                                                   int64_t n, int64_t nb, double edge_tol) {
    __shared__ double s[TB][TB + 1];
    const int tid = threadIdx.x;
    int I, J;
    gl_tile_decode((int64_t)blockIdx.x, nb, I, J);
    const int64_t tbase = (int64_t)blockIdx.x * GL_TILE_ELEMS;
    for (int e = tid; e < GL_TILE_ELEMS; e += 256) {
        const int r = e / TB, c = e % TB;
        const int64_t i = (int64_t)I * TB + r, j = (int64_t)J * TB + c;
        double x = 0.0;
        if (i < j && j < n) {
            x = src[tbase + e];
            x = (x < edge_tol) ? 0.0 : x;
        }
        s[r][c] = x;
    }
    __syncthreads();
    for (int e = tid; e < GL_TILE_ELEMS; e += 256) {
        const int r = e / TB, c = e % TB;
        const int64_t i = (int64_t)I * TB + r, j = (int64_t)J * TB + c;
        if (I == J) {
            // symmetric fill of the diagonal block, zero diagonal
            if (i < n && j < n) dst[i * n + j] = (r < c) ? s[r][c] : (r > c ? s[c][r] : 0.0);
        } else {
            if (i < n && j < n) dst[i * n + j] = s[r][c];
            // mirrored block: row = J*TB + r', col = I*TB + c'  (read transposed)
            const int64_t mi = (int64_t)J * TB + r, mj = (int64_t)I * TB + c;
            if (mi < n && mj < n) dst[mi * n + mj] = s[c][r];
        }
    }
}

// square (n x n) -> condensed upper triangle (download of pdist(x.T)**2)
__global__ void k_square_to_condensed(const double *__restrict__ src, double *__restrict__ dst, int64_t n) {
    const int64_t i = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > i && j < n) dst[gl_condensed_index(i, j, n)] = src[i * n + j];
}

__global__ void k_fill(double *p, int64_t count, double v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) p[i] = v;
}

// Inner products between two problems' last updates (batched path, smooth_learning.py:158-159
// with 2-D input: the norms there are SPECTRAL norms of the (E x B) / (n x B) matrices, which
// only need the B x B Gram matrices of their columns).  Per pair (a, b):
//   out[0] = <dw_a, dw_b>, dw = w_new - w_old (== q - y)     out[1] = <w_old_a, w_old_b>
//   out[2] = <dd_a, dd_b>, dd = d_new - d_old (== qbar-ybar)  out[3] = <d_old_a, d_old_b>
__global__ __launch_bounds__(256) void k_pair_dots(const double *__restrict__ wa_new, const double *__restrict__ wa_old,
                                                   const double *__restrict__ wb_new, const double *__restrict__ wb_old,
                                                   const double *__restrict__ da, const double *__restrict__ dpa,
                                                   const double *__restrict__ db, const double *__restrict__ dpb,
                                                   int64_t n_tiled, int64_t n_nodes, double *__restrict__ part) {
    __shared__ double sh[4][4];
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_tiled; i += (int64_t)gridDim.x * 256) {
        const double ao = wa_old[i], bo = wb_old[i];
        s0 += (wa_new[i] - ao) * (wb_new[i] - bo);
        s1 += ao * bo;
    }
    if (blockIdx.x == 0)
        for (int64_t i = threadIdx.x; i < n_nodes; i += 256) {
            s2 += (da[i] - dpa[i]) * (db[i] - dpb[i]);
            s3 += dpa[i] * dpb[i];
        }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    s3 = wave_sum(s3);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        sh[0][wave] = s0;
        sh[1][wave] = s1;
        sh[2][wave] = s2;
        sh[3][wave] = s3;
    }
    __syncthreads();
    if (threadIdx.x < 4)
        part[4 * blockIdx.x + threadIdx.x] =
            ((sh[threadIdx.x][0] + sh[threadIdx.x][1]) + sh[threadIdx.x][2]) + sh[threadIdx.x][3];
}

// The same four inner products for MANY pairs in one launch: blockIdx.y = pair index into `pairs`
// (session indices a, b), operands through a table of per-session pointers.
struct BatchPtrs {
    const double *w_new, *w_old, *d_new, *d_old;
};
// (the host's exact evaluation; inside a chunk of batched iterations k_batch_gram forms the same matrices on the matrix cores)
__global__ __launch_bounds__(256) void k_batch_dots(const BatchPtrs *__restrict__ ptrs, const int2 *__restrict__ pairs,
                                                    int64_t n_tiled, int64_t n_nodes, double *__restrict__ part) {
    __shared__ double sh[4][4];
    const int2 pr = pairs[blockIdx.x];  // pairs on grid x (up to B (B + 1) / 2 of them), blocks on y
    const BatchPtrs A = ptrs[pr.x], Bq = ptrs[pr.y];
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int64_t i = (int64_t)blockIdx.y * 256 + threadIdx.x; i < n_tiled; i += (int64_t)gridDim.y * 256) {
        const double ao = A.w_old[i], bo = Bq.w_old[i];
        s0 += (A.w_new[i] - ao) * (Bq.w_new[i] - bo);
        s1 += ao * bo;
    }
    if (blockIdx.y == 0)
        for (int64_t i = threadIdx.x; i < n_nodes; i += 256) {
            s2 += (A.d_new[i] - A.d_old[i]) * (Bq.d_new[i] - Bq.d_old[i]);
            s3 += A.d_old[i] * Bq.d_old[i];
        }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    s3 = wave_sum(s3);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        sh[0][wave] = s0;
        sh[1][wave] = s1;
        sh[2][wave] = s2;
        sh[3][wave] = s3;
    }
    __syncthreads();
    if (threadIdx.x < 4)
        part[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 4 + threadIdx.x] =
            ((sh[threadIdx.x][0] + sh[threadIdx.x][1]) + sh[threadIdx.x][2]) + sh[threadIdx.x][3];
}
// out[pair][q] = sum over the blocks of part[pair][block][q], fixed order
__global__ void k_batch_dots_reduce(const double *__restrict__ part, int npairs, int blocks, double *__restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npairs * 4) return;
    const int pr = t >> 2, q = t & 3;
    double acc = 0;
    for (int k = 0; k < blocks; ++k) acc += part[((size_t)pr * blocks + k) * 4 + q];
    out[t] = acc;
}

constexpr int BATCH_BRACKET_MAXB = 32;  // (what k_batch_gram takes: two groups of 16 graphs)
// The coupled rule judged on the device, in ONE launch inside a chunk of batched iterations: the four Gram matrices on the
// matrix cores, their reduction and the brackets of their largest eigenvalues (gl_bracket.h; rigorous) in the workgroup that
// finishes last.  The rule of smooth_learning.py:166-167 is decided for every value the four spectral norms can take: met ->
// every problem's done flag is raised (halt = 3: the host reports the exact values); surely not met -> the batch goes on; a
// tie inside the brackets' final width (1e-9) -> halt = 1, the host decides.  B <= 32.  G = sum over the entries e of a_e a_e' with a_e the B values
// of entry e: v_mfma_f64_16x16x4_f64 takes four entries of 16 graphs at a time -- lane (i = l & 15, k = l >> 4) holds entry k of
// graph i, which is its A operand (i, k) and its B operand (k, j = i) at once -- so a vector is read ONCE (k_batch_dots reads it
// B + 1 times; at B = 20 its 210 pairs cost 45 us per iteration against 6 here).  NG groups of 16 graphs: B <= 16 NG.
// Workgroups 0 .. gridDim.x - 2 share the tiled edge vectors (dW = W_new - W_old, W_old), the last one takes the node vectors.
// Every sum in a fixed order: per wave over its entries, per workgroup over its waves (LDS), then over the workgroups.
typedef double bg_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bg_d4 bg_mfma(double a, double b, bg_d4 acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
#else
    return acc;
#endif
}
constexpr int GRAM_WAVES = 8;
template <int NG>
__global__ __launch_bounds__(GRAM_WAVES * 64) void k_batch_gram(const BatchPtrs *__restrict__ ptrs, int nprob, int64_t n_tiled, int64_t n_pad,
                                                                const long long *__restrict__ it_done, double *__restrict__ part,
                                                                const LogdegDev *__restrict__ devs, double t2, BatchState *st) {
    constexpr int NBLK = NG == 1 ? 1 : 3;  // 16 x 16 blocks of the upper triangle: (0, 0) [, (0, 1), (1, 1)]
    constexpr int SLOT = 2 * NBLK * 256;   // doubles of one wave's accumulators {update, iterate} x blocks
    __shared__ double smem[(GRAM_WAVES / 2) * SLOT];
    __shared__ double br[8];
    __shared__ int s_last;
    if (st->want_brackets == 0) return;
    const unsigned long long T0 = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, gi = lane & 15, kq = lane >> 4;
    const bool node_blk = blockIdx.x == gridDim.x - 1;
    const double *vn[NG], *vo[NG];
    {  // (both parities' pointers are requested beside the parity itself: one round trip instead of two)
        BatchPtrs q0[NG], q1[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int a = 16 * g + gi;
            q0[g] = q1[g] = BatchPtrs{nullptr, nullptr, nullptr, nullptr};
            if (a < nprob) q0[g] = ptrs[a], q1[g] = ptrs[nprob + a];
        }
        const bool odd = (it_done[0] & 1) != 0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const BatchPtrs q = odd ? q1[g] : q0[g];
            vn[g] = node_blk ? q.d_new : q.w_new;
            vo[g] = node_blk ? q.d_old : q.w_old;
        }
    }
    const int64_t units = (node_blk ? n_pad : n_tiled) / 16;
    const int nw = node_blk ? GRAM_WAVES : ((int)gridDim.x - 1) * GRAM_WAVES, w0 = node_blk ? wave : (int)blockIdx.x * GRAM_WAVES + wave;
    bg_d4 acc[2 * NBLK];  // [update: blocks][iterate: blocks]
#pragma unroll
    for (int b = 0; b < 2 * NBLK; ++b) acc[b] = bg_d4{0.0, 0.0, 0.0, 0.0};
    // (the next unit's loads are in flight while this one's eight products run)
    double2 cn[NG][2], co[NG][2];
    auto fetch = [&](int64_t u, double2 (&fn)[NG][2], double2 (&fo)[NG][2]) {
        const int64_t e = 16 * u + 4 * kq;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            fn[g][0] = fn[g][1] = fo[g][0] = fo[g][1] = double2{0.0, 0.0};
            if (vn[g] && u < units) {
                fn[g][0] = *reinterpret_cast<const double2 *>(vn[g] + e), fn[g][1] = *reinterpret_cast<const double2 *>(vn[g] + e + 2);
                fo[g][0] = *reinterpret_cast<const double2 *>(vo[g] + e), fo[g][1] = *reinterpret_cast<const double2 *>(vo[g] + e + 2);
            }
        }
    };
    fetch(w0, cn, co);
    for (int64_t u = w0; u < units; u += nw) {
        double2 nn[NG][2], no[NG][2];
        fetch(u + nw, nn, no);
        double dn[NG][4], od[NG][4];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            od[g][0] = co[g][0].x, od[g][1] = co[g][0].y, od[g][2] = co[g][1].x, od[g][3] = co[g][1].y;
            dn[g][0] = cn[g][0].x - co[g][0].x, dn[g][1] = cn[g][0].y - co[g][0].y;
            dn[g][2] = cn[g][1].x - co[g][1].x, dn[g][3] = cn[g][1].y - co[g][1].y;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[0] = bg_mfma(dn[0][t], dn[0][t], acc[0]);
            acc[NBLK] = bg_mfma(od[0][t], od[0][t], acc[NBLK]);
            if (NG == 2) {
                acc[1] = bg_mfma(dn[0][t], dn[NG - 1][t], acc[1]);
                acc[NBLK - 1] = bg_mfma(dn[NG - 1][t], dn[NG - 1][t], acc[NBLK - 1]);
                acc[NBLK + 1] = bg_mfma(od[0][t], od[NG - 1][t], acc[NBLK + 1]);
                acc[2 * NBLK - 1] = bg_mfma(od[NG - 1][t], od[NG - 1][t], acc[2 * NBLK - 1]);
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) cn[g][0] = nn[g][0], cn[g][1] = nn[g][1], co[g][0] = no[g][0], co[g][1] = no[g][1];
    }
    const unsigned long long T1 = __builtin_amdgcn_s_memrealtime();
    // waves -> workgroup: a fixed tree (wave w + h adds into wave w, h = 8, 4, 2, 1).  Element x of a block: lane x / 4,
    // register x % 4 -> row (lane >> 4) + 4 (x % 4), column lane & 15.
    for (int h = GRAM_WAVES / 2; h >= 1; h >>= 1) {
        if (wave >= h && wave < 2 * h)
#pragma unroll
            for (int b = 0; b < 2 * NBLK; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) smem[(wave - h) * SLOT + b * 256 + lane * 4 + r] = acc[b][r];
        __syncthreads();
        if (wave < h)
#pragma unroll
            for (int b = 0; b < 2 * NBLK; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[b][r] += smem[wave * SLOT + b * 256 + lane * 4 + r];
        __syncthreads();
    }
    if (wave == 0) {
#pragma unroll
        for (int b = 0; b < 2 * NBLK; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                __hip_atomic_store(part + ((size_t)blockIdx.x * 2 * NBLK + b) * 256 + lane * 4 + r, acc[b][r], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {  // (wave 0 is the only one that stored; its lanes share the counter of outstanding stores)
            const unsigned arrived = __hip_atomic_fetch_add(&st->gram_blocks, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            s_last = arrived == gridDim.x ? 1 : 0;
        }
    }
    __syncthreads();
    if (!s_last) return;
    const unsigned long long T2 = __builtin_amdgcn_s_memrealtime();
    // ---- the last workgroup: the four B x B matrices {dW, W, dD, D} into LDS, brackets, decision.
    //      Thread = (matrix k, entry x), 2 NBLK x 256 of them in passes of 512.
    typedef SmMfma<NG> MB;  // (the matrices padded to 16 NG rows of MB::LD doubles: gl_bracket.h)
    static_assert(4 * MB::SIZE <= (GRAM_WAVES / 2) * SLOT, "the four padded matrices live in the reduction's LDS");
    double *gA = smem;
    const int B = nprob, ne = (int)gridDim.x - 1;
    for (int x = tid; x < 4 * MB::SIZE; x += GRAM_WAVES * 64) gA[x] = 0.0;
    __syncthreads();
    // (the workgroups' partials are read with sc1 loads -- written through by their owners before they were counted -- two
    //  entries at a time; the loads of one thread are independent and stay in flight together)
    typedef unsigned int bg_u4 __attribute__((ext_vector_type(4)));
    const auto r_part = __builtin_amdgcn_make_buffer_rsrc((void *)part, 0, (int)((size_t)gridDim.x * 2 * NBLK * 256 * 8), 0x00020000);
    auto two = [](bg_u4 g, double &a, double &b) {
        a = __longlong_as_double((long long)(((unsigned long long)g.y << 32) | g.x));
        b = __longlong_as_double((long long)(((unsigned long long)g.w << 32) | g.z));
    };
    for (int kp = tid; kp < 2 * NBLK * 128; kp += GRAM_WAVES * 64) {
        const int k = kp >> 7, x = 2 * (kp & 127);  // entries x, x + 1 of matrix block k
        const int which = k / NBLK, b = k % NBLK;   // 0: update, 1: iterate
        const int ln = x >> 2, r = x & 3;           // (the lane and registers r, r + 1 these entries came from)
        const int row = (ln >> 4) + 4 * r + (b == 2 ? 16 : 0), col = (ln & 15) + (b >= 1 ? 16 : 0);
        if (row >= B || col >= B) continue;          // (row + 4 >= B as well then)
        double e0 = 0.0, e1 = 0.0;
#pragma unroll 8
        for (int blk = 0; blk < ne; ++blk) {
            double a, c;
            two(__builtin_amdgcn_raw_buffer_load_b128(r_part, (int)((((size_t)blk * 2 * NBLK + k) * 256 + x) * 8), 0, 16), a, c);
            e0 += a, e1 += c;
        }
        double d0, d1;
        two(__builtin_amdgcn_raw_buffer_load_b128(r_part, (int)((((size_t)ne * 2 * NBLK + k) * 256 + x) * 8), 0, 16), d0, d1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rw = row + 4 * h;
            if (rw >= B) break;
            const double ve = h ? e1 : e0, vd = h ? d1 : d0;
            gA[(which + 0) * MB::SIZE + rw * MB::LD + col] = ve;
            gA[(which + 2) * MB::SIZE + rw * MB::LD + col] = vd;
            if (b == 1) {  // (the off-diagonal block fills both triangles; the diagonal blocks are symmetric as computed)
                gA[(which + 0) * MB::SIZE + col * MB::LD + rw] = ve;
                gA[(which + 2) * MB::SIZE + col * MB::LD + rw] = vd;
            }
        }
    }
    __syncthreads();
    const unsigned long long T3 = __builtin_amdgcn_s_memrealtime();
    // brackets in stages (each squaring halves the logarithmic width; four, then six more at a time): an iterate clearly short
    // of the rule -- most of them -- is decided after the first stage; the full 34 squarings are only run out by a tie
    __shared__ int s_dec;
    SmBracket S{};
    if (wave < 4) MB::begin(gA + wave * MB::SIZE, lane, S);
    for (int round = 0; round < 6; ++round) {
        if (wave < 4) {
            double lo, hi;
            MB::advance(gA + wave * MB::SIZE, lane, S, round == 0 ? 4 : 6);
            MB::bounds(gA + wave * MB::SIZE, lane, S, lo, hi);
            if (lane == 0) br[2 * wave] = lo, br[2 * wave + 1] = hi;
        }
        __syncthreads();
        if (tid == 0) {  // {lo, hi} x {update, iterate, dual update, dual}
            const double tp = t2 * (1 + 1e-9), tm = t2 * (1 - 1e-9);
            const bool not_yet = br[0] > tp * br[3] || br[4] > tp * br[7];
            const bool met = br[2] > 0 && br[6] > 0 && br[1] < tm * br[2] && br[5] < tm * br[6];
            s_dec = met ? 3 : (not_yet ? 0 : 1);
        }
        __syncthreads();
        if (s_dec != 1) break;  // (1: undecided at this width)
    }
    if (tid == 0) {
        const unsigned long long T4 = __builtin_amdgcn_s_memrealtime();
        st->diag[0] += (double)(T1 - T0), st->diag[1] += (double)(T2 - T1), st->diag[2] += (double)(T3 - T2), st->diag[3] += (double)(T4 - T3);
        st->want_brackets = 0;
        st->gram_blocks = 0u;
        const int halt = s_dec;
        if (halt) {
            for (int a = 0; a < nprob; ++a) devs[a].ctr[0] = 1;
            st->halt = halt;
        }
    }
}

// The same judgement for a whole CHUNK of iterations at once (the deciding chunks of a cruise, below): the tile engine has
// written the iterate of every iteration of its launch to a ring of tiled batches (slot j: after j iterations) and the dual
// variable beside it; blockIdx.y = j judges iteration j of the chunk (update = slot j + 1 - slot j, iterate = slot j) exactly
// as k_batch_gram does -- Gram blocks on MFMA, fixed reduction tree, staged brackets in the last workgroup of its row -- and
// leaves decision[j] = 0 (surely not met), 3 (met) or 1 (a tie inside 1e-9: the host decides).
struct RingJudge {
    unsigned done[64];
    int decision[64];
};
template <int NG>
__global__ __launch_bounds__(GRAM_WAVES * 64) __attribute__((amdgpu_waves_per_eu(4))) void k_batch_gram_ring(const double *__restrict__ ring, const double *__restrict__ dring, int nprob,
                                                                     int64_t ring_stride, int64_t tiled1, int64_t dring_stride, int64_t n_pad1,
                                                                     double *__restrict__ part_all, double t2, RingJudge *rj) {
    constexpr int NBLK = NG == 1 ? 1 : 3;
    constexpr int SLOT = 2 * NBLK * 256;
    __shared__ double smem[(GRAM_WAVES / 2) * SLOT];
    __shared__ double br[8];
    __shared__ int s_last, s_dec;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, gi = lane & 15, kq = lane >> 4, j = blockIdx.y;
    const bool node_blk = blockIdx.x == gridDim.x - 1;
    double *part = part_all + (size_t)j * gridDim.x * 2 * NBLK * 256;
    const double *vn[NG], *vo[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int a = 16 * g + gi;
        vn[g] = vo[g] = nullptr;
        if (a < nprob) {
            vo[g] = node_blk ? dring + (int64_t)j * dring_stride + a * n_pad1 : ring + (int64_t)j * ring_stride + a * tiled1;
            vn[g] = vo[g] + (node_blk ? dring_stride : ring_stride);
        }
    }
    const int64_t units = (node_blk ? n_pad1 : tiled1) / 16;
    const int nw = node_blk ? GRAM_WAVES : ((int)gridDim.x - 1) * GRAM_WAVES, w0 = node_blk ? wave : (int)blockIdx.x * GRAM_WAVES + wave;
    bg_d4 acc[2 * NBLK];
#pragma unroll
    for (int b = 0; b < 2 * NBLK; ++b) acc[b] = bg_d4{0.0, 0.0, 0.0, 0.0};
    double2 cn[NG][2], co[NG][2];
    auto fetch = [&](int64_t u, double2 (&fn)[NG][2], double2 (&fo)[NG][2]) {
        const int64_t e = 16 * u + 4 * kq;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            fn[g][0] = fn[g][1] = fo[g][0] = fo[g][1] = double2{0.0, 0.0};
            if (vn[g] && u < units) {
                fn[g][0] = *reinterpret_cast<const double2 *>(vn[g] + e), fn[g][1] = *reinterpret_cast<const double2 *>(vn[g] + e + 2);
                fo[g][0] = *reinterpret_cast<const double2 *>(vo[g] + e), fo[g][1] = *reinterpret_cast<const double2 *>(vo[g] + e + 2);
            }
        }
    };
    fetch(w0, cn, co);
    for (int64_t u = w0; u < units; u += nw) {
        double2 nn[NG][2], no[NG][2];
        fetch(u + nw, nn, no);
        double dn[NG][4], od[NG][4];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            od[g][0] = co[g][0].x, od[g][1] = co[g][0].y, od[g][2] = co[g][1].x, od[g][3] = co[g][1].y;
            dn[g][0] = cn[g][0].x - co[g][0].x, dn[g][1] = cn[g][0].y - co[g][0].y;
            dn[g][2] = cn[g][1].x - co[g][1].x, dn[g][3] = cn[g][1].y - co[g][1].y;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[0] = bg_mfma(dn[0][t], dn[0][t], acc[0]);
            acc[NBLK] = bg_mfma(od[0][t], od[0][t], acc[NBLK]);
            if (NG == 2) {
                acc[1] = bg_mfma(dn[0][t], dn[NG - 1][t], acc[1]);
                acc[NBLK - 1] = bg_mfma(dn[NG - 1][t], dn[NG - 1][t], acc[NBLK - 1]);
                acc[NBLK + 1] = bg_mfma(od[0][t], od[NG - 1][t], acc[NBLK + 1]);
                acc[2 * NBLK - 1] = bg_mfma(od[NG - 1][t], od[NG - 1][t], acc[2 * NBLK - 1]);
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) cn[g][0] = nn[g][0], cn[g][1] = nn[g][1], co[g][0] = no[g][0], co[g][1] = no[g][1];
    }
    for (int h = GRAM_WAVES / 2; h >= 1; h >>= 1) {
        if (wave >= h && wave < 2 * h)
#pragma unroll
            for (int b = 0; b < 2 * NBLK; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) smem[(wave - h) * SLOT + b * 256 + lane * 4 + r] = acc[b][r];
        __syncthreads();
        if (wave < h)
#pragma unroll
            for (int b = 0; b < 2 * NBLK; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[b][r] += smem[wave * SLOT + b * 256 + lane * 4 + r];
        __syncthreads();
    }
    if (wave == 0) {
#pragma unroll
        for (int b = 0; b < 2 * NBLK; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                __hip_atomic_store(part + ((size_t)blockIdx.x * 2 * NBLK + b) * 256 + lane * 4 + r, acc[b][r], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            const unsigned arrived = __hip_atomic_fetch_add(&rj->done[j], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            s_last = arrived == gridDim.x ? 1 : 0;
        }
    }
    __syncthreads();
    if (!s_last) return;
    typedef SmMfma<NG> MB;
    static_assert(4 * MB::SIZE <= (GRAM_WAVES / 2) * SLOT, "the four padded matrices live in the reduction's LDS");
    double *gA = smem;
    const int B = nprob, ne = (int)gridDim.x - 1;
    for (int x = tid; x < 4 * MB::SIZE; x += GRAM_WAVES * 64) gA[x] = 0.0;
    __syncthreads();
    typedef unsigned int bg_u4 __attribute__((ext_vector_type(4)));
    const auto r_part = __builtin_amdgcn_make_buffer_rsrc((void *)part, 0, (int)((size_t)gridDim.x * 2 * NBLK * 256 * 8), 0x00020000);
    auto two = [](bg_u4 g, double &a, double &b) {
        a = __longlong_as_double((long long)(((unsigned long long)g.y << 32) | g.x));
        b = __longlong_as_double((long long)(((unsigned long long)g.w << 32) | g.z));
    };
    for (int kp = tid; kp < 2 * NBLK * 128; kp += GRAM_WAVES * 64) {
        const int k = kp >> 7, x = 2 * (kp & 127);
        const int which = k / NBLK, b = k % NBLK;
        const int ln = x >> 2, r = x & 3;
        const int row = (ln >> 4) + 4 * r + (b == 2 ? 16 : 0), col = (ln & 15) + (b >= 1 ? 16 : 0);
        if (row >= B || col >= B) continue;
        double e0 = 0.0, e1 = 0.0;
#pragma unroll 8
        for (int blk = 0; blk < ne; ++blk) {
            double a, c;
            two(__builtin_amdgcn_raw_buffer_load_b128(r_part, (int)((((size_t)blk * 2 * NBLK + k) * 256 + x) * 8), 0, 16), a, c);
            e0 += a, e1 += c;
        }
        double d0, d1;
        two(__builtin_amdgcn_raw_buffer_load_b128(r_part, (int)((((size_t)ne * 2 * NBLK + k) * 256 + x) * 8), 0, 16), d0, d1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rw = row + 4 * h;
            if (rw >= B) break;
            const double ve = h ? e1 : e0, vd = h ? d1 : d0;
            gA[(which + 0) * MB::SIZE + rw * MB::LD + col] = ve;
            gA[(which + 2) * MB::SIZE + rw * MB::LD + col] = vd;
            if (b == 1) {
                gA[(which + 0) * MB::SIZE + col * MB::LD + rw] = ve;
                gA[(which + 2) * MB::SIZE + col * MB::LD + rw] = vd;
            }
        }
    }
    __syncthreads();
    SmBracket S{};
    if (wave < 4) MB::begin(gA + wave * MB::SIZE, lane, S);
    for (int round = 0; round < 6; ++round) {
        if (wave < 4) {
            double lo, hi;
            MB::advance(gA + wave * MB::SIZE, lane, S, round == 0 ? 4 : 6);
            MB::bounds(gA + wave * MB::SIZE, lane, S, lo, hi);
            if (lane == 0) br[2 * wave] = lo, br[2 * wave + 1] = hi;
        }
        __syncthreads();
        if (tid == 0) {
            const double tp = t2 * (1 + 1e-9), tm = t2 * (1 - 1e-9);
            const bool not_yet = br[0] > tp * br[3] || br[4] > tp * br[7];
            const bool met = br[2] > 0 && br[6] > 0 && br[1] < tm * br[2] && br[5] < tm * br[6];
            s_dec = met ? 3 : (not_yet ? 0 : 1);
        }
        __syncthreads();
        if (s_dec != 1) break;
    }
    if (tid == 0) {
        rj->decision[j] = s_dec;
        rj->done[j] = 0u;
    }
}

int check(gl_logdeg *s) { return (s && s->ctx) ? GL_OK : GL_E_INVALID; }

// Where the column partials go.  The partition of the flat row-pair list over the
// G waves is static, so the host knows which (wave, segment) holds partials of which
// tile: colslab entries are laid out contiguously per column block J (col_ptr, read
// by the node pass without indirection) and seg_dest[wave*kmax + seg] tells the edge
// pass where to put them.
int build_column_index(gl_logdeg *s, std::vector<int> &ptr, std::vector<int> &dest) {
    gl_ctx *ctx = s->ctx;
    LogdegDev &D = s->dev;
    std::vector<std::vector<int>> lists((size_t)D.nb);
    for (int b = 0; b < D.G; ++b) {
        int64_t p = ((int64_t)b * D.npairs) / D.G;
        const int64_t pend = ((int64_t)(b + 1) * D.npairs) / D.G;
        int seg = 0;
        int64_t last_tile = -1;
        while (p < pend) {
            const int64_t t = p >> 5;
            const int a = (int)(p & (PAIRS_PER_TILE - 1));
            const int cnt = (int)std::min<int64_t>(PAIRS_PER_TILE - a, pend - p);
            if (t != last_tile) {
                int I, J;
                gl_tile_decode(t, D.nb, I, J);
                if (seg >= D.kmax) return gl_set_error(ctx, GL_E_STATE, "internal: segment table overflow");
                lists[(size_t)J].push_back(b * D.kmax + seg);
                ++seg;
                last_tile = t;
            }
            p += cnt;
        }
    }
    ptr.assign((size_t)D.nb + 1, 0);
    dest.assign((size_t)D.G * D.kmax, 0);
    for (int64_t J = 0; J < D.nb; ++J) {
        int pos = ptr[(size_t)J];
        for (int key : lists[(size_t)J]) dest[(size_t)key] = pos++;
        ptr[(size_t)J + 1] = pos;
    }
    s->n_segments = ptr[(size_t)D.nb];
    return GL_OK;
}

void drop_graph(gl_logdeg *s) {
    if (s->graph) (void)hipGraphExecDestroy(s->graph);
    s->graph = nullptr;
    s->graph_iters = 0;
}

void enqueue_iteration(gl_logdeg *s) {
    const LogdegDev &D = s->dev;
    k_edge<false><<<dim3((unsigned)D.G), dim3(64), 0, s->ctx->stream>>>(D, D.dvar, D.pbar);
    k_node<false><<<dim3((unsigned)D.nb), dim3(NODE_WAVES * 64), 0, s->ctx->stream>>>(D);
}

// ITER_BATCH iterations captured once per init() into a graph: the kernels take
// their iteration index from device memory, so replaying the graph continues the loop.
int ensure_graph(gl_logdeg *s) {
    if (s->graph || getenv("GRAPHLEARN_AMD_NO_GRAPH")) return GL_OK;
    gl_ctx *ctx = s->ctx;
    hipGraph_t graph = nullptr;
    GL_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < ITER_BATCH; ++k) enqueue_iteration(s);
    hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
    if (e != hipSuccess) return gl_set_error(ctx, GL_E_HIP, "graph capture failed: %s", hipGetErrorString(e));
    e = hipGraphInstantiate(&s->graph, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        s->graph = nullptr;
        return gl_set_error(ctx, GL_E_HIP, "graph instantiate failed: %s", hipGetErrorString(e));
    }
    s->graph_iters = ITER_BATCH;
    return GL_OK;
}

int launch_iterations(gl_logdeg *s, int64_t count) {
    gl_ctx *ctx = s->ctx;
    int64_t left = count;
    if (left >= ITER_BATCH) GL_TRY(ensure_graph(s));
    while (s->graph && left >= s->graph_iters) {
        GL_HIP(ctx, hipGraphLaunch(s->graph, ctx->stream));
        left -= s->graph_iters;
    }
    for (; left > 0; --left) enqueue_iteration(s);
    GL_HIP(ctx, hipGetLastError());
    s->it += count;
    return GL_OK;
}

}  // namespace

int gl_logdeg_tiled_to_condensed(gl_logdeg *s, const double *src_tiled, double *dst_condensed) {
    k_to_condensed<<<dim3((unsigned)s->dev.ntiles), dim3(256), 0, s->ctx->stream>>>(
        src_tiled, dst_condensed, s->n, s->dev.nb, -std::numeric_limits<double>::infinity());
    GL_HIP(s->ctx, hipGetLastError());
    return GL_OK;
}
int gl_logdeg_condensed_to_tiled(gl_logdeg *s, const double *src_condensed, double *dst_tiled) {
    k_to_tiled<0><<<dim3((unsigned)s->dev.ntiles), dim3(256), 0, s->ctx->stream>>>(src_condensed, dst_tiled, s->n, s->dev.nb,
                                                                                  1.0);
    GL_HIP(s->ctx, hipGetLastError());
    return GL_OK;
}

// ===========================================================================
GL_API int gl_logdeg_create(gl_ctx *ctx, int64_t n_nodes, gl_logdeg **out) {
    if (!ctx || !out) return GL_E_INVALID;
    *out = nullptr;
    if (n_nodes < 2) return gl_set_error(ctx, GL_E_INVALID, "n_nodes must be >= 2, got %lld", (long long)n_nodes);
    GL_HIP(ctx, hipSetDevice(ctx->device));
    gl_logdeg *s = new (std::nothrow) gl_logdeg();
    if (!s) return GL_E_NOMEM;
    s->ctx = ctx;
    s->n = n_nodes;
    s->n_edges = n_nodes * (n_nodes - 1) / 2;
    LogdegDev &D = s->dev;
    D.n = n_nodes;
    D.nb = gl_num_blocks(n_nodes);
    D.n_pad = D.nb * GL_TB;
    D.ntiles = gl_num_tiles(n_nodes);
    D.npairs = D.ntiles * PAIRS_PER_TILE;
    // persistent edge grid: `wpc` single-wave workgroups per CU (all co-resident), >= 8 row pairs each
    s->wpc = 4 * GL_EDGE_WAVES_PER_SIMD;
    if (const char *e = getenv("GRAPHLEARN_AMD_WPC")) s->wpc = std::max(1, std::min(16, atoi(e)));
    D.G = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)s->wpc * ctx->prop.multiProcessorCount, D.npairs / 8));
    const int64_t maxchunk = (D.npairs + D.G - 1) / D.G;
    D.kmax = (int)((maxchunk + PAIRS_PER_TILE - 1) / PAIRS_PER_TILE) + 1;
    const size_t tiled_bytes = (size_t)D.ntiles * GL_TILE_ELEMS * sizeof(double);
    std::vector<int> ptr, dest;
    int rc = build_column_index(s, ptr, dest);
    if (rc != GL_OK) {
        delete s;
        return rc;
    }
    // every buffer of the session in ONE allocation (a session used to cost 17 hipMalloc / hipFree pairs and
    // two synchronous pageable copies: 150 us, most of the set-up of a batch of small graphs)
    double *zbuf = nullptr;
    struct Part {
        void **p;
        size_t bytes;
    };
    const Part parts[] = {
        // w[0], w[1], bnorm, dprev, ctr in this order: init() zeroes them with one memset
        {(void **)&D.w[0], tiled_bytes}, {(void **)&D.w[1], tiled_bytes}, {(void **)&D.bnorm, (size_t)2 * D.nb * 4 * 8},
        {(void **)&D.dprev, (size_t)D.n_pad * 8}, {(void **)&D.ctr, 8 * 8},
        {(void **)&zbuf, tiled_bytes}, {(void **)&D.dvar, (size_t)D.n_pad * 8},
        {(void **)&D.ybar, (size_t)D.n_pad * 8}, {(void **)&D.pbar, (size_t)D.n_pad * 8},
        {(void **)&D.rowslab_p, (size_t)D.nb * D.n_pad * 8}, {(void **)&D.rowslab_w, (size_t)D.nb * D.n_pad * 8},
        {(void **)&D.tnorm, (size_t)D.G * 2 * 8},
        {(void **)&D.dstat, 8 * 8}, {(void **)&s->stage_rec, (size_t)(s->n_edges + 8) * 8},
        {(void **)&s->col_ptr, ptr.size() * sizeof(int)}, {(void **)&s->seg_dest, dest.size() * sizeof(int)},
        {(void **)&D.colslab, (size_t)std::max(s->n_segments, 1) * 2 * GL_TB * 8}};
    size_t total = 0;
    for (const Part &q : parts) total += (q.bytes + 255) / 256 * 256;
    if ((rc = gl_ctx_alloc(ctx, total, &s->arena)) != GL_OK) {  // (a block of this size released by an earlier session, or hipMalloc)
        delete s;
        return rc;
    }
    s->arena_bytes = total;
    {
        char *at = (char *)s->arena;
        for (const Part &q : parts) {
            *q.p = at;
            at += (q.bytes + 255) / 256 * 256;
        }
    }
    D.z = zbuf;
    s->stage = s->stage_rec + 8;  // the 64-byte record header of the gather payload sits right in front (wire format)
    D.col_ptr = s->col_ptr;
    D.seg_dest = s->seg_dest;
    // (the persistent engines are prepared at the session's first own iterate call -- ensure_engines --: the sessions of a
    //  batch, which runs on the batch's own stand-in, never pay for their decomposition, allocation and upload)
    if ((rc = gl_h2d(ctx, s->col_ptr, ptr.data(), ptr.size() * sizeof(int))) != GL_OK ||
        (rc = gl_h2d(ctx, s->seg_dest, dest.data(), dest.size() * sizeof(int))) != GL_OK) {
        gl_logdeg_destroy(s);
        return rc;
    }
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));  // ptr / dest are locals: the staged uploads have left them
    *out = s;
    return GL_OK;
}

GL_API int gl_logdeg_destroy(gl_logdeg *s) {
    if (!s) return GL_OK;
    if (s->ctx) {
        (void)hipSetDevice(s->ctx->device);
        (void)hipStreamSynchronize(s->ctx->stream);
    }
    drop_graph(s);
    gl_resident_free(s);
    gl_tile_free(s);
    if (s->ctx) {  // (kept by the context for the next session of this size)
        gl_ctx_release(s->ctx, s->arena, s->arena_bytes);
        gl_ctx_release(s->ctx, s->sig_arena, s->sig_bytes);
    } else {
        if (s->arena) (void)hipFree(s->arena);
        if (s->sig_arena) (void)hipFree(s->sig_arena);
    }
    if (s->square_out) (void)hipFree(s->square_out);
    delete s;
    return GL_OK;
}

GL_API int gl_logdeg_set_distances(gl_logdeg *s, const double *z_condensed) {
    GL_TRY(check(s));
    gl_ctx *ctx = s->ctx;
    if (!z_condensed) return gl_set_error(ctx, GL_E_INVALID, "z_condensed is NULL");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    GL_TRY(gl_h2d(ctx, s->stage, z_condensed, (size_t)s->n_edges * 8));
    k_to_tiled<0><<<dim3((unsigned)s->dev.ntiles), dim3(256), 0, ctx->stream>>>(
        s->stage, const_cast<double *>(s->dev.z), s->n, s->dev.nb, 1.0);
    GL_HIP(ctx, hipGetLastError());
    if ((size_t)s->n_edges * 8 > GL_STAGE_BYTES) GL_HIP(ctx, hipStreamSynchronize(ctx->stream));  // pageable source still in use
    s->have_z = true;
    s->have_zsq = false;
    s->theta = 1.0;
    s->theta_pending = false;  // a theta left on the device by an earlier set_signals belongs to the old distances
    s->inited = false;
    return GL_OK;
}

GL_API int gl_logdeg_set_signals(gl_logdeg *s, const double *x, int64_t d, int64_t n_nodes, int avg_degree,
                                 double *theta_out) {
    GL_TRY(check(s));
    gl_ctx *ctx = s->ctx;
    if (!x || d < 1) return gl_set_error(ctx, GL_E_INVALID, "x is NULL or d < 1");
    if (n_nodes != s->n)
        return gl_set_error(ctx, GL_E_INVALID, "x has %lld nodes, session was created for %lld", (long long)n_nodes,
                            (long long)s->n);
    if (avg_degree > 0 && avg_degree >= s->n)
        return gl_set_error(ctx, GL_E_INVALID, "avg_degree %d must be < n_nodes %lld", avg_degree, (long long)s->n);
    GL_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t xe = d * s->n;
    if (s->xdev_elems < xe || !s->zsq) {  // signals, squared distances and the theta terms in one allocation
        gl_ctx_release(ctx, s->sig_arena, s->sig_bytes);
        s->sig_arena = nullptr;
        s->xdev = s->zsq = s->theta_terms = s->gram_diag = nullptr;
        s->xdev_elems = 0;
        const size_t bz = ((size_t)s->n * s->n * 8 + 255) / 256 * 256, bt = ((size_t)3 * s->n * 8 + 255) / 256 * 256;
        s->sig_bytes = bz + bt + (size_t)2 * xe * 8;  // x and its centred copy
        GL_TRY(gl_ctx_alloc(ctx, s->sig_bytes, &s->sig_arena));
        s->zsq = (double *)s->sig_arena;
        s->theta_terms = (double *)((char *)s->sig_arena + bz);
        s->gram_diag = s->theta_terms + 2 * s->n;
        s->xdev = (double *)((char *)s->sig_arena + bz + bt);
        s->xdev_elems = xe;
    }
    GL_TRY(gl_h2d(ctx, s->xdev, x, (size_t)xe * 8));
    GL_TRY(gl_gram_distances(ctx, s->xdev, s->xdev + s->xdev_elems, d, s->n, s->zsq, s->gram_diag));
    double theta = 1.0;
    // theta_out == NULL: the caller fetches theta later (gl_logdeg_theta); it then never leaves the device before the
    // distances are scaled, and a batch of windows goes through without a round trip each (the finish of a large graph's
    // theta by one thread costs tens of microseconds: only below 2048 nodes)
    const bool deferred = !theta_out && avg_degree > 0 && s->n <= 2048;
    s->theta_pending = deferred;
    if (deferred) {
        double *theta_dev = s->dev.dstat + 4;
        GL_TRY(gl_theta_from_square_dev(ctx, s->zsq, s->n, avg_degree, s->theta_terms, theta_dev));
        k_to_tiled<1><<<dim3((unsigned)s->dev.ntiles), dim3(256), 0, ctx->stream>>>(
            s->zsq, const_cast<double *>(s->dev.z), s->n, s->dev.nb, 1.0, theta_dev);
    } else {
        if (avg_degree > 0) GL_TRY(gl_theta_from_square(ctx, s->zsq, s->n, avg_degree, s->theta_terms, &theta));
        k_to_tiled<1><<<dim3((unsigned)s->dev.ntiles), dim3(256), 0, ctx->stream>>>(
            s->zsq, const_cast<double *>(s->dev.z), s->n, s->dev.nb, theta);
    }
    GL_HIP(ctx, hipGetLastError());
    if ((size_t)xe * 8 > GL_STAGE_BYTES) GL_HIP(ctx, hipStreamSynchronize(ctx->stream));  // pageable source still in use
    s->theta = theta;
    s->have_z = true;
    s->have_zsq = true;
    s->inited = false;
    if (theta_out) *theta_out = theta;
    return GL_OK;
}

GL_API int gl_logdeg_theta(gl_logdeg *s, double *theta_out) {
    GL_TRY(check(s));
    if (!theta_out) return gl_set_error(s->ctx, GL_E_INVALID, "theta_out is NULL");
    if (s->theta_pending) {
        GL_HIP(s->ctx, hipSetDevice(s->ctx->device));
        GL_TRY(gl_d2h(s->ctx, &s->theta, s->dev.dstat + 4, 8));
        s->theta_pending = false;
    }
    *theta_out = s->theta;
    return GL_OK;
}

GL_API int gl_logdeg_get_distances(gl_logdeg *s, double *z_condensed) {
    GL_TRY(check(s));
    gl_ctx *ctx = s->ctx;
    if (!z_condensed) return gl_set_error(ctx, GL_E_INVALID, "z_condensed is NULL");
    if (!s->have_zsq) return gl_set_error(ctx, GL_E_STATE, "get_distances needs a preceding set_signals");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    dim3 grid((unsigned)((s->n + 255) / 256), (unsigned)s->n);
    k_square_to_condensed<<<grid, dim3(256), 0, ctx->stream>>>(s->zsq, s->stage, s->n);
    GL_HIP(ctx, hipGetLastError());
    GL_TRY(gl_d2h(ctx, z_condensed, s->stage, (size_t)s->n_edges * 8));
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GL_OK;
}

GL_API int gl_logdeg_init(gl_logdeg *s, double alpha, double beta, double step_size, const double *edge_init) {
    GL_TRY(check(s));
    gl_ctx *ctx = s->ctx;
    if (!s->have_z) return gl_set_error(ctx, GL_E_STATE, "init needs set_signals or set_distances first");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    LogdegDev &D = s->dev;
    drop_graph(s);  // the scalars below are baked into captured launches
    // :135-137
    D.gamma = step_size / (2 * beta + sqrt((double)(2 * (s->n - 1))));
    D.c = 1 - D.gamma * 2 * beta;
    D.g2 = 2 * D.gamma;
    D.a4g = 4 * alpha * D.gamma;
    const size_t edge_bytes = (size_t)s->n_edges * 8;
    if (edge_init) {
        GL_TRY(gl_h2d(ctx, s->stage, edge_init, edge_bytes));
        k_to_tiled<0><<<dim3((unsigned)D.ntiles), dim3(256), 0, ctx->stream>>>(s->stage, D.w[0], s->n, D.nb, 1.0);
        GL_HIP(ctx, hipGetLastError());
    }
    // w[1], bnorm, dprev, ctr (and w[0] without a start) are adjacent in the session's allocation: one memset
    {
        char *from = (char *)(edge_init ? D.w[1] : D.w[0]);
        GL_HIP(ctx, hipMemsetAsync(from, 0, (size_t)((char *)D.ctr + 8 * 8 - from), ctx->stream));
    }
    s->it = 0;
    ++s->init_version;
    k_edge<true><<<dim3((unsigned)D.G), dim3(64), 0, ctx->stream>>>(D, D.dvar, D.pbar);
    GL_HIP(ctx, hipGetLastError());
    k_node<true><<<dim3((unsigned)D.nb), dim3(NODE_WAVES * 64), 0, ctx->stream>>>(D);  // also dstat = {inf, inf, tol 0}
    GL_HIP(ctx, hipGetLastError());
    // stream-ordered from here on; only a start handed over in pageable memory too large for the staging
    // area is still being read by the copy engine
    if (edge_init && edge_bytes > GL_STAGE_BYTES) GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    s->tol_dev = 0.0;  // k_node<true> has just written it
    s->tol_set = true;
    s->inited = true;
    return GL_OK;
}

// The resident engine (gl_resident.hip) takes over when it applies and the call is long enough
// to amortise loading / storing the register-held iterate.  GRAPHLEARN_AMD_ENGINE=stream disables it.
static int ensure_engines(gl_logdeg *s) {
    if (s->engines_ready) return GL_OK;
    s->engines_ready = true;  // (also after a failure: the streaming engine serves the session then)
    GL_HIP(s->ctx, hipSetDevice(s->ctx->device));
    GL_TRY(gl_resident_prepare(s));
    GL_TRY(gl_tile_prepare(s));
    return GL_OK;
}
static bool use_small(const gl_logdeg *s, int64_t iters) { return iters >= 8 && gl_small_applicable(s); }
// (a call of the tile engine costs ~30 us beyond its iterations -- loading w and half of z, storing w -- and 5-7 us per
//  iteration, the streaming engine 10-40 us per iteration: from four iterations on the tile engine is never slower)
static bool use_resident(const gl_logdeg *s, int64_t iters) {
    return s->tile != nullptr ? iters >= GL_TILE_MIN_CALL : (s->res != nullptr && iters >= 16);
}
// the persistent launch of the session's resident-family engine: the tile engine where it applies
static int persistent_launch(gl_logdeg *s, int64_t iters, double tol, bool norms, int final_parity = -1, double gate = 0.0) {
    return s->tile ? gl_tile_launch(s, iters, tol, norms, final_parity, gate) : gl_resident_launch(s, iters, tol);
}
static void persistent_free(gl_logdeg *s) {
    gl_resident_free(s);
    gl_tile_free(s);
}

// error word of earlier asynchronous resident launches (read at the next synchronising call).  A launch whose
// workgroups never met (shared GPU) modifies nothing, records where it stood (ctr[2]) and raises ctr[4] = 1;
// launches queued behind it see the word and return at once, so everything from ctr[2] on is redone here on
// the streaming engine.  ctr[4] = 2: a barrier failed after the rendezvous -- an error.
static int resident_check(gl_logdeg *s) {
    if (!s->res_unchecked) return GL_OK;
    gl_ctx *ctx = s->ctx;
    long long *ct = (long long *)((double *)ctx->pinned + 8);
    GL_HIP(ctx, hipMemcpyAsync(ct, s->dev.ctr, 5 * 8, hipMemcpyDeviceToHost, ctx->stream));
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    s->res_unchecked = false;
    gl_tile_dump_stamps(s);  // development only (GRAPHLEARN_AMD_RES_STAMPS)
    if (ct[4] == 0) return GL_OK;
    const int64_t stood = (int64_t)ct[2];  // (ctr[1] is re-armed after every launch; ctr[2] = the streaming engine's iteration index)
    if (ct[4] != 1 || stood < s->it - s->res_pending || stood > s->it)
        return gl_set_error(ctx, GL_E_HIP, "resident engine: grid barrier timed out in mid-run");
    GL_HIP(ctx, hipMemsetAsync(s->dev.ctr + 4, 0, 8, ctx->stream));
    persistent_free(s);
    const int64_t redo = s->it - stood;
    s->it = stood;
    s->res_pending = 0;
    return launch_iterations(s, redo);
}

// tol lives in dstat[2] for the streaming kernels; uploaded only when it changes (the pinned slot is fenced
// by a stream sync before it is rewritten)
static int set_tol(gl_logdeg *s, double tol) {
    gl_ctx *ctx = s->ctx;
    if (s->tol_set && s->tol_dev == tol) return GL_OK;
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the slot may still be in flight
    double *slot = (double *)ctx->pinned + 16;
    *slot = tol;
    GL_HIP(ctx, hipMemcpyAsync(s->dev.dstat + 2, slot, 8, hipMemcpyHostToDevice, ctx->stream));
    s->tol_dev = tol;
    s->tol_set = true;
    return GL_OK;
}

// Stream-ordered: nothing here waits for the device (no synchronisation, no read-back) once tol = 0 is in place,
// so a caller's event bracket around this call contains the kernels only.
GL_API int gl_logdeg_iterate_async(gl_logdeg *s, int64_t n_iter) {
    GL_TRY(check(s));
    gl_ctx *ctx = s->ctx;
    if (!s->inited) return gl_set_error(ctx, GL_E_STATE, "iterate needs init first");
    if (n_iter < 0) return gl_set_error(ctx, GL_E_INVALID, "n_iter < 0");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    GL_TRY(set_tol(s, 0.0));
    GL_TRY(ensure_engines(s));
    if (use_small(s, n_iter)) {  // tol = 0 never stops: the iteration count is known now
        GL_TRY(gl_small_launch(s, n_iter, 0.0));
        GL_HIP(ctx, hipMemsetAsync(s->dev.ctr, 0, 2 * 8, ctx->stream));
        s->it += n_iter;
        return GL_OK;
    }
    if (use_resident(s, n_iter)) {
        // tol = 0 never stops: the iteration count and the parity of the final buffer are known now
        const int64_t itf = s->it + n_iter;
        const bool direct = gl_tile_writes_final_parity(s);  // the tile engine writes w[itf & 1] itself (below the rounds)
        GL_TRY(persistent_launch(s, n_iter, 0.0, false, direct ? (int)(itf & 1) : -1));  // nobody reads the ratios of an asynchronous call
        if (!direct && (itf & 1) != (s->it & 1)) {
            const size_t bytes = (size_t)s->dev.ntiles * GL_TILE_ELEMS * sizeof(double);
            GL_HIP(ctx, hipMemcpyAsync(s->dev.w[itf & 1], s->dev.w[s->it & 1], bytes, hipMemcpyDeviceToDevice,
                                       ctx->stream));
        }
        // (the tile engine leaves ctr[0] = 0 itself after a tol = 0 call: its launch is alone on the stream)
        if (!s->tile) GL_HIP(ctx, hipMemsetAsync(s->dev.ctr, 0, 2 * 8, ctx->stream));
        s->it = itf;
        s->res_pending = (s->res_unchecked ? s->res_pending : 0) + n_iter;
        s->res_unchecked = true;
        return GL_OK;
    }
    GL_TRY(resident_check(s));
    return launch_iterations(s, n_iter);
}

GL_API int gl_logdeg_iterate(gl_logdeg *s, int64_t maxit, double tol, int64_t *n_iter_out, int *converged_out,
                             double *rel_out) {
    GL_TRY(check(s));
    gl_ctx *ctx = s->ctx;
    if (!s->inited) return gl_set_error(ctx, GL_E_STATE, "iterate needs init first");
    if (maxit < 0) return gl_set_error(ctx, GL_E_INVALID, "maxit < 0");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    GL_TRY(resident_check(s));
    LogdegDev &D = s->dev;
    GL_TRY(set_tol(s, tol));
    GL_TRY(ensure_engines(s));
    if (use_small(s, maxit)) {
        double *st = (double *)ctx->pinned;
        long long *ct = (long long *)(st + 8);
        GL_TRY(gl_small_launch(s, maxit, tol));
        GL_HIP(ctx, hipMemcpyAsync(st, D.dstat, 2 * 8, hipMemcpyDeviceToHost, ctx->stream));
        GL_HIP(ctx, hipMemcpyAsync(ct, D.ctr, 4 * 8, hipMemcpyDeviceToHost, ctx->stream));
        GL_HIP(ctx, hipMemsetAsync(D.ctr, 0, 2 * 8, ctx->stream));  // the done flag belongs to this call
        GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (n_iter_out) *n_iter_out = (int64_t)ct[1] - s->it;
        if (converged_out) *converged_out = ct[0] != 0 ? 1 : 0;
        if (rel_out) {
            rel_out[0] = st[0];
            rel_out[1] = st[1];
        }
        s->it = (int64_t)ct[1];
        return GL_OK;
    }
    if (use_resident(s, maxit)) {
        double *st = (double *)ctx->pinned;
        int64_t ran = 0, total = 0;
        bool stopped = false;
        bool not_resident = false;
        // Converge mode on the tile engine, long calls: the rule (smooth_learning.py:166-167) needs BOTH ratios below tol, and
        // the dual one -- ||qbar - ybar|| / ||d||, n numbers that the owners of the node blocks form in passing -- is the one
        // that holds it up for most of a run (measured on the bench problem and its smaller siblings: above tol for 94 % of
        // the iterations to tol = 1e-5, while the primal ratio sits orders of magnitude below).  The primal norms cost a third
        // of the pass (three of nine fp64 operations per edge).  So a long call runs in up to three launches:
        //   head    the variant with the norms, GATE_HEAD iterations, judged as ever (it may stop the run);
        //   middle  if the head's last dual ratio is at least GATE_IN x tol: the variant WITHOUT the norms, which only
        //           watches the dual ratio and ends when it comes within GATE_OUT x tol (TileDev::gate) -- no iteration of
        //           it can meet the rule, so none has to be judged;
        //   tail    the variant with the norms to the end, every iteration judged.
        // The stopping iteration is the reference's as long as the dual ratio does not fall from GATE_OUT x tol (25 % above tol:
        // the threshold of the tile engine's lagged rule as well) to tol within the two iterations between the gate's look
        // and the tail's first judgement -- it decays like 1/k, k >= 16, and the largest drop per iteration seen where edges
        // switch on or off is 0.5 %.  GRAPHLEARN_AMD_TILE_GATE=0 switches the middle off.
        constexpr int64_t GATE_HEAD = 16, GATE_MIN_CALL = 512;
        constexpr double GATE_IN = 2.0, GATE_OUT = 1.25;
        const bool try_gate = s->tile != nullptr && tol > 0 && maxit >= GATE_MIN_CALL &&
                              !(getenv("GRAPHLEARN_AMD_TILE_GATE") && atoi(getenv("GRAPHLEARN_AMD_TILE_GATE")) == 0);
        int64_t left = maxit;
        GL_TRY(persistent_launch(s, try_gate ? GATE_HEAD : left, tol, true));
        GL_TRY(gl_resident_finish(s, &ran, &stopped, &not_resident));
        if (!not_resident) {
            total += ran, left -= ran;
            GL_HIP(ctx, hipMemcpyAsync(st, D.dstat, 2 * 8, hipMemcpyDeviceToHost, ctx->stream));
            GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (try_gate && !stopped && left > 0) {
                int phases = 1;
                // (st[1]: the dual ratio of the head's last iteration; a NaN or inf -- degenerate norms -- keeps the norms on)
                if (st[1] >= GATE_IN * tol && st[1] < std::numeric_limits<double>::infinity() && left > 2 * GATE_HEAD) {
                    bool gated = false, nr2 = false, st2 = false;
                    // (the tail keeps at least GATE_HEAD iterations of the budget: the ratios a call reports are those of
                    //  its last iteration, which only the variant with the norms forms)
                    GL_TRY(persistent_launch(s, left - GATE_HEAD, tol, false, -1, GATE_OUT));
                    GL_TRY(gl_resident_finish(s, &ran, &st2, &nr2, &gated));
                    if (nr2) return gl_set_error(ctx, GL_E_HIP, "tile engine: the workgroups of a continued call did not meet");
                    total += ran, left -= ran;
                    ++phases;
                }
                GL_TRY(persistent_launch(s, left, tol, true));
                GL_TRY(gl_resident_finish(s, &ran, &stopped, &not_resident));
                if (not_resident) return gl_set_error(ctx, GL_E_HIP, "tile engine: the workgroups of a continued call did not meet");
                total += ran, left -= ran;
                ++phases;
                GL_HIP(ctx, hipMemcpyAsync(st, D.dstat, 2 * 8, hipMemcpyDeviceToHost, ctx->stream));
                GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
                if (getenv("GRAPHLEARN_AMD_TILE_DEBUG"))
                    fprintf(stderr, "[tile gate] n %lld: %d launches, %lld iterations, stopped %d\n", (long long)s->n, phases, (long long)total, (int)stopped);
            }
            if (n_iter_out) *n_iter_out = total;
            if (converged_out) *converged_out = stopped ? 1 : 0;
            if (rel_out) {
                rel_out[0] = st[0];
                rel_out[1] = st[1];
            }
            return GL_OK;
        }
        persistent_free(s);  // shared GPU: this session stays on the streaming engine
    }
    const int64_t start = s->it;
    const int64_t end = start + maxit;
    double *st = (double *)ctx->pinned;          // [0..2]  dstat
    long long *ct = (long long *)(st + 8);       // [8..11] ctr
    bool done = false;
    st[0] = st[1] = std::numeric_limits<double>::infinity();
    while (!done && s->it < end) {
        const int64_t todo = std::min<int64_t>(ITER_BATCH, end - s->it);
        GL_TRY(launch_iterations(s, todo));
        k_status<<<dim3(1), dim3(64), 0, ctx->stream>>>(D);
        GL_HIP(ctx, hipGetLastError());
        GL_HIP(ctx, hipMemcpyAsync(st, D.dstat, 2 * 8, hipMemcpyDeviceToHost, ctx->stream));
        GL_HIP(ctx, hipMemcpyAsync(ct, D.ctr, 4 * 8, hipMemcpyDeviceToHost, ctx->stream));
        GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ct[0] != 0) {
            done = true;
            s->it = (int64_t)ct[1];
        }
    }
    if (done) {
        // re-arm the session: the flag belongs to this call's stopping rule
        GL_HIP(ctx, hipMemsetAsync(D.ctr, 0, 2 * 8, ctx->stream));
        GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (n_iter_out) *n_iter_out = s->it - start;
    if (converged_out) *converged_out = done ? 1 : 0;
    if (rel_out) {
        rel_out[0] = st[0];
        rel_out[1] = st[1];
    }
    return GL_OK;
}

GL_API int64_t gl_logdeg_total_iterations(gl_logdeg *s) { return s ? s->it : -1; }

// w_out == NULL: only produce the condensed vector in the session's device staging buffer
// (used by the RCCL gather, which sends it device-to-device).
GL_API int gl_logdeg_get_weights_condensed(gl_logdeg *s, double edge_tol, double *w_out) {
    GL_TRY(check(s));
    gl_ctx *ctx = s->ctx;
    if (!s->inited) return gl_set_error(ctx, GL_E_STATE, "no iterate yet: call init first");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    GL_TRY(resident_check(s));
    k_to_condensed<<<dim3((unsigned)s->dev.ntiles), dim3(256), 0, ctx->stream>>>(s->dev.w[s->it & 1], s->stage, s->n,
                                                                              s->dev.nb, edge_tol);
    GL_HIP(ctx, hipGetLastError());
    if (w_out)
        GL_TRY(gl_d2h(ctx, w_out, s->stage, (size_t)s->n_edges * 8));
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GL_OK;
}

// The thresholded iterates of B sessions (same context, same n) in one go: B conversions into one device
// buffer, ONE copy, ONE synchronisation (a batch of windows otherwise pays a round trip per graph).
// w_out is B x E, row b = session b (the layout of the reference's edge_w.T, smooth_learning.py:174).
namespace {
struct ThetaPtrs {
    const double *p[32];
};
__global__ void k_gather_thetas(ThetaPtrs q, int count, double *__restrict__ out) {
    if ((int)threadIdx.x < count) out[threadIdx.x] = q.p[threadIdx.x][0];
}
}  // namespace

GL_API int gl_logdeg_batch_get_weights(gl_logdeg **sessions, int B, double edge_tol, double *w_out) {
    if (!sessions || B < 1 || !w_out) return GL_E_INVALID;
    for (int a = 0; a < B; ++a) GL_TRY(check(sessions[a]));
    gl_ctx *ctx = sessions[0]->ctx;
    const int64_t E = sessions[0]->n_edges;
    for (int a = 0; a < B; ++a) {
        if (sessions[a]->ctx != ctx || sessions[a]->n != sessions[0]->n)
            return gl_set_error(ctx, GL_E_INVALID, "batched sessions must share the context and n_nodes");
        if (!sessions[a]->inited) return gl_set_error(ctx, GL_E_STATE, "no iterate yet: call init first");
    }
    GL_HIP(ctx, hipSetDevice(ctx->device));
    double *buf = nullptr;
    GL_TRY(gl_ctx_scratch(ctx, (size_t)B * E * 8, (void **)&buf));
    int rc = GL_OK;
    for (int a = 0; a < B && rc == GL_OK; ++a) {
        gl_logdeg *s = sessions[a];
        rc = resident_check(s);
        if (rc != GL_OK) break;
        k_to_condensed<<<dim3((unsigned)s->dev.ntiles), dim3(256), 0, ctx->stream>>>(s->dev.w[s->it & 1], buf + (size_t)a * E, s->n,
                                                                                  s->dev.nb, edge_tol);
    }
    // thetas that never left the device (set_signals with a deferred theta) come back with the same synchronisation: a batch of
    // windows otherwise pays a round trip per window for them afterwards (10 x 20 us for ten windows)
    constexpr size_t THETA_WORD0 = 4096;  // (word offset in the pinned status area: far behind the kernels' status words)
    const bool gather = rc == GL_OK && (size_t)B <= (GL_STATUS_BYTES / 8 - THETA_WORD0);
    for (int a0 = 0; gather && a0 < B; a0 += 32) {
        ThetaPtrs q{};
        const int cnt = std::min(32, B - a0);
        for (int a = 0; a < cnt; ++a) q.p[a] = sessions[a0 + a]->dev.dstat + 4;
        k_gather_thetas<<<1, 64, 0, ctx->stream>>>(q, cnt, (double *)ctx->pinned_dev + THETA_WORD0 + a0);
    }
    if (rc == GL_OK && hipGetLastError() != hipSuccess) rc = gl_set_error(ctx, GL_E_HIP, "k_to_condensed launch failed");
    if (rc == GL_OK) rc = gl_d2h(ctx, w_out, buf, (size_t)B * E * 8);
    (void)hipStreamSynchronize(ctx->stream);
    if (rc == GL_OK && gather)
        for (int a = 0; a < B; ++a)
            if (sessions[a]->theta_pending) {
                sessions[a]->theta = ((const double *)ctx->pinned)[THETA_WORD0 + a];
                sessions[a]->theta_pending = false;
            }
    return rc;
}

GL_API int gl_logdeg_get_weights_square(gl_logdeg *s, double edge_tol, double *w_out) {
    GL_TRY(check(s));
    gl_ctx *ctx = s->ctx;
    if (!w_out) return gl_set_error(ctx, GL_E_INVALID, "w_out is NULL");
    if (!s->inited) return gl_set_error(ctx, GL_E_STATE, "no iterate yet: call init first");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    GL_TRY(resident_check(s));
    // the square result (n^2) is staged in session-owned scratch (allocated at the first use, then kept)
    if (!s->square_out) GL_MALLOC(ctx, (void **)&s->square_out, (size_t)s->n * s->n * 8, "logdeg");
    k_to_square<<<dim3((unsigned)s->dev.ntiles), dim3(256), 0, ctx->stream>>>(s->dev.w[s->it & 1], s->square_out, s->n,
                                                                           s->dev.nb, edge_tol);
    GL_HIP(ctx, hipGetLastError());
    return gl_d2h(ctx, w_out, s->square_out, (size_t)s->n * s->n * 8);  // returns with the data in place
}

GL_API int gl_logdeg_get_dual(gl_logdeg *s, double *d_out) {
    GL_TRY(check(s));
    gl_ctx *ctx = s->ctx;
    if (!d_out) return gl_set_error(ctx, GL_E_INVALID, "d_out is NULL");
    if (!s->inited) return gl_set_error(ctx, GL_E_STATE, "no iterate yet: call init first");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    GL_TRY(resident_check(s));
    GL_TRY(gl_d2h(ctx, d_out, s->dev.dvar, (size_t)s->n * 8));
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GL_OK;
}

// ---------------------------------------------------------------------------
// One-shot forms.
GL_API int gl_log_degrees_fit(gl_ctx *ctx, const double *z_condensed, int64_t n_nodes, double alpha, double beta,
                              const double *edge_init, int64_t maxit, double tol, double step_size, double edge_tol,
                              double *w_out_condensed, int64_t *n_iter_out, double *rel_out) {
    if (!ctx) return GL_E_INVALID;
    gl_logdeg *s = nullptr;
    GL_TRY(gl_logdeg_create(ctx, n_nodes, &s));
    int rc = gl_logdeg_set_distances(s, z_condensed);
    if (rc == GL_OK) rc = gl_logdeg_init(s, alpha, beta, step_size, edge_init);
    if (rc == GL_OK) rc = gl_logdeg_iterate(s, maxit, tol, n_iter_out, nullptr, rel_out);
    if (rc == GL_OK) rc = gl_logdeg_get_weights_condensed(s, edge_tol, w_out_condensed);
    gl_logdeg_destroy(s);
    return rc;
}

GL_API int gl_logmodel_fit_signals(gl_ctx *ctx, const double *x, int64_t d, int64_t n_nodes, int avg_degree,
                                   const double *edge_init, int64_t maxit, double tol, double step_size,
                                   double edge_tol, double *weights_out, double *theta_out, int64_t *n_iter_out,
                                   double *rel_out) {
    if (!ctx) return GL_E_INVALID;
    gl_logdeg *s = nullptr;
    GL_TRY(gl_logdeg_create(ctx, n_nodes, &s));
    int rc = gl_logdeg_set_signals(s, x, d, n_nodes, avg_degree, theta_out);
    if (rc == GL_OK) rc = gl_logdeg_init(s, 1.0, 1.0, step_size, edge_init);  // alpha = beta = 1 (:291-292)
    if (rc == GL_OK) rc = gl_logdeg_iterate(s, maxit, tol, n_iter_out, nullptr, rel_out);
    if (rc == GL_OK) rc = gl_logdeg_get_weights_square(s, edge_tol, weights_out);
    gl_logdeg_destroy(s);
    return rc;
}

// Gram matrices of the last iteration across B sessions (same context, same n):
// out is 4 x B x B (row-major): [dw.dw, w_old.w_old, dd.dd, d_old.d_old].
GL_API int gl_logdeg_batch_gram(gl_logdeg **sessions, int B, double *out) {
    if (!sessions || B < 1 || !out) return GL_E_INVALID;
    for (int a = 0; a < B; ++a) GL_TRY(check(sessions[a]));
    gl_ctx *ctx = sessions[0]->ctx;
    for (int a = 0; a < B; ++a) {
        if (sessions[a]->ctx != ctx || sessions[a]->n != sessions[0]->n)
            return gl_set_error(ctx, GL_E_INVALID, "batched sessions must share the context and n_nodes");
        if (!sessions[a]->inited || sessions[a]->it < 1 || sessions[a]->it != sessions[0]->it)
            return gl_set_error(ctx, GL_E_STATE, "batched sessions must have run the same number (>= 1) of iterations");
    }
    GL_HIP(ctx, hipSetDevice(ctx->device));
    const LogdegDev &D0 = sessions[0]->dev;
    const int64_t n_tiled = D0.ntiles * GL_TILE_ELEMS;
    const int blocks = (int)std::min<int64_t>(512, (n_tiled + 255) / 256);
    const int npairs = B * (B + 1) / 2;
    double *part = nullptr;
    GL_TRY(gl_ctx_scratch(ctx, (size_t)npairs * blocks * 4 * 8, (void **)&part));
    int pi = 0;
    for (int a = 0; a < B; ++a)
        for (int b = a; b < B; ++b, ++pi) {
            const gl_logdeg *sa = sessions[a], *sb = sessions[b];
            const int64_t it = sa->it;
            k_pair_dots<<<blocks, 256, 0, ctx->stream>>>(sa->dev.w[it & 1], sa->dev.w[(it + 1) & 1], sb->dev.w[it & 1],
                                                         sb->dev.w[(it + 1) & 1], sa->dev.dvar, sa->dev.dprev,
                                                         sb->dev.dvar, sb->dev.dprev, n_tiled, sa->n,
                                                         part + (size_t)pi * blocks * 4);
        }
    std::vector<double> host((size_t)npairs * blocks * 4);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(host.data(), part, host.size() * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return gl_set_error(ctx, GL_E_HIP, "batch_gram: %s", hipGetErrorString(e));
    pi = 0;
    for (int a = 0; a < B; ++a)
        for (int b = a; b < B; ++b, ++pi)
            for (int q = 0; q < 4; ++q) {
                double acc = 0;
                for (int k = 0; k < blocks; ++k) acc += host[((size_t)pi * blocks + k) * 4 + q];
                out[((size_t)q * B + a) * B + b] = acc;
                out[((size_t)q * B + b) * B + a] = acc;
            }
    return GL_OK;
}

namespace {
// largest eigenvalue of a symmetric B x B matrix (cyclic Jacobi, in place); the matrices here are
// Gram matrices, and np.linalg.norm(M, 2)^2 of the reference (:158-159) is exactly this value
double sym_lambda_max(std::vector<double> &a, int B) {
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, diag = 0;
        for (int p = 0; p < B; ++p)
            for (int q = 0; q < B; ++q) (p == q ? diag : off) += a[(size_t)p * B + q] * a[(size_t)p * B + q];
        if (off <= 1e-32 * diag || off == 0.0) break;
        for (int p = 0; p < B - 1; ++p)
            for (int q = p + 1; q < B; ++q) {
                const double apq = a[(size_t)p * B + q];
                if (apq == 0.0) continue;
                const double theta = (a[(size_t)q * B + q] - a[(size_t)p * B + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < B; ++k) {  // columns p, q
                    const double akp = a[(size_t)k * B + p], akq = a[(size_t)k * B + q];
                    a[(size_t)k * B + p] = c * akp - sn * akq;
                    a[(size_t)k * B + q] = sn * akp + c * akq;
                }
                for (int k = 0; k < B; ++k) {  // rows p, q
                    const double apk = a[(size_t)p * B + k], aqk = a[(size_t)q * B + k];
                    a[(size_t)p * B + k] = c * apk - sn * aqk;
                    a[(size_t)q * B + k] = sn * apk + c * aqk;
                }
            }
    }
    double m = a[0];
    for (int p = 1; p < B; ++p) m = std::max(m, a[(size_t)p * B + p]);
    return m;
}
}  // namespace

// ---- the batch's cruise on the tile engine -----------------------------------------------------------------------------
// While the coupled rule is out of reach the B graphs advance as ONE block-diagonal lattice in one persistent launch of the
// tile engine (gl_tile.hip, k_tile<..., BATCH>: 5 us per iteration where the two batched launches take 18): a stand-in
// session holds the batch's state graph after graph, the sessions' buffers are copied in before and out after (their tiled
// layouts agree, so these are plain copies), and the launch ends at the first iteration whose screen no longer rules the
// rule out with 5 % to spare -- from there the chunked launches below take over, exact spectral norms and all.
struct BatchCruise {
    gl_logdeg cs;
    int B = 0;
    int64_t n = 0;
    void *arena = nullptr;
    // the deciding chunks (below): the ring of iterates, the dual variable beside it, a checkpoint of the rest of the dual
    // state, the Gram partials and the per-iteration decisions -- one allocation, made when a call first gets there
    void *ring_arena = nullptr;
    int ring_k = 0;  // iterations per chunk
    double *ring = nullptr, *dring = nullptr, *gpart = nullptr;
    RingJudge *judge = nullptr;
};
static void batch_cruise_free(void *p) {
    BatchCruise *bc = (BatchCruise *)p;
    if (!bc) return;
    gl_tile_free(&bc->cs);
    if (bc->arena) (void)hipFree(bc->arena);
    if (bc->ring_arena) (void)hipFree(bc->ring_arena);
    delete bc;
}
namespace {
// dir 0: sessions -> stand-in (iterate of parity par, dual state); dir 1: back, and the sessions' counters stand at `itf`
// wold (dir 1 only): the iterate before the last one, graph after graph -- it goes to the sessions' other ping-pong buffer
__global__ __launch_bounds__(256) void k_batch_copy_state(const LogdegDev *__restrict__ devs, LogdegDev C, int64_t tiled1, int64_t npad1,
                                                          int64_t nb1, int par, int dir, long long itf, const double *__restrict__ wold = nullptr) {
    const LogdegDev S = devs[blockIdx.y];
    const int64_t g = blockIdx.y, stride = (int64_t)gridDim.x * 256, t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double *cw = C.w[par] + g * tiled1, *sw = S.w[par];
    for (int64_t e = t0; e < tiled1; e += stride) {
        if (dir == 0) cw[e] = sw[e];
        else sw[e] = cw[e];
    }
    if (dir == 1 && wold) {
        double *so = S.w[par ^ 1];
        for (int64_t e = t0; e < tiled1; e += stride) so[e] = wold[g * tiled1 + e];
    }
    for (int64_t i = t0; i < npad1; i += stride) {
        const int64_t k = g * npad1 + i;
        if (dir == 0) {
            C.dvar[k] = S.dvar[i], C.ybar[k] = S.ybar[i], C.pbar[k] = S.pbar[i], C.dprev[k] = S.dprev[i];
        } else {
            S.dvar[i] = C.dvar[k], S.ybar[i] = C.ybar[k], S.pbar[i] = C.pbar[k], S.dprev[i] = C.dprev[k];
        }
    }
    if (dir == 1) {
        for (int64_t i = t0; i < 2 * nb1 * 4; i += stride) {  // [parity][block][4]
            const int64_t pr = i / (nb1 * 4), r = i % (nb1 * 4);
            S.bnorm[i] = C.bnorm[pr * C.nb * 4 + g * nb1 * 4 + r];
        }
        if (t0 == 0) {
            S.ctr[0] = 0, S.ctr[1] = 0, S.ctr[2] = itf, S.ctr[3] = itf > 0 ? itf - 1 : 0, S.ctr[4] = 0;
        }
    }
}
}  // namespace

// the stand-in session for (n, B) on this context: built once, kept (GL_OK with *out == nullptr: the engine does not apply)
static int batch_cruise_get(gl_ctx *ctx, gl_logdeg **sessions, int B, BatchCruise **out) {
    *out = nullptr;
    const int64_t n = sessions[0]->n;
    BatchCruise *bc = (BatchCruise *)ctx->batch_cruise;
    if (bc && (bc->B != B || bc->n != n)) {
        GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        batch_cruise_free(bc);
        ctx->batch_cruise = bc = nullptr;
    }
    if (!bc) {
        bc = new (std::nothrow) BatchCruise();
        if (!bc) return GL_E_NOMEM;
        bc->B = B;
        bc->n = n;
        ctx->batch_cruise = bc;
        ctx->batch_cruise_free = batch_cruise_free;
        gl_logdeg &cs = bc->cs;
        cs.ctx = ctx;
        cs.n = n;
        cs.n_edges = n * (n - 1) / 2;
        LogdegDev &D = cs.dev;
        const LogdegDev &D0 = sessions[0]->dev;
        D.n = n;
        D.nb = D0.nb * B;
        D.n_pad = D.nb * GL_TB;
        D.ntiles = D0.ntiles * B;
        D.npairs = D.ntiles * PAIRS_PER_TILE;
        const size_t tiled = (size_t)D.ntiles * GL_TILE_ELEMS * 8, nodes = (size_t)D.n_pad * 8;
        struct Part {
            void **p;
            size_t bytes;
        };
        const Part parts[] = {{(void **)&D.w[0], tiled},  {(void **)&D.w[1], tiled},  {(void **)&D.dvar, nodes},
                              {(void **)&D.ybar, nodes},  {(void **)&D.pbar, nodes},  {(void **)&D.dprev, nodes},
                              {(void **)&D.bnorm, (size_t)2 * D.nb * 4 * 8}, {(void **)&D.dstat, 8 * 8}, {(void **)&D.ctr, 8 * 8}};
        size_t total = 0;
        for (const Part &q : parts) total += (q.bytes + 255) / 256 * 256;
        // No room for the B-fold stand-in: the batched launches serve the batch (their buffers are the sessions' own).  Nothing
        // half-built stays behind -- the next call tries again --, and the runtime's last-error word is clean (gl_ctx_malloc),
        // so that the launch path's hipGetLastError() checks do not report this allocation as their own failure.
        ctx->batch_cruise_busy = true;  // (gl_ctx_malloc may trim what the context retains: not this half-built entry)
        const hipError_t ea = gl_ctx_malloc(ctx, &bc->arena, total, "cruise");
        ctx->batch_cruise_busy = false;
        if (ea != hipSuccess) {
            batch_cruise_free(bc);
            ctx->batch_cruise = nullptr;
            return GL_OK;
        }
        char *at = (char *)bc->arena;
        for (const Part &q : parts) {
            *q.p = at;
            at += (q.bytes + 255) / 256 * 256;
        }
        GL_HIP(ctx, hipMemsetAsync(bc->arena, 0, total, ctx->stream));
        GL_TRY(gl_tile_prepare_batch(&cs, B));
        if (!cs.tile) {  // the batch does not fit the chip: the entry stays (so that the next call does not try again) without its arena
            GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
            (void)hipFree(bc->arena);
            bc->arena = nullptr;
        }
    }
    if (bc->cs.tile && bc->arena) *out = bc;
    return GL_OK;
}

// The 2-D (batched) branch of gsp_learn_graph_log_degrees (smooth_learning.py:128-129,146-167): B
// graphs advance in lockstep and ALL stop at the iteration where both spectral-norm ratios
//   ||Q - Y||_2 / ||W_old||_2  and  ||Qbar - Ybar||_2 / ||D_old||_2      ((E x B) and (n x B) matrices)
// fall below tol.  The whole loop runs here.  One iteration of all graphs (two batched launches,
// blockIdx.y = graph) + the B diagonal Gram
// entries + their copy to pinned memory are captured once per parity of the iteration count and
// replayed (one graph launch and one sync per iteration).  lambda_max of a Gram matrix lies in
// [max diagonal, trace], which decides most iterations without the off-diagonal inner products --
// those (one launch for all pairs) and the B x B eigenproblems (Jacobi, host) are computed only when
// the bounds straddle tol and for the values reported at the end.  (Per-session launches cost
// ~12 us per graph and iteration; a fork / join hipGraph with one branch per session was 6x
// slower still on this runtime.)  Sessions must
// share the context and n, be initialised and have run the same number of iterations.
GL_API int gl_logdeg_batch_iterate(gl_logdeg **sessions, int B, int64_t maxit, double tol, int64_t *n_iter_out,
                                   int *converged_out, double *rel_out) {
    if (!sessions || B < 1) return GL_E_INVALID;
    for (int a = 0; a < B; ++a) GL_TRY(check(sessions[a]));
    gl_ctx *ctx = sessions[0]->ctx;
    if (maxit < 0) return gl_set_error(ctx, GL_E_INVALID, "maxit < 0");
    for (int a = 0; a < B; ++a) {
        gl_logdeg *s = sessions[a];
        if (s->ctx != ctx || s->n != sessions[0]->n)
            return gl_set_error(ctx, GL_E_INVALID, "batched sessions must share the context and n_nodes");
        if (!s->inited || s->it != sessions[0]->it)
            return gl_set_error(ctx, GL_E_STATE, "batched sessions must be initialised and at the same iteration");
    }
    GL_HIP(ctx, hipSetDevice(ctx->device));
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int a = 0; a < B; ++a) {
        GL_TRY(resident_check(sessions[a]));
        GL_TRY(set_tol(sessions[a], 0.0));  // the per-graph rule never fires; the coupled rule is evaluated here
    }
    const LogdegDev &D0 = sessions[0]->dev;
    const int64_t n_tiled = D0.ntiles * GL_TILE_ELEMS;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(64, (n_tiled + 2047) / 2048));
    const int npairs = B * (B + 1) / 2;
    // pair table: the B diagonal pairs first, then the strict upper triangle
    std::vector<int2> pairs;
    for (int a = 0; a < B; ++a) pairs.push_back(int2{a, a});
    for (int a = 0; a < B; ++a)
        for (int b = a + 1; b < B; ++b) pairs.push_back(int2{a, b});
    std::vector<BatchPtrs> ptrs((size_t)2 * B);  // [parity of the iteration count after the step][session]
    for (int par = 0; par < 2; ++par)
        for (int a = 0; a < B; ++a) {
            const LogdegDev &D = sessions[a]->dev;
            ptrs[(size_t)par * B + a] = BatchPtrs{D.w[par], D.w[par ^ 1], D.dvar, D.dprev};
        }
    // one carve-up of the context's scratch (kept across calls: no hipMalloc / hipFree here)
    struct Scratch {
        void *pairs = nullptr, *ptrs = nullptr, *part = nullptr, *out = nullptr, *devs = nullptr, *small = nullptr;
    } sc;
    {
        auto up = [](size_t b) { return (b + 255) / 256 * 256; };
        const bool small_path = maxit >= 8 && gl_small_applicable(sessions[0]) && B <= ctx->prop.multiProcessorCount;
        const size_t b_pairs = up((size_t)npairs * sizeof(int2)), b_ptrs = up((size_t)2 * B * sizeof(BatchPtrs));
        // (pair partials of k_batch_dots, or the 16 x 16 block partials of k_batch_gram: up to 257 workgroups x 6 blocks)
        const size_t b_part = 256 + std::max(up((size_t)npairs * blocks * 4 * 8), (size_t)257 * 6 * 256 * 8), b_out = up((size_t)npairs * 4 * 8);
        const size_t b_devs = up((size_t)B * sizeof(LogdegDev));
        const size_t b_small = small_path ? up(gl_small_batch_scratch_bytes(B, (int)sessions[0]->n)) : 0;
        char *at = nullptr;
        GL_TRY(gl_ctx_scratch(ctx, b_pairs + b_ptrs + b_part + b_out + b_devs + b_small, (void **)&at));
        sc.pairs = at, at += b_pairs;
        sc.ptrs = at, at += b_ptrs;
        sc.part = at, at += b_part;
        sc.out = at, at += b_out;
        sc.devs = at, at += b_devs;
        sc.small = at;
    }
    GL_HIP(ctx, hipMemcpyAsync(sc.pairs, pairs.data(), pairs.size() * sizeof(int2), hipMemcpyHostToDevice, ctx->stream));
    GL_HIP(ctx, hipMemcpyAsync(sc.ptrs, ptrs.data(), ptrs.size() * sizeof(BatchPtrs), hipMemcpyHostToDevice, ctx->stream));
    std::vector<LogdegDev> devs((size_t)B);  // the problems' device descriptors, for the batched launches
    for (int a = 0; a < B; ++a) devs[(size_t)a] = sessions[a]->dev;
    GL_HIP(ctx, hipMemcpyAsync(sc.devs, devs.data(), devs.size() * sizeof(LogdegDev), hipMemcpyHostToDevice, ctx->stream));
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<double> host((size_t)npairs * 4);

    auto exact = [&](double &rp, double &rd) -> int {  // spectral-norm ratios from the full Gram matrices
        const int64_t it = sessions[0]->it;
        k_batch_dots<<<dim3((unsigned)npairs, (unsigned)blocks), 256, 0, ctx->stream>>>(
            (const BatchPtrs *)sc.ptrs + (size_t)(it & 1) * B, (const int2 *)sc.pairs, n_tiled, sessions[0]->n,
            (double *)sc.part);
        k_batch_dots_reduce<<<dim3((unsigned)((npairs * 4 + 255) / 256)), 256, 0, ctx->stream>>>((const double *)sc.part,
                                                                                           npairs, blocks, (double *)sc.out);
        GL_HIP(ctx, hipGetLastError());
        GL_HIP(ctx, hipMemcpyAsync(host.data(), sc.out, (size_t)npairs * 4 * 8, hipMemcpyDeviceToHost, ctx->stream));
        GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        double lam[4];
        std::vector<double> g((size_t)B * B);
        for (int q = 0; q < 4; ++q) {
            for (int k = 0; k < npairs; ++k) {
                const int2 pr = pairs[(size_t)k];
                g[(size_t)pr.x * B + pr.y] = g[(size_t)pr.y * B + pr.x] = host[(size_t)k * 4 + q];
            }
            lam[q] = std::max(sym_lambda_max(g, B), 0.0);
        }
        rp = sqrt(lam[0]) / sqrt(lam[1]);  // inf / nan when the previous iterate is zero, like NumPy
        rd = sqrt(lam[2]) / sqrt(lam[3]);
        return GL_OK;
    };

    // Small graphs (n <= 128): one persistent launch, one workgroup per graph, the norm bounds evaluated
    // on the device (gl_small.hip); the host only steps in when the exact spectral norms are needed.
    if (maxit >= 8 && gl_small_applicable(sessions[0]) && B <= ctx->prop.multiProcessorCount) {
        void *small_scratch = sc.small;
        double rp = std::numeric_limits<double>::infinity(), rd = rp;
        bool have_exact = false, converged = false, small_failed = false;
        int64_t n_iter = 0;
        while (n_iter < maxit) {
            int64_t ran = 0;
            bool need_exact = false, stopped = false;
            const int rc = gl_small_batch_run(sessions, B, maxit - n_iter, tol, small_scratch, &ran, &need_exact, &stopped);
            if (rc == GL_E_STATE && n_iter == 0) {  // workgroups not co-resident: per-iteration launches instead
                small_failed = true;
                break;
            }
            if (rc != GL_OK) return rc;
            n_iter += ran;
            if (ran > 0) have_exact = false;
            if (stopped) {  // the rule (:166-167) holds for every value the four spectral norms can take (device brackets)
                converged = true;
                break;
            }
            if (!need_exact) break;  // maxit reached
            GL_TRY(exact(rp, rd));
            have_exact = true;
            if (rp < tol && rd < tol) {  // :166-167, after the update
                converged = true;
                break;
            }
        }
        if (!small_failed) {
            if (!have_exact && n_iter > 1) GL_TRY(exact(rp, rd));
            if (n_iter_out) *n_iter_out = n_iter;
            if (converged_out) *converged_out = converged ? 1 : 0;
            if (rel_out) {
                rel_out[0] = rp;
                rel_out[1] = rd;
            }
            return GL_OK;
        }
    }

    // Larger graphs: two batched launches per iteration (edge pass, node pass; blockIdx.y = graph), enqueued a CHUNK of
    // iterations at a time as one captured graph.  The screen of the coupled rule runs on the device, in the node pass's
    // last workgroup (k_node_batch): while it can rule the rule out nothing stops; when it cannot -- or when the call's
    // budget is used up -- every problem's done flag is raised, the rest of the chunk falls through, and the host, which
    // looks once per chunk, forms the exact spectral norms for exactly that iterate.  (Round 2 synchronised after every
    // iteration: 45-57 us per iteration at n = 256, against ~10 us for the two launches.)
    if (ctx->stage_busy) {
        GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        gl_stage_idle(ctx);
    }
    // The cruise (above): everything but the call's last iteration at most -- the values reported at the end need the last
    // two iterates side by side, which only the launches below keep.
    int64_t cruised = 0;
    bool ring_finished = false;  // the deciding chunks (below) have brought the call to its end
    bool ring_exact = false;     // ... and ring_rp / ring_rd are the exact ratios of its last iteration
    double ring_rp = std::numeric_limits<double>::infinity(), ring_rd = ring_rp;
    {
        bool same = sessions[0]->it == 0 && maxit >= 64 && !getenv("GRAPHLEARN_AMD_BATCH_NO_CRUISE");
        for (int a = 1; a < B && same; ++a) {
            const LogdegDev &Da = sessions[a]->dev;
            same = Da.gamma == D0.gamma && Da.c == D0.c && Da.a4g == D0.a4g;
        }
        BatchCruise *bc = nullptr;
        const auto t_get0 = std::chrono::steady_clock::now();
        if (same) GL_TRY(batch_cruise_get(ctx, sessions, B, &bc));
        struct Busy {  // while this call works on the stand-in, an allocation that runs short must not trim it away
            gl_ctx *c;
            ~Busy() { c->batch_cruise_busy = false; }
        } busy{ctx};
        ctx->batch_cruise_busy = bc != nullptr;
        if (getenv("GRAPHLEARN_AMD_BATCH_DEBUG"))
            fprintf(stderr, "[batch setup] stand-in session: %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_get0).count());
        if (bc) {
            gl_logdeg &cs = bc->cs;
            LogdegDev &C = cs.dev;
            C.gamma = D0.gamma, C.c = D0.c, C.g2 = D0.g2, C.a4g = D0.a4g;
            cs.it = 0;
            cs.inited = true;
            ++cs.init_version;
            const int64_t tiled1 = D0.ntiles * GL_TILE_ELEMS;
            const dim3 cgrid((unsigned)std::min<int64_t>(64, (tiled1 + 2047) / 2048), (unsigned)B);
            GL_HIP(ctx, hipMemsetAsync(C.ctr, 0, 8 * 8, ctx->stream));
            k_batch_copy_state<<<cgrid, 256, 0, ctx->stream>>>((const LogdegDev *)sc.devs, C, tiled1, D0.n_pad, D0.nb, 0, 0, 0);
            GL_HIP(ctx, hipGetLastError());
            GL_TRY(gl_tile_batch_make_zg(&cs, (const LogdegDev *)sc.devs, B));
            const auto t_cruise0 = std::chrono::steady_clock::now();
            GL_TRY(gl_tile_launch(&cs, maxit - 1, tol, true, -1));
            int64_t ran = 0;
            bool stopped = false, not_resident = false;
            GL_TRY(gl_resident_finish(&cs, &ran, &stopped, &not_resident));
            bool keep = !not_resident && ran > 0;
            if (keep && stopped) {  // (dstat[0] = 1: the screen of the stopping iteration could not rule the rule out at all)
                double *slot = (double *)ctx->pinned;
                GL_HIP(ctx, hipMemcpyAsync(slot, C.dstat, 8, hipMemcpyDeviceToHost, ctx->stream));
                GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
                keep = slot[0] == 0.0;
            }
            if (getenv("GRAPHLEARN_AMD_BATCH_TEST_OVERSHOOT")) keep = false;  // (test hook: as if the screen had overshot)
            if (getenv("GRAPHLEARN_AMD_BATCH_DEBUG"))
                fprintf(stderr, "[batch cruise] n %lld B %d: ran %lld of %lld in %.2f ms, stopped %d, not resident %d, kept %d\n", (long long)sessions[0]->n, B,
                        (long long)ran, (long long)(maxit - 1), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_cruise0).count(),
                        (int)stopped, (int)not_resident, (int)keep);
            if (not_resident) {  // shared GPU: this context's batches stay on the launches below
                batch_cruise_free(bc);
                ctx->batch_cruise = nullptr;
            } else if (keep) {
                cruised = ran;
                // ---- The deciding chunks.  From here on the exact rule is needed, iteration by iteration -- and the tile engine
                // can still run them: K iterations per launch, every iterate written to a ring (gl_tile.hip: TileDev::ring), then
                // ONE launch judges all K at once (k_batch_gram_ring).  The first iteration whose rule is met (or tied) ends
                // the chunk: the state after it is rebuilt from the chunk's start -- the engine is deterministic -- by a launch
                // of exactly that many iterations.  ~10 us per iteration where the three launches per iteration below take 32-45.
                const int nbk = B <= 16 ? 1 : 3;
                const int64_t gram_units = tiled1 / 16;
                // (measured without effect: a workgroup count per judged iteration that is a multiple of 8, so that the two readers
                //  of a slot's stretch share an XCD's L2.  At B = 20 the judging launch takes ~5 us per iteration -- two workgroups
                //  per CU at 128 registers, 16 dependent load-multiply turns each --, at B = 8 1.7)
                const unsigned ge = (unsigned)std::max<int64_t>(1, std::min<int64_t>(128, gram_units / (4 * GRAM_WAVES)));
                bool ring_ok = B <= BATCH_BRACKET_MAXB && !getenv("GRAPHLEARN_AMD_BATCH_NO_RING");
                const auto t_ring0 = std::chrono::steady_clock::now();
                if (ring_ok && !bc->ring_arena) {
                    int K = getenv("GRAPHLEARN_AMD_BATCH_RING_K") ? std::max(2, std::min(64, atoi(getenv("GRAPHLEARN_AMD_BATCH_RING_K")))) : 64;
                    // (a slot of the dual ring: {d, ybar, pbar, dprev} of the whole lattice and the node blocks' four norm partials)
                    const size_t slot = (size_t)B * tiled1 * 8, dslot = (size_t)(4 * C.n_pad + 4 * C.nb) * 8;
                    while (K > 4 && (K + 1) * slot > ((size_t)3 << 30)) K /= 2;
                    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
                    const size_t b_ring = up((K + 1) * slot), b_dring = up((K + 1) * dslot);
                    const size_t b_part = up((size_t)K * (ge + 1) * 2 * nbk * 256 * 8), b_j = up(sizeof(RingJudge));
                    if (gl_ctx_malloc(ctx, &bc->ring_arena, b_ring + b_dring + b_part + b_j, "ring") == hipSuccess) {
                        char *at = (char *)bc->ring_arena;
                        bc->ring = (double *)at, at += b_ring;
                        bc->dring = (double *)at, at += b_dring;
                        bc->gpart = (double *)at, at += b_part;
                        bc->judge = (RingJudge *)at;
                        bc->ring_k = K;
                        GL_HIP(ctx, hipMemsetAsync(bc->judge, 0, sizeof(RingJudge), ctx->stream));
                    } else {  // (no room for the ring: the batched launches judge the rest of the run)
                        bc->ring_arena = nullptr;
                        ring_ok = false;
                    }
                }
                if (getenv("GRAPHLEARN_AMD_BATCH_DEBUG"))
                    fprintf(stderr, "[batch setup] ring arena: %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_ring0).count());
                int64_t n_done = ran;
                bool back_to_launches = !ring_ok;
                const double *w_old = nullptr;  // the iterate before the current one (a ring slot), once a chunk has run
                const size_t wbytes = (size_t)B * tiled1 * 8, dbytes = (size_t)C.n_pad * 8;
                const double t2r = tol * tol;
                int chunks = 0, replays = 0;
                const size_t dstride = (size_t)4 * C.n_pad + 4 * C.nb;  // doubles per slot of the dual ring
                const bool replay_hits = getenv("GRAPHLEARN_AMD_BATCH_REPLAY") != nullptr;
                double ms_run = 0, ms_judge = 0, ms_replay = 0, ms_exact = 0;  // (debug print below)
                auto now = []() { return std::chrono::steady_clock::now(); };
                auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
                // (chunks of 16, 32, then K = 64 iterations: a short deciding stretch wastes little behind its stopping
                //  iteration, a long one spreads a chunk's fixed cost -- ~150 us -- over many)
                int kcur = getenv("GRAPHLEARN_AMD_BATCH_RING_K") ? bc->ring_k : std::min(16, bc->ring_k);
                while (ring_ok && !back_to_launches && n_done < maxit) {
                    const int Kc = (int)std::min<int64_t>(kcur, maxit - n_done);
                    kcur = std::min(bc->ring_k, 2 * kcur);
                    const int64_t it_ck = cs.it;
                    auto t_ph = now();
                    GL_HIP(ctx, hipMemcpyAsync(bc->ring, C.w[it_ck & 1], wbytes, hipMemcpyDeviceToDevice, ctx->stream));
                    // the dual state of ring slot q -> the stand-in session (its norm partials belong to iteration `itn - 1`); slot 0 is
                    // the chunk's start, slot q the state after q of its iterations
                    auto dual_from_slot = [&](int q, int64_t itn, bool with_norms) -> int {
                        const double *dr = bc->dring + (size_t)q * dstride;
                        GL_HIP(ctx, hipMemcpyAsync(C.dvar, dr, dbytes, hipMemcpyDeviceToDevice, ctx->stream));
                        GL_HIP(ctx, hipMemcpyAsync(C.ybar, dr + C.n_pad, dbytes, hipMemcpyDeviceToDevice, ctx->stream));
                        GL_HIP(ctx, hipMemcpyAsync(C.pbar, dr + 2 * C.n_pad, dbytes, hipMemcpyDeviceToDevice, ctx->stream));
                        GL_HIP(ctx, hipMemcpyAsync(C.dprev, dr + 3 * C.n_pad, dbytes, hipMemcpyDeviceToDevice, ctx->stream));
                        if (with_norms)
                            GL_HIP(ctx, hipMemcpyAsync(C.bnorm + ((itn - 1) & 1) * 4 * C.nb, dr + 4 * C.n_pad, (size_t)4 * C.nb * 8, hipMemcpyDeviceToDevice,
                                                       ctx->stream));
                        return GL_OK;
                    };
                    GL_HIP(ctx, hipMemcpyAsync(bc->dring, C.dvar, dbytes, hipMemcpyDeviceToDevice, ctx->stream));
                    GL_HIP(ctx, hipMemcpyAsync(bc->dring + C.n_pad, C.ybar, dbytes, hipMemcpyDeviceToDevice, ctx->stream));
                    GL_HIP(ctx, hipMemcpyAsync(bc->dring + 2 * C.n_pad, C.pbar, dbytes, hipMemcpyDeviceToDevice, ctx->stream));
                    GL_HIP(ctx, hipMemcpyAsync(bc->dring + 3 * C.n_pad, C.dprev, dbytes, hipMemcpyDeviceToDevice, ctx->stream));
                    gl_tile_batch_set_ring(&cs, bc->ring, (int64_t)B * tiled1, bc->dring, (int64_t)dstride);
                    GL_TRY(gl_tile_launch(&cs, Kc, 0.0, true, -1));
                    gl_tile_batch_set_ring(&cs, nullptr, 0, nullptr, 0);
                    int64_t got = 0;
                    bool st2 = false, nr2 = false;
                    GL_TRY(gl_resident_finish(&cs, &got, &st2, &nr2));
                    auto restore = [&]() -> int {  // the chunk's start
                        cs.it = it_ck;
                        GL_HIP(ctx, hipMemcpyAsync(C.w[it_ck & 1], bc->ring, wbytes, hipMemcpyDeviceToDevice, ctx->stream));
                        return dual_from_slot(0, it_ck, false);
                    };
                    // (the workgroups did not meet: nothing was written; the launches below go on from here.  Test hook: as if)
                    if (nr2 || got != Kc || (chunks == 1 && getenv("GRAPHLEARN_AMD_BATCH_TEST_RING_FAIL"))) {
                        GL_TRY(restore());
                        back_to_launches = true;
                        break;
                    }
                    ++chunks;
                    ms_run += since(t_ph), t_ph = now();
                    const dim3 gg(ge + 1, (unsigned)Kc);
                    if (B <= 16)
                        k_batch_gram_ring<1><<<gg, GRAM_WAVES * 64, 0, ctx->stream>>>(bc->ring, bc->dring, B, (int64_t)B * tiled1, tiled1, (int64_t)dstride, D0.n_pad,
                                                                                      bc->gpart, t2r, bc->judge);
                    else
                        k_batch_gram_ring<2><<<gg, GRAM_WAVES * 64, 0, ctx->stream>>>(bc->ring, bc->dring, B, (int64_t)B * tiled1, tiled1, (int64_t)dstride, D0.n_pad,
                                                                                      bc->gpart, t2r, bc->judge);
                    GL_HIP(ctx, hipGetLastError());
                    RingJudge *jh = (RingJudge *)((char *)ctx->pinned + GL_STATUS_BYTES);
                    GL_HIP(ctx, hipMemcpyAsync(jh, bc->judge, sizeof(RingJudge), hipMemcpyDeviceToHost, ctx->stream));
                    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
                    int hit = -1;
                    for (int j = 0; j < Kc && hit < 0; ++j)
                        if (jh->decision[j] != 0) hit = j;
                    ms_judge += since(t_ph), t_ph = now();
                    if (hit < 0) {  // the whole chunk was short of the rule
                        n_done += Kc;
                        w_old = bc->ring + (size_t)(Kc - 1) * B * tiled1;
                        continue;
                    }
                    if (hit + 1 < Kc && !replay_hits) {
                        // The state after iteration `hit` is in the rings (round 4: the iterate always was; the dual state -- ybar, pbar,
                        // dprev, the norm partials beside d -- was not, and a launch of hit + 1 iterations from the chunk's start rebuilt
                        // it: 54 iterations, 0.36 ms of a 6.5 ms WindowLogModel fit).  GRAPHLEARN_AMD_BATCH_REPLAY=1: the old way.
                        cs.it = it_ck + hit + 1;
                        GL_HIP(ctx, hipMemcpyAsync(C.w[cs.it & 1], bc->ring + (size_t)(hit + 1) * B * tiled1, wbytes, hipMemcpyDeviceToDevice, ctx->stream));
                        GL_TRY(dual_from_slot(hit + 1, cs.it, true));
                    } else if (hit + 1 < Kc) {  // the state after iteration `hit`: from the chunk's start, hit + 1 iterations
                        GL_TRY(restore());
                        GL_TRY(gl_tile_launch(&cs, hit + 1, 0.0, true, -1));
                        GL_TRY(gl_resident_finish(&cs, &got, &st2, &nr2));
                        ++replays;
                        if (nr2 || got != hit + 1) {
                            GL_TRY(restore());
                            back_to_launches = true;
                            break;
                        }
                    }
                    n_done += hit + 1;
                    ms_replay += since(t_ph), t_ph = now();
                    w_old = bc->ring + (size_t)hit * B * tiled1;
                    // the sessions get this state (both iterates), the host forms the exact ratios: met -> done; a tie that was
                    // not met after all -> the chunks go on from here
                    k_batch_copy_state<<<cgrid, 256, 0, ctx->stream>>>((const LogdegDev *)sc.devs, C, tiled1, D0.n_pad, D0.nb, (int)(n_done & 1), 1,
                                                                       (long long)n_done, w_old);
                    GL_HIP(ctx, hipGetLastError());
                    for (int a = 0; a < B; ++a) sessions[a]->it = n_done;
                    GL_TRY(exact(ring_rp, ring_rd));
                    ring_exact = true;
                    ms_exact += since(t_ph);
                    if (jh->decision[hit] == 3 || (ring_rp < tol && ring_rd < tol)) {
                        ring_finished = true;
                        break;
                    }
                    ring_exact = false;  // (a tie that was not met: the chunks go on, these ratios go stale)
                }
                if (getenv("GRAPHLEARN_AMD_BATCH_DEBUG") && ring_ok)
                    fprintf(stderr, "[batch ring] %d chunks of up to %d iterations, %d replays, at iteration %lld, finished %d, back to the launches %d; "
                            "ms: chunks' launches %.2f, judging %.2f, replays %.2f, exact ratios %.2f\n", chunks,
                            bc->ring_k, replays, (long long)n_done, (int)ring_finished, (int)back_to_launches, ms_run, ms_judge, ms_replay, ms_exact);
                if (!ring_finished) {  // the budget ran out, or the ring is not available: the sessions take over where the engine stands
                    k_batch_copy_state<<<cgrid, 256, 0, ctx->stream>>>((const LogdegDev *)sc.devs, C, tiled1, D0.n_pad, D0.nb, (int)(n_done & 1), 1,
                                                                       (long long)n_done, w_old);
                    GL_HIP(ctx, hipGetLastError());
                    for (int a = 0; a < B; ++a) sessions[a]->it = n_done;
                    if (ring_ok && !back_to_launches) ring_finished = true;  // (n_done == maxit)
                }
                cruised = n_done;
                // a ring beyond half a GiB is given back at the end of the call (the next long batch pays ~1 ms for a new one;
                // a context that keeps 3 GiB for the sake of that millisecond starves its neighbours on a shared device)
                if (bc->ring_arena && getenv("GRAPHLEARN_AMD_BATCH_KEEP_RING") == nullptr) {
                    size_t ring_bytes = (size_t)(bc->ring_k + 1) * wbytes;
                    if (ring_bytes > ((size_t)512 << 20)) {
                        GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
                        (void)hipFree(bc->ring_arena);
                        bc->ring_arena = nullptr;
                        bc->ring = bc->dring = bc->gpart = nullptr;
                        bc->judge = nullptr;
                        bc->ring_k = 0;
                    }
                }
            }
        }
    }
    BatchState *st_dev = (BatchState *)sc.part;  // (the partials of the exact evaluation live behind it)
    sc.part = (char *)sc.part + 256;
    BatchState *st_host = (BatchState *)((char *)ctx->pinned + GL_STATUS_BYTES);
    {
        BatchState init{};
        init.first = cruised == 0 ? 1 : 0;
        init.ran = (int)cruised;
        init.remaining = (int)std::min<int64_t>(maxit - cruised, 1 << 30);
        *st_host = init;
        GL_HIP(ctx, hipMemcpyAsync(st_dev, st_host, sizeof(BatchState), hipMemcpyHostToDevice, ctx->stream));
        GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    const double t2 = tol * tol;
    const int chunk = (int)std::min<int64_t>(maxit, getenv("GRAPHLEARN_AMD_BATCH_CHUNK") ? std::max(1, atoi(getenv("GRAPHLEARN_AMD_BATCH_CHUNK"))) : 16);
    // Two forms of an iteration.  "Cruise": edge pass + node pass (whose last workgroup screens the rule and halts the batch
    // when it cannot rule it out).  "Decide": the same two, and behind them ONE launch that forms the four Gram matrices on
    // the matrix cores and brackets their largest eigenvalues (k_batch_gram; it falls through unless the screen asked for
    // it), so that the stretch of a run in which the cheap bounds straddle tol (a third to a half of it) is judged on the
    // device too.
    // A call cruises until the screen halts it for the first time, then continues in the other form (B <= 32).
    const bool can_decide = B <= BATCH_BRACKET_MAXB && !getenv("GRAPHLEARN_AMD_BATCH_HOST_RULE");
    auto one_iteration = [&](bool decide, int par) {  // par: parity of the iteration's index (its input lives in w[par])
        k_edge_batch<<<dim3((unsigned)D0.G, (unsigned)B), dim3(64), 0, ctx->stream>>>((const LogdegDev *)sc.devs, par);
        k_node_batch<<<dim3((unsigned)D0.nb, (unsigned)B), dim3(NODE_WAVES * 64), 0, ctx->stream>>>((const LogdegDev *)sc.devs, st_dev, B, t2, par);
        if (decide) {
            // (the four Gram matrices on the matrix cores, reduced and bracketed by the launch's last workgroup: k_batch_gram)
            const int64_t units = n_tiled / 16;  // (about four of them per wave, on up to 128 workgroups of 8 waves)
            const unsigned ge = (unsigned)std::max<int64_t>(1, std::min<int64_t>(128, units / (4 * GRAM_WAVES)));
            if (B <= 16)
                k_batch_gram<1><<<ge + 1, GRAM_WAVES * 64, 0, ctx->stream>>>((const BatchPtrs *)sc.ptrs, B, n_tiled, D0.n_pad, D0.ctr + 2,
                                                                              (double *)sc.part, (const LogdegDev *)sc.devs, t2, st_dev);
            else
                k_batch_gram<2><<<ge + 1, GRAM_WAVES * 64, 0, ctx->stream>>>((const BatchPtrs *)sc.ptrs, B, n_tiled, D0.n_pad, D0.ctr + 2,
                                                                              (double *)sc.part, (const LogdegDev *)sc.devs, t2, st_dev);
        }
    };
    // (the launches carry the parity of their iteration: one captured chunk per form and per parity of its first iteration;
    //  a chunk that is halted half way leaves the next one to start on either)
    // (off by default: the launches of a chunk queue faster than they run -- 20 us of kernels per iteration against 8 us of
    //  host time --, and capturing a chunk costs ~0.4 ms per form and parity: 21.6 against 23.6 us per iteration at n = 256,
    //  B = 8 over 1299 iterations, 3.4 against 4.0 ms for a 212-iteration WindowLogModel fit.  GRAPHLEARN_AMD_BATCH_GRAPH=1)
    const bool use_graph = chunk >= 4 && chunk % 2 == 0 && getenv("GRAPHLEARN_AMD_BATCH_GRAPH") && !getenv("GRAPHLEARN_AMD_NO_GRAPH");
    struct Graphs {
        hipGraphExec_t g[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [decide][parity of the first iteration]
        ~Graphs() {
            for (auto &row : g)
                for (hipGraphExec_t q : row)
                    if (q) (void)hipGraphExecDestroy(q);
        }
    } graphs;
    auto capture = [&](bool decide, int par0, hipGraphExec_t *out) -> int {
        hipGraph_t gr = nullptr;
        GL_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < chunk; ++k) one_iteration(decide, (par0 + k) & 1);
        hipError_t e = hipStreamEndCapture(ctx->stream, &gr);
        if (e == hipSuccess) e = hipGraphInstantiate(out, gr, nullptr, nullptr, 0);
        if (gr) (void)hipGraphDestroy(gr);
        if (e != hipSuccess) return gl_set_error(ctx, GL_E_HIP, "batch graph capture failed: %s", hipGetErrorString(e));
        return GL_OK;
    };
    bool deciding = false;  // the form the call is in
    auto rearm = [&]() -> int {
        k_batch_rearm<<<dim3((unsigned)((B + 63) / 64)), 64, 0, ctx->stream>>>((const LogdegDev *)sc.devs, B, st_dev);
        GL_HIP(ctx, hipGetLastError());
        return GL_OK;
    };

    double rp = std::numeric_limits<double>::infinity(), rd = rp;
    bool have_exact = false, converged = false;
    int64_t n_iter = cruised;
    const int64_t it0 = sessions[0]->it - cruised;
    if (ring_finished && ring_exact) {  // (formed where the chunks ended: not a second time below)
        rp = ring_rp, rd = ring_rd;
        have_exact = true;
        converged = rp < tol && rd < tol;
    }
    while (n_iter < maxit && !ring_finished) {
        const int par0 = (int)((it0 + n_iter) & 1);
        if (use_graph) {
            hipGraphExec_t &gq = graphs.g[deciding ? 1 : 0][par0];
            if (!gq) GL_TRY(capture(deciding, par0, &gq));
            GL_HIP(ctx, hipGraphLaunch(gq, ctx->stream));
        } else {
            for (int k = 0; k < chunk; ++k) one_iteration(deciding, (par0 + k) & 1);
        }
        GL_HIP(ctx, hipGetLastError());
        GL_HIP(ctx, hipMemcpyAsync(st_host, st_dev, sizeof(BatchState), hipMemcpyDeviceToHost, ctx->stream));
        GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (st_host->ran > n_iter) have_exact = false;
        n_iter = st_host->ran;
        for (int a = 0; a < B; ++a) sessions[a]->it = it0 + n_iter;
        if (st_host->halt == 0) continue;  // the whole chunk ran
        if (st_host->halt == 1 || st_host->halt == 3) {  // the rule could not be ruled out for this iterate (3: the device found it met)
            GL_TRY(exact(rp, rd));
            have_exact = true;
            if (st_host->halt == 3 || (rp < tol && rd < tol)) {  // :166-167, after the update
                converged = true;
                break;
            }
        }
        if (n_iter >= maxit) break;
        if (can_decide && !deciding) {  // from here on the device judges such iterates itself
            deciding = true;
            st_host->decide_here = 1;
            GL_HIP(ctx, hipMemcpyAsync(&st_dev->decide_here, &st_host->decide_here, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        }
        GL_TRY(rearm());
    }
    GL_TRY(rearm());  // the done flags belong to this call's screen
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (getenv("GRAPHLEARN_AMD_BATCH_DEBUG")) {
        GL_HIP(ctx, hipMemcpy(st_host, st_dev, sizeof(BatchState), hipMemcpyDeviceToHost));
        fprintf(stderr, "[batch gram, last workgroup, 10 ns ticks summed] loop %.0f  tree+store+count %.0f  reduce %.0f  brackets %.0f\n", st_host->diag[0],
                st_host->diag[1], st_host->diag[2], st_host->diag[3]);
    }
    if (!have_exact && n_iter > 1) {  // the values the caller sees; the budget's last iteration is judged like any other (:166-167)
        GL_TRY(exact(rp, rd));
        if (rp < tol && rd < tol) converged = true;
    }
    if (n_iter_out) *n_iter_out = n_iter;
    if (converged_out) *converged_out = converged ? 1 : 0;
    if (rel_out) {
        rel_out[0] = rp;
        rel_out[1] = rd;
    }
    return GL_OK;
}

// Measurement facility for bench.py: average duration of the edge pass alone, timed with HIP
// events on the solver's stream over `reps` back-to-back launches.  The iteration counter is not
// advanced, so every launch recomputes the same pass (reads w[it&1], writes w[(it+1)&1]) and the
// session state is unchanged.
GL_API int gl_logdeg_engine(gl_logdeg *s) {
    if (!s) return 0;
    (void)ensure_engines(s);
    if (s->inited && gl_small_applicable(s)) return 2;
    return s->tile ? 3 : (s->res ? 1 : 0);
}

GL_API int gl_logdeg_time_edge_pass(gl_logdeg *s, int reps, double *avg_us_out) {
    GL_TRY(check(s));
    gl_ctx *ctx = s->ctx;
    if (!s->inited) return gl_set_error(ctx, GL_E_STATE, "time_edge_pass needs init first");
    if (reps < 1 || !avg_us_out) return gl_set_error(ctx, GL_E_INVALID, "bad argument");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    const LogdegDev &D = s->dev;
    for (int i = 0; i < 5; ++i) k_edge<false><<<dim3((unsigned)D.G), dim3(64), 0, ctx->stream>>>(D, D.dvar, D.pbar);
    GL_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < reps; ++i) k_edge<false><<<dim3((unsigned)D.G), dim3(64), 0, ctx->stream>>>(D, D.dvar, D.pbar);
    GL_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    GL_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    GL_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *avg_us_out = (double)ms * 1e3 / reps;
    return GL_OK;
}


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
#include "gl_logdeg.h"

#include <algorithm>
#include <limits>

namespace {

constexpr int TB = GL_TB;            // 64 x 64 tiles
constexpr int EDGE_THREADS = 256;    // 4 waves, 16 tile rows each
constexpr int ROWS_PER_LANE = 8;     // lane: 2 columns x 8 rows

__device__ __forceinline__ double shfl_xor_d(double v, int mask) { return __shfl_xor(v, mask, 64); }

// Sum v[0..7] (row partials of 8 different rows) across the 32 lanes of each
// half-wave with a halving butterfly: 4+2+1+1+1 exchanges instead of 8*5.
// On return every lane holds the complete sum of row k = (lane >> 2) & 7.
__device__ __forceinline__ double halfwave_rows8(const double (&v)[8], int lane) {
    const bool b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
    double a[4], b[2], c;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        double keep = b4 ? v[m + 4] : v[m];
        double send = b4 ? v[m] : v[m + 4];
        a[m] = keep + shfl_xor_d(send, 16);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        double keep = b3 ? a[m + 2] : a[m];
        double send = b3 ? a[m] : a[m + 2];
        b[m] = keep + shfl_xor_d(send, 8);
    }
    {
        double keep = b2 ? b[1] : b[0];
        double send = b2 ? b[0] : b[1];
        c = keep + shfl_xor_d(send, 4);
    }
    c += shfl_xor_d(c, 2);
    c += shfl_xor_d(c, 1);
    return c;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_d(v, m);
    return v;
}

// The stopping rule of smooth_learning.py:156-161,166 evaluated from the nb x 4
// norm partials that the node pass of the previous iteration left behind.
// Every caller gets bit-identical rp/rd (fixed reduction order).  One wave.
__device__ __forceinline__ void eval_rel_changes(const double *bn, int64_t nb, int lane, double &rp,
                                                 double &rd) {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int64_t b = lane; b < nb; b += 64) {
        s0 += bn[4 * b + 0];
        s1 += bn[4 * b + 1];
        s2 += bn[4 * b + 2];
        s3 += bn[4 * b + 3];
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    s3 = wave_sum(s3);
    rp = sqrt(s0) / sqrt(s1);
    rd = sqrt(s2) / sqrt(s3);
}

// ---------------------------------------------------------------------------
// Edge pass.  One workgroup per 64x64 tile (I <= J).  INIT = true computes only
// the degree sums of the current iterate (dvar = S w, :144).
template <bool INIT>
__global__ __launch_bounds__(EDGE_THREADS) void k_edge(LogdegDev P, int64_t it) {
    __shared__ double s_row[2][TB];
    __shared__ double s_col[2][4][TB];
    __shared__ double s_nrm[2][4];

    if (!INIT && P.status[0] != 0.0) return;  // converged earlier: leave everything as is

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int I, J;
    gl_tile_decode((int64_t)blockIdx.x, P.nb, I, J);
    const int half = lane >> 5;
    const int c0 = 2 * (lane & 31);
    const int r0 = wave * 16 + half;  // + 2k
    const double *__restrict__ wsrc = P.w[it & 1];
    double *__restrict__ wdst = P.w[(it + 1) & 1];
    const int64_t base = (int64_t)blockIdx.x * GL_TILE_ELEMS + (int64_t)r0 * TB + c0;

    double2 wv[ROWS_PER_LANE], zv[ROWS_PER_LANE];
#pragma unroll
    for (int k = 0; k < ROWS_PER_LANE; ++k) {
        wv[k] = *reinterpret_cast<const double2 *>(wsrc + base + (int64_t)k * 2 * TB);
        if (!INIT) zv[k] = *reinterpret_cast<const double2 *>(P.z + base + (int64_t)k * 2 * TB);
    }

    const int64_t j0 = (int64_t)J * TB + c0;
    const int64_t ibase = (int64_t)I * TB + r0;
    double dj[2] = {0, 0}, pj[2] = {0, 0};
    if (!INIT) {
        dj[0] = P.dvar[j0];
        dj[1] = P.dvar[j0 + 1];
        pj[0] = P.pbar[j0];
        pj[1] = P.pbar[j0 + 1];
    }
    const bool full = (I < J) && ((int64_t)J * TB + TB <= P.n);

    double rowp[ROWS_PER_LANE], roww[ROWS_PER_LANE];
    double colp[2] = {0, 0}, colw[2] = {0, 0};
    double nqy = 0, nw = 0;
    const double c = P.c, g = P.gamma, g2 = P.g2;

#pragma unroll
    for (int k = 0; k < ROWS_PER_LANE; ++k) {
        const int64_t i = ibase + 2 * k;
        double di = 0, pi = 0;
        if (!INIT) {
            di = P.dvar[i];
            pi = P.pbar[i];
        }
        double wn[2], pp[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int64_t j = j0 + e;
            const bool valid = full || (i < j && j < P.n);
            const double w = e ? wv[k].y : wv[k].x;
            if (INIT) {
                pp[e] = 0.0;
                wn[e] = valid ? w : 0.0;
            } else {
                const double z = e ? zv[k].y : zv[k].x;
                // :147  y = (1 - 2 gamma beta) w - gamma (S^T d)
                const double y = c * w - g * (di + dj[e]);
                // :150  p = max(0, y - 2 gamma z)
                const double t = y - g2 * z;
                const double p = t > 0 ? t : 0.0;
                // :153  q = (1 - 2 gamma beta) p - gamma (S^T pbar)
                const double q = c * p - g * (pi + pj[e]);
                // :163  w <- w - y + q
                const double wnew = (w - y) + q;
                const double dq = q - y;  // == -y + q (:158)
                pp[e] = valid ? p : 0.0;
                wn[e] = valid ? wnew : 0.0;
                nqy += valid ? dq * dq : 0.0;
                nw += valid ? w * w : 0.0;
            }
            colp[e] += pp[e];
            colw[e] += wn[e];
        }
        rowp[k] = pp[0] + pp[1];
        roww[k] = wn[0] + wn[1];
        if (!INIT) {
            double2 o;
            o.x = wn[0];
            o.y = wn[1];
            *reinterpret_cast<double2 *>(wdst + base + (int64_t)k * 2 * TB) = o;
        }
    }

    // ---- degree partials of this tile: rows (nodes of block I), columns (block J)
    const double rw = halfwave_rows8(roww, lane);
    double rp_ = 0;
    if (!INIT) rp_ = halfwave_rows8(rowp, lane);
    if ((lane & 3) == 0) {
        const int r = wave * 16 + 2 * ((lane >> 2) & 7) + half;
        s_row[0][r] = rp_;
        s_row[1][r] = rw;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        colp[e] += shfl_xor_d(colp[e], 32);
        colw[e] += shfl_xor_d(colw[e], 32);
    }
    if (half == 0) {
        s_col[0][wave][c0] = colp[0];
        s_col[0][wave][c0 + 1] = colp[1];
        s_col[1][wave][c0] = colw[0];
        s_col[1][wave][c0 + 1] = colw[1];
    }
    if (!INIT) {
        nqy = wave_sum(nqy);
        nw = wave_sum(nw);
        if (lane == 0) {
            s_nrm[0][wave] = nqy;
            s_nrm[1][wave] = nw;
        }
    }
    __syncthreads();

    if (tid < 2 * TB) {
        const int q = tid >> 6;  // 0: p sums, 1: w sums
        const int r = tid & 63;
        const double rs = s_row[q][r];
        const double cs = ((s_col[q][0][r] + s_col[q][1][r]) + s_col[q][2][r]) + s_col[q][3][r];
        double *part = q ? P.part_w : P.part_p;
        if (!(INIT && q == 0)) {
            if (I == J) {
                part[(int64_t)I * P.n_pad + (int64_t)I * TB + r] = rs + cs;
            } else {
                part[(int64_t)J * P.n_pad + (int64_t)I * TB + r] = rs;  // node I*TB+r, slot J
                part[(int64_t)I * P.n_pad + (int64_t)J * TB + r] = cs;  // node J*TB+r, slot I
            }
        }
    }
    if (!INIT && tid == 0) {
        P.tnorm[2 * (int64_t)blockIdx.x + 0] = ((s_nrm[0][0] + s_nrm[0][1]) + s_nrm[0][2]) + s_nrm[0][3];
        P.tnorm[2 * (int64_t)blockIdx.x + 1] = ((s_nrm[1][0] + s_nrm[1][1]) + s_nrm[1][2]) + s_nrm[1][3];
    }
}

// ---------------------------------------------------------------------------
// Node pass.  One workgroup per block of 64 nodes; wave v sums slab slots
// s = v, v+4, ... ; lanes are nodes.  Order of every sum is fixed.
// Also evaluates the stopping rule of the PREVIOUS iteration (deferred by one
// pass; the ping-pong edge buffers keep the iterate it refers to intact).
template <bool INIT>
__global__ __launch_bounds__(256) void k_node(LogdegDev P, int64_t it) {
    __shared__ double s_sum[2][4][TB];
    __shared__ double s_red[4][4];
    __shared__ int s_stop;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t b = blockIdx.x;

    if (!INIT) {
        if (P.status[0] != 0.0) return;
        if (it >= 2) {  // rule of iteration it-1 >= 1 (the reference skips i == 0, :156-161)
            if (wave == 0) {
                double rp, rd;
                eval_rel_changes(P.bnorm + ((it - 1) & 1) * 4 * P.nb, P.nb, lane, rp, rd);
                const int stop = (rp < P.tol && rd < P.tol) ? 1 : 0;
                if (lane == 0) {
                    s_stop = stop;
                    if (b == 0) {
                        P.status[2] = rp;
                        P.status[3] = rd;
                        if (stop) {
                            P.status[1] = (double)it;  // iterations completed
                            P.status[0] = 1.0;
                        }
                    }
                }
            }
            __syncthreads();
            if (s_stop) return;
        }
    }

    // slab sums for my 64 nodes
    const int64_t i = b * TB + lane;
    double sp = 0, sw = 0;
#pragma unroll 4
    for (int64_t s = wave; s < P.nb; s += 4) {
        if (!INIT) sp += P.part_p[s * P.n_pad + i];
        sw += P.part_w[s * P.n_pad + i];
    }
    s_sum[0][wave][lane] = sp;
    s_sum[1][wave][lane] = sw;

    // my share of the per-tile primal norm partials
    double t0 = 0, t1 = 0;
    if (!INIT) {
        for (int64_t t = b + (int64_t)tid * P.nb; t < P.ntiles; t += 256 * P.nb) {
            t0 += P.tnorm[2 * t + 0];
            t1 += P.tnorm[2 * t + 1];
        }
        t0 = wave_sum(t0);
        t1 = wave_sum(t1);
        if (lane == 0) {
            s_red[0][wave] = t0;
            s_red[1][wave] = t1;
        }
    }
    __syncthreads();

    if (wave == 0) {
        sp = ((s_sum[0][0][lane] + s_sum[0][1][lane]) + s_sum[0][2][lane]) + s_sum[0][3][lane];
        sw = ((s_sum[1][0][lane] + s_sum[1][1][lane]) + s_sum[1][2][lane]) + s_sum[1][3][lane];
        const bool real = i < P.n;
        double dnew, dq2 = 0, d2 = 0;
        if (INIT) {
            dnew = sw;  // :144  dvar = S w
        } else {
            const double d = P.dvar[i], yb = P.ybar[i], pb = P.pbar[i];
            const double qb = pb + P.gamma * sp;  // :154
            const double dq = qb - yb;            // == -ybar + qbar (:159)
            dq2 = real ? dq * dq : 0.0;
            d2 = real ? d * d : 0.0;
            dnew = (d - yb) + qb;                 // :164
        }
        // next iteration's forward + prox on the dual side (:148, :151)
        const double ybn = dnew + P.gamma * sw;
        const double pbn = (ybn - sqrt(ybn * ybn + P.a4g)) / 2;
        P.dvar[i] = real ? dnew : 0.0;
        P.ybar[i] = real ? ybn : 0.0;
        P.pbar[i] = real ? pbn : 0.0;
        if (!INIT) {
            dq2 = wave_sum(dq2);
            d2 = wave_sum(d2);
            if (lane == 0) {
                double *bn = P.bnorm + (it & 1) * 4 * P.nb + 4 * b;
                bn[0] = ((s_red[0][0] + s_red[0][1]) + s_red[0][2]) + s_red[0][3];
                bn[1] = ((s_red[1][0] + s_red[1][1]) + s_red[1][2]) + s_red[1][3];
                bn[2] = dq2;
                bn[3] = d2;
            }
        }
    }
}

// After a batch of launches: rel. changes of the last completed iteration, and
// the stopping rule for it (the deferred check has not seen it yet).
__global__ void k_status(LogdegDev P, int64_t it) {
    if (P.status[0] != 0.0 || it < 2) return;  // iteration 0 never stops (:156-161)
    double rp, rd;
    eval_rel_changes(P.bnorm + ((it - 1) & 1) * 4 * P.nb, P.nb, threadIdx.x, rp, rd);
    if (threadIdx.x == 0) {
        P.status[2] = rp;
        P.status[3] = rd;
        if (rp < P.tol && rd < P.tol) {
            P.status[1] = (double)it;
            P.status[0] = 1.0;
        }
    }
}

// ---------------------------------------------------------------------------
// Layout conversion kernels (one workgroup per tile, same lane map as k_edge).
template <int MODE>  // 0: from condensed, 1: from square (upper triangle)
__global__ __launch_bounds__(256) void k_to_tiled(const double *__restrict__ src, double *__restrict__ dst,
                                                  int64_t n, int64_t nb, double scale) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int I, J;
    gl_tile_decode((int64_t)blockIdx.x, nb, I, J);
    const int c0 = 2 * (lane & 31);
    const int r0 = wave * 16 + (lane >> 5);
    const int64_t base = (int64_t)blockIdx.x * GL_TILE_ELEMS + (int64_t)r0 * TB + c0;
#pragma unroll
    for (int k = 0; k < ROWS_PER_LANE; ++k) {
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
    for (int k = 0; k < ROWS_PER_LANE; ++k) {
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

int check(gl_logdeg *s) { return (s && s->ctx) ? GL_OK : GL_E_INVALID; }

}  // namespace

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
    const size_t tiled_bytes = (size_t)D.ntiles * GL_TILE_ELEMS * sizeof(double);
    auto alloc = [&](double **p, size_t bytes) -> int {
        hipError_t e = hipMalloc((void **)p, bytes);
        if (e != hipSuccess) return gl_set_error(ctx, GL_E_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return GL_OK;
    };
    int rc = GL_OK;
    double *zbuf = nullptr;
    if ((rc = alloc(&D.w[0], tiled_bytes)) || (rc = alloc(&D.w[1], tiled_bytes)) || (rc = alloc(&zbuf, tiled_bytes)) ||
        (rc = alloc(&D.dvar, D.n_pad * 8)) || (rc = alloc(&D.ybar, D.n_pad * 8)) || (rc = alloc(&D.pbar, D.n_pad * 8)) ||
        (rc = alloc(&D.part_p, (size_t)D.nb * D.n_pad * 8)) || (rc = alloc(&D.part_w, (size_t)D.nb * D.n_pad * 8)) ||
        (rc = alloc(&D.tnorm, (size_t)D.ntiles * 2 * 8)) || (rc = alloc(&D.bnorm, (size_t)2 * D.nb * 4 * 8)) ||
        (rc = alloc(&D.status, 8 * 8)) || (rc = alloc(&s->stage, (size_t)s->n_edges * 8))) {
        D.z = zbuf;
        gl_logdeg_destroy(s);
        return rc;
    }
    D.z = zbuf;
    *out = s;
    return GL_OK;
}

GL_API int gl_logdeg_destroy(gl_logdeg *s) {
    if (!s) return GL_OK;
    if (s->ctx) {
        (void)hipSetDevice(s->ctx->device);
        (void)hipStreamSynchronize(s->ctx->stream);
    }
    LogdegDev &D = s->dev;
    double *ptrs[] = {D.w[0], D.w[1], const_cast<double *>(D.z), D.dvar, D.ybar, D.pbar, D.part_p, D.part_w,
                      D.tnorm, D.bnorm, D.status, s->stage, s->zsq, s->xdev, s->theta_terms};
    for (double *p : ptrs)
        if (p) (void)hipFree(p);
    delete s;
    return GL_OK;
}

GL_API int gl_logdeg_set_distances(gl_logdeg *s, const double *z_condensed) {
    GL_TRY(check(s));
    gl_ctx *ctx = s->ctx;
    if (!z_condensed) return gl_set_error(ctx, GL_E_INVALID, "z_condensed is NULL");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    GL_HIP(ctx, hipMemcpyAsync(s->stage, z_condensed, (size_t)s->n_edges * 8, hipMemcpyHostToDevice, ctx->stream));
    k_to_tiled<0><<<dim3((unsigned)s->dev.ntiles), dim3(256), 0, ctx->stream>>>(
        s->stage, const_cast<double *>(s->dev.z), s->n, s->dev.nb, 1.0);
    GL_HIP(ctx, hipGetLastError());
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    s->have_z = true;
    s->have_zsq = false;
    s->theta = 1.0;
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
    if (s->xdev_elems < xe) {
        if (s->xdev) (void)hipFree(s->xdev);
        s->xdev = nullptr;
        s->xdev_elems = 0;
        GL_HIP(ctx, hipMalloc((void **)&s->xdev, (size_t)xe * 8));
        s->xdev_elems = xe;
    }
    if (!s->zsq) GL_HIP(ctx, hipMalloc((void **)&s->zsq, (size_t)s->n * s->n * 8));
    if (!s->theta_terms) GL_HIP(ctx, hipMalloc((void **)&s->theta_terms, (size_t)2 * s->n * 8));
    GL_HIP(ctx, hipMemcpyAsync(s->xdev, x, (size_t)xe * 8, hipMemcpyHostToDevice, ctx->stream));
    GL_TRY(gl_gram_distances(ctx, s->xdev, d, s->n, s->zsq));
    double theta = 1.0;
    if (avg_degree > 0) GL_TRY(gl_theta_from_square(ctx, s->zsq, s->n, avg_degree, s->theta_terms, &theta));
    k_to_tiled<1><<<dim3((unsigned)s->dev.ntiles), dim3(256), 0, ctx->stream>>>(
        s->zsq, const_cast<double *>(s->dev.z), s->n, s->dev.nb, theta);
    GL_HIP(ctx, hipGetLastError());
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    s->theta = theta;
    s->have_z = true;
    s->have_zsq = true;
    s->inited = false;
    if (theta_out) *theta_out = theta;
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
    GL_HIP(ctx, hipMemcpyAsync(z_condensed, s->stage, (size_t)s->n_edges * 8, hipMemcpyDeviceToHost, ctx->stream));
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GL_OK;
}

GL_API int gl_logdeg_init(gl_logdeg *s, double alpha, double beta, double step_size, const double *edge_init) {
    GL_TRY(check(s));
    gl_ctx *ctx = s->ctx;
    if (!s->have_z) return gl_set_error(ctx, GL_E_STATE, "init needs set_signals or set_distances first");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    LogdegDev &D = s->dev;
    // :135-137
    D.gamma = step_size / (2 * beta + sqrt((double)(2 * (s->n - 1))));
    D.c = 1 - D.gamma * 2 * beta;
    D.g2 = 2 * D.gamma;
    D.a4g = 4 * alpha * D.gamma;
    D.tol = 0.0;
    const size_t tiled_bytes = (size_t)D.ntiles * GL_TILE_ELEMS * sizeof(double);
    if (edge_init) {
        GL_HIP(ctx, hipMemcpyAsync(s->stage, edge_init, (size_t)s->n_edges * 8, hipMemcpyHostToDevice, ctx->stream));
        k_to_tiled<0><<<dim3((unsigned)D.ntiles), dim3(256), 0, ctx->stream>>>(s->stage, D.w[0], s->n, D.nb, 1.0);
        GL_HIP(ctx, hipGetLastError());
    } else {
        GL_HIP(ctx, hipMemsetAsync(D.w[0], 0, tiled_bytes, ctx->stream));
    }
    GL_HIP(ctx, hipMemsetAsync(D.w[1], 0, tiled_bytes, ctx->stream));
    GL_HIP(ctx, hipMemsetAsync(D.bnorm, 0, (size_t)2 * D.nb * 4 * 8, ctx->stream));
    const double inf = std::numeric_limits<double>::infinity();
    double *st = (double *)ctx->pinned;
    st[0] = 0.0;
    st[1] = 0.0;
    st[2] = inf;
    st[3] = inf;
    st[4] = st[5] = st[6] = st[7] = 0.0;
    GL_HIP(ctx, hipMemcpyAsync(D.status, st, 8 * 8, hipMemcpyHostToDevice, ctx->stream));
    s->it = 0;
    k_edge<true><<<dim3((unsigned)D.ntiles), dim3(EDGE_THREADS), 0, ctx->stream>>>(D, 0);
    GL_HIP(ctx, hipGetLastError());
    k_node<true><<<dim3((unsigned)D.nb), dim3(256), 0, ctx->stream>>>(D, 0);
    GL_HIP(ctx, hipGetLastError());
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    s->inited = true;
    return GL_OK;
}

static int launch_iterations(gl_logdeg *s, int64_t count) {
    gl_ctx *ctx = s->ctx;
    LogdegDev &D = s->dev;
    for (int64_t k = 0; k < count; ++k) {
        const int64_t it = s->it + k;
        k_edge<false><<<dim3((unsigned)D.ntiles), dim3(EDGE_THREADS), 0, ctx->stream>>>(D, it);
        k_node<false><<<dim3((unsigned)D.nb), dim3(256), 0, ctx->stream>>>(D, it);
    }
    GL_HIP(ctx, hipGetLastError());
    s->it += count;
    return GL_OK;
}

GL_API int gl_logdeg_iterate_async(gl_logdeg *s, int64_t n_iter) {
    GL_TRY(check(s));
    gl_ctx *ctx = s->ctx;
    if (!s->inited) return gl_set_error(ctx, GL_E_STATE, "iterate needs init first");
    if (n_iter < 0) return gl_set_error(ctx, GL_E_INVALID, "n_iter < 0");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    s->dev.tol = 0.0;
    return launch_iterations(s, n_iter);
}

GL_API int gl_logdeg_iterate(gl_logdeg *s, int64_t maxit, double tol, int64_t *n_iter_out, int *converged_out,
                             double *rel_out) {
    GL_TRY(check(s));
    gl_ctx *ctx = s->ctx;
    if (!s->inited) return gl_set_error(ctx, GL_E_STATE, "iterate needs init first");
    if (maxit < 0) return gl_set_error(ctx, GL_E_INVALID, "maxit < 0");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    LogdegDev &D = s->dev;
    D.tol = tol;
    const int64_t start = s->it;
    const int64_t end = start + maxit;
    double *st = (double *)ctx->pinned;
    bool done = false;
    st[2] = st[3] = std::numeric_limits<double>::infinity();
    // poll period: ~1% of the work wasted at most once, and a bounded launch queue
    const int64_t chunk = 64;
    while (!done && s->it < end) {
        const int64_t todo = std::min<int64_t>(chunk, end - s->it);
        GL_TRY(launch_iterations(s, todo));
        k_status<<<dim3(1), dim3(64), 0, ctx->stream>>>(D, s->it);
        GL_HIP(ctx, hipGetLastError());
        GL_HIP(ctx, hipMemcpyAsync(st, D.status, 8 * 8, hipMemcpyDeviceToHost, ctx->stream));
        GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (st[0] != 0.0) {
            done = true;
            s->it = (int64_t)st[1];
        }
    }
    if (done) {
        // re-arm the session: the flag belongs to this call's stopping rule
        GL_HIP(ctx, hipMemsetAsync(D.status, 0, 2 * 8, ctx->stream));
        GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (n_iter_out) *n_iter_out = s->it - start;
    if (converged_out) *converged_out = done ? 1 : 0;
    if (rel_out) {
        rel_out[0] = st[2];
        rel_out[1] = st[3];
    }
    return GL_OK;
}

GL_API int64_t gl_logdeg_total_iterations(gl_logdeg *s) { return s ? s->it : -1; }

GL_API int gl_logdeg_get_weights_condensed(gl_logdeg *s, double edge_tol, double *w_out) {
    GL_TRY(check(s));
    gl_ctx *ctx = s->ctx;
    if (!w_out) return gl_set_error(ctx, GL_E_INVALID, "w_out is NULL");
    if (!s->inited) return gl_set_error(ctx, GL_E_STATE, "no iterate yet: call init first");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    k_to_condensed<<<dim3((unsigned)s->dev.ntiles), dim3(256), 0, ctx->stream>>>(s->dev.w[s->it & 1], s->stage, s->n,
                                                                              s->dev.nb, edge_tol);
    GL_HIP(ctx, hipGetLastError());
    GL_HIP(ctx, hipMemcpyAsync(w_out, s->stage, (size_t)s->n_edges * 8, hipMemcpyDeviceToHost, ctx->stream));
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GL_OK;
}

GL_API int gl_logdeg_get_weights_square(gl_logdeg *s, double edge_tol, double *w_out) {
    GL_TRY(check(s));
    gl_ctx *ctx = s->ctx;
    if (!w_out) return gl_set_error(ctx, GL_E_INVALID, "w_out is NULL");
    if (!s->inited) return gl_set_error(ctx, GL_E_STATE, "no iterate yet: call init first");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    // the square result (n^2) is staged in zsq-sized scratch; allocate on demand
    double *sq = nullptr;
    GL_HIP(ctx, hipMalloc((void **)&sq, (size_t)s->n * s->n * 8));
    k_to_square<<<dim3((unsigned)s->dev.ntiles), dim3(256), 0, ctx->stream>>>(s->dev.w[s->it & 1], sq, s->n,
                                                                           s->dev.nb, edge_tol);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(w_out, sq, (size_t)s->n * s->n * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(sq);
    if (e != hipSuccess) return gl_set_error(ctx, GL_E_HIP, "get_weights_square: %s", hipGetErrorString(e));
    return GL_OK;
}

GL_API int gl_logdeg_get_dual(gl_logdeg *s, double *d_out) {
    GL_TRY(check(s));
    gl_ctx *ctx = s->ctx;
    if (!d_out) return gl_set_error(ctx, GL_E_INVALID, "d_out is NULL");
    if (!s->inited) return gl_set_error(ctx, GL_E_STATE, "no iterate yet: call init first");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    GL_HIP(ctx, hipMemcpyAsync(d_out, s->dev.dvar, (size_t)s->n * 8, hipMemcpyDeviceToHost, ctx->stream));
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

// The theta heuristic of the log-degree model.
//
// Reference: get_theta, blocks=None branch, graph_learn/smooth_learning.py:201-219:
//   sorted = sort(rows of the SQUARE distance matrix, zero diagonal included)
//   head_i = sum(sorted[i, :k]);  lo_i = (k v_k^2 - head v_k)^-1/2 with v_k = sorted[i, k],
//   hi_i the same with v_{k-1};  theta = sqrt(mean(lo) * mean(hi)).
// Only the k+1 smallest entries of each row matter, so instead of sorting, each
// workgroup runs an exact 8-pass radix select on order-preserving keys made
// from the IEEE bit patterns and then one pass
// for {count, sum, max} of the entries below v_k.  Ties are handled exactly.
#include "gl_logdeg.h"

#include <vector>

namespace {

// order-preserving map double -> uint64 (negative values too, for user-supplied matrices)
__device__ __forceinline__ unsigned long long key_of(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return u ^ ((u >> 63) ? ~0ull : 0x8000000000000000ull);
}
__device__ __forceinline__ double value_of(unsigned long long key) {
    const unsigned long long u = key ^ ((key >> 63) ? 0x8000000000000000ull : ~0ull);
    return __longlong_as_double((long long)u);
}

// one workgroup per row; rows of length n; terms[row] and terms[rows + row].
// RPT > 0: the row's keys live in registers (RPT per thread, n <= 256 RPT) and are read once; RPT = 0: longer rows are
// re-read from memory in every pass.  The bin that holds the wanted rank is found by one wave (four bins per lane, a
// prefix sum over the lanes).  Round 4: nine dependent reads of the row and a 256-step scan by one thread per pass made a
// row cost 80 us whatever its length; the sums of the last pass keep their order (thread tid: entries tid, tid + 256, ...).
template <int RPT>
__global__ __launch_bounds__(256) void k_theta_rows(const double *__restrict__ zsq, int64_t n, int k,
                                                    double *__restrict__ terms, int64_t rows) {
    __shared__ unsigned s_hist[256];
    __shared__ unsigned long long s_prefix;
    __shared__ unsigned s_rank;
    __shared__ double s_sum[256];
    __shared__ double s_max[256];
    __shared__ unsigned s_cnt[256];

    const int tid = threadIdx.x;
    const double *row = zsq + (int64_t)blockIdx.x * n;
    constexpr int NR = RPT > 0 ? RPT : 1;
    unsigned long long keys[NR];
    if (RPT > 0) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int64_t j = tid + 256 * r;
            keys[r] = j < n ? key_of(row[j]) : ~0ull;  // (past the row: behind every entry, never counted below v_k)
        }
    }
    if (tid == 0) {
        s_prefix = 0ull;
        s_rank = (unsigned)k;  // 0-based position of v_k in the sorted row
    }
    for (int pass = 7; pass >= 0; --pass) {
        s_hist[tid] = 0u;
        __syncthreads();
        const unsigned long long prefix = s_prefix;
        if (RPT > 0) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const unsigned long long key = keys[r];
                if (tid + 256 * r < n && (pass == 7 || (key >> (8 * (pass + 1))) == prefix)) atomicAdd(&s_hist[(key >> (8 * pass)) & 255u], 1u);
            }
        } else {
            for (int64_t j = tid; j < n; j += 256) {
                const unsigned long long key = key_of(row[j]);
                if (pass == 7 || (key >> (8 * (pass + 1))) == prefix) atomicAdd(&s_hist[(key >> (8 * pass)) & 255u], 1u);
            }
        }
        __syncthreads();
        if (tid < 64) {  // the first bin whose cumulative count exceeds the rank: lane l takes bins 4 l .. 4 l + 3
            const unsigned rank = s_rank;
            const unsigned h0 = s_hist[4 * tid], h1 = s_hist[4 * tid + 1], h2 = s_hist[4 * tid + 2], h3 = s_hist[4 * tid + 3];
            const unsigned t = h0 + h1 + h2 + h3;
            unsigned incl = t;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned up = __shfl_up(incl, o);
                if (tid >= o) incl += up;
            }
            const unsigned excl = incl - t;
            if (excl <= rank && rank < incl) {  // exactly one lane (the row holds more than `rank` entries)
                unsigned cum = excl;
                int bin = 4 * tid;
                if (cum + h0 <= rank) {
                    cum += h0, ++bin;
                    if (cum + h1 <= rank) {
                        cum += h1, ++bin;
                        if (cum + h2 <= rank) cum += h2, ++bin;
                    }
                }
                s_rank = rank - cum;
                s_prefix = (prefix << 8) | (unsigned long long)bin;
            }
        }
        __syncthreads();
    }
    const unsigned long long vk_key = s_prefix;
    const double vk = value_of(vk_key);

    double sum = 0, mx = -INFINITY;
    unsigned cnt = 0;
    if (RPT > 0) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const double v = value_of(keys[r]);
            if (tid + 256 * r < n && keys[r] < vk_key) {
                sum += v;
                mx = v > mx ? v : mx;
                ++cnt;
            }
        }
    } else {
        for (int64_t j = tid; j < n; j += 256) {
            const double v = row[j];
            if (key_of(v) < vk_key) {
                sum += v;
                mx = v > mx ? v : mx;
                ++cnt;
            }
        }
    }
    s_sum[tid] = sum;
    s_max[tid] = mx;
    s_cnt[tid] = cnt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            s_sum[tid] += s_sum[tid + s];
            s_max[tid] = s_max[tid + s] > s_max[tid] ? s_max[tid + s] : s_max[tid];
            s_cnt[tid] += s_cnt[tid + s];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const unsigned below = s_cnt[0];  // <= k by construction
        const double head = s_sum[0] + (double)((unsigned)k - below) * vk;
        const double vkm1 = (below >= (unsigned)k) ? s_max[0] : vk;
        const double kk = (double)k;
        terms[blockIdx.x] = pow(kk * (vk * vk) - head * vk, -0.5);
        terms[rows + blockIdx.x] = pow(kk * (vkm1 * vkm1) - head * vkm1, -0.5);
    }
}

void theta_rows_launch(gl_ctx *ctx, const double *zsq, int64_t rows, int64_t cols, int k, double *terms) {
    const dim3 g((unsigned)rows), b(256);
    if (cols <= 256)
        k_theta_rows<1><<<g, b, 0, ctx->stream>>>(zsq, cols, k, terms, rows);
    else if (cols <= 1024)
        k_theta_rows<4><<<g, b, 0, ctx->stream>>>(zsq, cols, k, terms, rows);
    else if (cols <= 4096)
        k_theta_rows<16><<<g, b, 0, ctx->stream>>>(zsq, cols, k, terms, rows);
    else
        k_theta_rows<0><<<g, b, 0, ctx->stream>>>(zsq, cols, k, terms, rows);
}

// numpy-style pairwise summation (np.mean -> np.add.reduce), blocks of 8 at the leaves
double pairwise_sum(const double *a, int64_t n) {
    if (n <= 128) {
        double s = 0;
        for (int64_t i = 0; i < n; ++i) s += a[i];
        return s;
    }
    const int64_t h = (n / 2) & ~(int64_t)7;
    return pairwise_sum(a, h) + pairwise_sum(a + h, n - h);
}

// the host's finish of theta on the device -- the same pairwise sums in the same order, by one thread --, so that nothing
// has to come back before the distances are scaled (a batch of windows: one host round trip per window otherwise)
__device__ double pairwise_sum_dev(const double *a, int64_t n) {
    if (n <= 128) {
        double s = 0;
        for (int64_t i = 0; i < n; ++i) s += a[i];
        return s;
    }
    const int64_t h = (n / 2) & ~(int64_t)7;
    return pairwise_sum_dev(a, h) + pairwise_sum_dev(a + h, n - h);
}
// The leaves of that recursion (runs of <= 128 entries, summed left to right) are independent: one thread each, the
// entries staged in LDS, and one thread folds the leaf sums in the recursion's order -- the same additions in the same
// order as pairwise_sum, 5 us instead of 41 for one thread's 2 x 256 dependent loads.
constexpr int TH_MAXROWS = 2048, TH_MAXLEAF = 64;
__device__ int theta_leaves(int64_t n, int64_t at, int *start, int *len, int nl) {
    if (n <= 128) {
        start[nl] = (int)at, len[nl] = (int)n;
        return nl + 1;
    }
    const int64_t h = (n / 2) & ~(int64_t)7;
    nl = theta_leaves(h, at, start, len, nl);
    return theta_leaves(n - h, at + h, start, len, nl);
}
__device__ double theta_fold(int64_t n, int &leaf, const double *ls) {
    if (n <= 128) return ls[leaf++];
    const int64_t h = (n / 2) & ~(int64_t)7;
    const double l = theta_fold(h, leaf, ls);
    const double r = theta_fold(n - h, leaf, ls);
    return l + r;
}
__global__ __launch_bounds__(256) void k_theta_finish(const double *__restrict__ terms, int rows, double *__restrict__ theta_dev) {
    __shared__ double s_t[2 * TH_MAXROWS];
    __shared__ double s_leaf[2][TH_MAXLEAF];
    __shared__ int s_start[TH_MAXLEAF], s_len[TH_MAXLEAF], s_nl;
    const int tid = threadIdx.x;
    if (rows > TH_MAXROWS) {  // (longer than the staging area: one thread, as the host would)
        if (tid == 0) {
            const double lo = pairwise_sum_dev(terms, rows), hi = pairwise_sum_dev(terms + rows, rows);
            theta_dev[0] = sqrt((lo / (double)rows) * (hi / (double)rows));
        }
        return;
    }
    for (int q = tid; q < 2 * rows; q += 256) s_t[q] = terms[q];
    if (tid == 0) s_nl = theta_leaves(rows, 0, s_start, s_len, 0);  // (rows <= 2048: at most 32 leaves)
    __syncthreads();
    const int nl = s_nl;
    if (tid < 2 * nl) {
        const int arr = tid / nl, lf = tid % nl;
        const double *a = s_t + arr * rows + s_start[lf];
        double sum = 0;
        for (int i = 0; i < s_len[lf]; ++i) sum += a[i];
        s_leaf[arr][lf] = sum;
    }
    __syncthreads();
    if (tid == 0) {
        int l0 = 0, l1 = 0;
        const double lo = theta_fold(rows, l0, s_leaf[0]), hi = theta_fold(rows, l1, s_leaf[1]);
        theta_dev[0] = sqrt((lo / (double)rows) * (hi / (double)rows));
    }
}

}  // namespace

// theta of a square matrix left in device memory; no synchronisation
int gl_theta_from_square_dev(gl_ctx *ctx, const double *zsq_dev, int64_t n, int k, double *terms_dev, double *theta_dev) {
    if (k < 1 || k >= n) return gl_set_error(ctx, GL_E_INVALID, "avg_degree must be in [1, %lld], got %d", (long long)n - 1, k);
    theta_rows_launch(ctx, zsq_dev, n, n, k, terms_dev);
    k_theta_finish<<<1, 256, 0, ctx->stream>>>(terms_dev, (int)n, theta_dev);
    GL_HIP(ctx, hipGetLastError());
    return GL_OK;
}

// rows x cols matrix (the square n x n case of LogModel, or a block of it: get_theta's blocks branch,
// smooth_learning.py:220-235, calls the plain branch on rectangular sub-matrices)
int gl_theta_from_rows(gl_ctx *ctx, const double *zsq_dev, int64_t rows, int64_t cols, int k, double *terms_dev,
                       double *theta_out) {
    if (k < 1 || k >= cols) return gl_set_error(ctx, GL_E_INVALID, "avg_degree must be in [1, %lld], got %d", (long long)cols - 1, k);
    theta_rows_launch(ctx, zsq_dev, rows, cols, k, terms_dev);
    GL_HIP(ctx, hipGetLastError());
    std::vector<double> host((size_t)2 * rows);
    GL_HIP(ctx, hipMemcpyAsync(host.data(), terms_dev, (size_t)2 * rows * 8, hipMemcpyDeviceToHost, ctx->stream));
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const double lo = pairwise_sum(host.data(), rows) / (double)rows;
    const double hi = pairwise_sum(host.data() + rows, rows) / (double)rows;
    *theta_out = sqrt(lo * hi);
    return GL_OK;
}

int gl_theta_from_square(gl_ctx *ctx, const double *zsq_dev, int64_t n, int k, double *terms_dev,
                         double *theta_out) {
    return gl_theta_from_rows(ctx, zsq_dev, n, n, k, terms_dev, theta_out);
}

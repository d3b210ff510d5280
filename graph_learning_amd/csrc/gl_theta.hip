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

// one workgroup per row; rows of length n; terms[row] and terms[rows + row]
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
    if (tid == 0) {
        s_prefix = 0ull;
        s_rank = (unsigned)k;  // 0-based position of v_k in the sorted row
    }
    for (int pass = 7; pass >= 0; --pass) {
        s_hist[tid] = 0u;
        __syncthreads();
        const unsigned long long prefix = s_prefix;
        for (int64_t j = tid; j < n; j += 256) {
            const unsigned long long key = key_of(row[j]);
            if (pass == 7 || (key >> (8 * (pass + 1))) == prefix) atomicAdd(&s_hist[(key >> (8 * pass)) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned rank = s_rank, cum = 0;
            int bin = 0;
            for (; bin < 256; ++bin) {
                if (cum + s_hist[bin] > rank) break;
                cum += s_hist[bin];
            }
            s_rank = rank - cum;
            s_prefix = (prefix << 8) | (unsigned long long)bin;
        }
        __syncthreads();
    }
    const unsigned long long vk_key = s_prefix;
    const double vk = value_of(vk_key);

    double sum = 0, mx = -INFINITY;
    unsigned cnt = 0;
    for (int64_t j = tid; j < n; j += 256) {
        const double v = row[j];
        if (key_of(v) < vk_key) {
            sum += v;
            mx = v > mx ? v : mx;
            ++cnt;
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
__global__ void k_theta_finish(const double *__restrict__ terms, int rows, double *__restrict__ theta_dev) {
    const double lo = pairwise_sum_dev(terms, rows), hi = pairwise_sum_dev(terms + rows, rows);
    theta_dev[0] = sqrt((lo / (double)rows) * (hi / (double)rows));
}

}  // namespace

// theta of a square matrix left in device memory; no synchronisation
int gl_theta_from_square_dev(gl_ctx *ctx, const double *zsq_dev, int64_t n, int k, double *terms_dev, double *theta_dev) {
    if (k < 1 || k >= n) return gl_set_error(ctx, GL_E_INVALID, "avg_degree must be in [1, %lld], got %d", (long long)n - 1, k);
    k_theta_rows<<<dim3((unsigned)n), dim3(256), 0, ctx->stream>>>(zsq_dev, n, k, terms_dev, n);
    k_theta_finish<<<1, 1, 0, ctx->stream>>>(terms_dev, (int)n, theta_dev);
    GL_HIP(ctx, hipGetLastError());
    return GL_OK;
}

// rows x cols matrix (the square n x n case of LogModel, or a block of it: get_theta's blocks branch,
// smooth_learning.py:220-235, calls the plain branch on rectangular sub-matrices)
int gl_theta_from_rows(gl_ctx *ctx, const double *zsq_dev, int64_t rows, int64_t cols, int k, double *terms_dev,
                       double *theta_out) {
    if (k < 1 || k >= cols) return gl_set_error(ctx, GL_E_INVALID, "avg_degree must be in [1, %lld], got %d", (long long)cols - 1, k);
    k_theta_rows<<<dim3((unsigned)rows), dim3(256), 0, ctx->stream>>>(zsq_dev, cols, k, terms_dev, rows);
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

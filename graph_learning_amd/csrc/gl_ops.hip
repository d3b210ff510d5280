// Laplacian <-> edge-vector operators (batched over a leading axis).
//
// Reference: graph_learn/operators.py
//   square_to_vec                 :39-41    -> gl_square_to_vec
//   laplacian_squareform(_vec)    :44-70    -> gl_laplacian_from_vec
//   laplacian_squareform_adj(_vec):73-94    -> gl_laplacian_adjoint (order 0)
//   op_adj_dual                   :176-180  -> gl_laplacian_adjoint (order 1)
//   squared_pdiffs                :224-226  -> gl_squared_pdiffs
// and get_theta (graph_learn/smooth_learning.py:201-219) on a host square matrix.
// All are one-pass HBM-bound kernels; host buffers in, host buffers out.
#include "gl_logdeg.h"

namespace {

// grid: x over j, y = i, z = batch.  out[b][e(i,j)] = x[b][i][j], i < j
__global__ void k_square_to_vec(const double *__restrict__ x, double *__restrict__ out, int64_t n, int64_t E) {
    const int64_t i = blockIdx.y, b = blockIdx.z;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > i && j < n) out[b * E + gl_condensed_index(i, j, n)] = x[(b * n + i) * n + j];
}

// off-diagonal part of L = D - W
__global__ void k_lap_offdiag(const double *__restrict__ w, double *__restrict__ out, int64_t n, int64_t E) {
    const int64_t i = blockIdx.y, b = blockIdx.z;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double v = 0.0;
    if (j != i) {
        const int64_t lo = i < j ? i : j, hi = i < j ? j : i;
        v = -w[b * E + gl_condensed_index(lo, hi, n)];
    }
    out[(b * n + i) * n + j] = v;
}

// diagonal: L_ii = -(sum_j L_ij) over the row written by k_lap_offdiag (diag still 0)
__global__ __launch_bounds__(256) void k_lap_diag(double *__restrict__ out, int64_t n) {
    __shared__ double s[256];
    const int64_t i = blockIdx.x, b = blockIdx.y;
    double *row = out + (b * n + i) * n;
    double acc = 0;
    for (int64_t j = threadIdx.x; j < n; j += 256) acc += row[j];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) row[i] = -s[0];
}

// (L* Y)_(i,j): order 0 = ((-Y_ij - Y_ji) + Y_ii) + Y_jj   (operators.py:88-93)
//               order 1 = ((Y_ii + Y_jj) - Y_ij) - Y_ji    (operators.py:178-180)
__global__ void k_lap_adjoint(const double *__restrict__ y, double *__restrict__ out, int64_t n, int64_t E,
                              int order) {
    const int64_t i = blockIdx.y, b = blockIdx.z;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > i && j < n) {
        const double *Y = y + b * n * n;
        const double yij = Y[i * n + j], yji = Y[j * n + i], yii = Y[i * n + i], yjj = Y[j * n + j];
        const double v = order == 0 ? ((-yij - yji) + yii) + yjj : ((yii + yjj) - yij) - yji;
        out[b * E + gl_condensed_index(i, j, n)] = v;
    }
}

__global__ void k_sq_pdiffs(const double *__restrict__ x, double *__restrict__ out, int64_t n, int64_t E) {
    const int64_t i = blockIdx.y, t = blockIdx.z;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > i && j < n) {
        const double dlt = x[t * n + i] - x[t * n + j];
        out[t * E + gl_condensed_index(i, j, n)] = dlt * dlt;
    }
}

struct DevBuf {
    double *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
};

int check_dims(gl_ctx *ctx, const void *a, const void *b, int64_t B, int64_t n) {
    if (!ctx) return GL_E_INVALID;
    if (!a || !b) return gl_set_error(ctx, GL_E_INVALID, "NULL buffer");
    if (B < 1 || n < 2) return gl_set_error(ctx, GL_E_INVALID, "need batch >= 1 and n >= 2");
    if (B > 65535 || n > 65535) return gl_set_error(ctx, GL_E_INVALID, "batch and n are limited to 65535");
    return GL_OK;
}

}  // namespace

#define OPS_PROLOGUE(in_elems, out_elems)                                                          \
    GL_TRY(check_dims(ctx, in, out, B, n));                                                        \
    GL_HIP(ctx, hipSetDevice(ctx->device));                                                        \
    const int64_t E = n * (n - 1) / 2;                                                             \
    (void)E;                                                                                       \
    DevBuf din, dout;                                                                              \
    GL_MALLOC(ctx, (void **)&din.p, (size_t)(in_elems) * 8, "ops");                               \
    GL_MALLOC(ctx, (void **)&dout.p, (size_t)(out_elems) * 8, "ops");                             \
    GL_TRY(gl_h2d(ctx, din.p, in, (size_t)(in_elems) * 8)); \
    const dim3 grid((unsigned)((n + 255) / 256), (unsigned)n, (unsigned)B), block(256)

#define OPS_EPILOGUE(out_elems)                                                                    \
    GL_HIP(ctx, hipGetLastError());                                                                \
    GL_TRY(gl_d2h(ctx, out, dout.p, (size_t)(out_elems) * 8)); \
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));                                                \
    return GL_OK

GL_API int gl_square_to_vec(gl_ctx *ctx, const double *in, int64_t B, int64_t n, double *out) {
    OPS_PROLOGUE(B * n * n, B * E);
    k_square_to_vec<<<grid, block, 0, ctx->stream>>>(din.p, dout.p, n, E);
    OPS_EPILOGUE(B * E);
}

GL_API int gl_laplacian_from_vec(gl_ctx *ctx, const double *in, int64_t B, int64_t n, double *out) {
    OPS_PROLOGUE(B * E, B * n * n);
    k_lap_offdiag<<<grid, block, 0, ctx->stream>>>(din.p, dout.p, n, E);
    k_lap_diag<<<dim3((unsigned)n, (unsigned)B), block, 0, ctx->stream>>>(dout.p, n);
    OPS_EPILOGUE(B * n * n);
}

GL_API int gl_laplacian_adjoint(gl_ctx *ctx, const double *in, int64_t B, int64_t n, int order, double *out) {
    OPS_PROLOGUE(B * n * n, B * E);
    k_lap_adjoint<<<grid, block, 0, ctx->stream>>>(din.p, dout.p, n, E, order);
    OPS_EPILOGUE(B * E);
}

GL_API int gl_squared_pdiffs(gl_ctx *ctx, const double *in, int64_t B, int64_t n, double *out) {
    OPS_PROLOGUE(B * n, B * E);
    k_sq_pdiffs<<<grid, block, 0, ctx->stream>>>(din.p, dout.p, n, E);
    OPS_EPILOGUE(B * E);
}

GL_API int gl_theta_from_distances(gl_ctx *ctx, const double *zsq, int64_t n, int avg_degree, double *theta_out) {
    if (!ctx) return GL_E_INVALID;
    if (!zsq || !theta_out) return gl_set_error(ctx, GL_E_INVALID, "NULL buffer");
    if (n < 2) return gl_set_error(ctx, GL_E_INVALID, "n must be >= 2");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf dz, dt;
    GL_MALLOC(ctx, (void **)&dz.p, (size_t)n * n * 8, "ops");
    GL_MALLOC(ctx, (void **)&dt.p, (size_t)2 * n * 8, "ops");
    GL_TRY(gl_h2d(ctx, dz.p, zsq, (size_t)n * n * 8));
    return gl_theta_from_square(ctx, dz.p, n, avg_degree, dt.p, theta_out);
}

GL_API int gl_theta_from_matrix(gl_ctx *ctx, const double *zsq, int64_t rows, int64_t cols, int avg_degree,
                                double *theta_out) {
    if (!ctx) return GL_E_INVALID;
    if (!zsq || !theta_out) return gl_set_error(ctx, GL_E_INVALID, "NULL buffer");
    if (rows < 1 || cols < 2) return gl_set_error(ctx, GL_E_INVALID, "need rows >= 1 and cols >= 2");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf dz, dt;
    GL_MALLOC(ctx, (void **)&dz.p, (size_t)rows * cols * 8, "ops");
    GL_MALLOC(ctx, (void **)&dt.p, (size_t)2 * rows * 8, "ops");
    GL_TRY(gl_h2d(ctx, dz.p, zsq, (size_t)rows * cols * 8));
    return gl_theta_from_rows(ctx, dz.p, rows, cols, avg_degree, dt.p, theta_out);
}

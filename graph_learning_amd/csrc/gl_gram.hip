// Squared pairwise distances between the columns of the signal matrix.
//
// Reference: LogModel._initialize, graph_learn/smooth_learning.py:278
//     sq_pdists = pdist(x.T) ** 2            (x is (d samples, n nodes))
// Here: z_ij = G_ii + G_jj - 2 G_ij with the Gram matrix G = Xc^T Xc computed on
// the fp64 matrix cores (v_mfma_f64_16x16x4_f64) -- the one dense contraction of
// the hot path.  Xc = X minus the per-sample mean over nodes: pairwise
// differences are unchanged, but the cancellation in G_ii + G_jj - 2 G_ij is
// taken on centred data.  G is bitwise symmetric and G_ii comes out of the same
// accumulation chain as G_ij, so coincident columns give z = 0 exactly and the
// diagonal is exactly 0.
#include "gl_logdeg.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int GT = 64;   // output tile edge
constexpr int GK = 16;   // samples per LDS stage
constexpr int GLD = 80;  // LDS row stride in doubles (64 + 16: ds_read_b64 conflict-free)

// xc[t][:] = x[t][:] - mean_j x[t][j]   (one workgroup per sample; x itself is kept for the direct differences)
__global__ __launch_bounds__(256) void k_center_rows(const double *__restrict__ x, double *__restrict__ xc, int64_t d,
                                                     int64_t n) {
    __shared__ double s_part[256];
    const int64_t t = blockIdx.x;
    const double *row = x + t * n;
    double acc = 0;
    for (int64_t j = threadIdx.x; j < n; j += 256) acc += row[j];
    s_part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) s_part[threadIdx.x] += s_part[threadIdx.x + s];
        __syncthreads();
    }
    const double mean = s_part[0] / (double)n;
    for (int64_t j = threadIdx.x; j < n; j += 256) xc[t * n + j] = row[j] - mean;
}

// G[I*64.., J*64..] = sum_t x[t][i] x[t][j] for the tile pairs I <= J only (G is symmetric: half the contraction);
// the mirror image G[J*64.., I*64..] is written from the same accumulators, transposed 16 x 16 block by block
// through a wave-private LDS patch so that its stores are row-contiguous too.  4 waves in a 2x2 arrangement,
// each 32x32 outputs = 2x2 MFMA accumulators of 16x16.  Every element, mirrored or not, is the same sum over t in
// the same order: G is bitwise symmetric and G_ii, G_jj, G_ij of coincident columns are equal bit for bit.
__global__ __launch_bounds__(256) void k_gram(const double *__restrict__ x, int64_t d, int64_t n,
                                             double *__restrict__ G, int64_t nb) {
    __shared__ double As[GK][GLD];
    __shared__ double Bs[GK][GLD];
    __shared__ double Ts[4][16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int I, J;
    gl_tile_decode((int64_t)blockIdx.x, nb, I, J);
    const int64_t i0 = (int64_t)I * GT, j0 = (int64_t)J * GT;
    const int lrow = tid >> 4;        // 0..15  sample within the stage
    const int lcol = (tid & 15) * 4;  // 0..60  node within the tile

    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};

    // the next stage's 16 x 64 + 16 x 64 signal values travel from memory into registers while the current stage is
    // multiplied out of LDS (one stage = 16 MFMAs per wave = 1024 matrix-pipe cycles, about one memory round trip)
    double ra[4], rb[4];
    auto fetch = [&](int64_t t0) {
        const int64_t t = t0 + lrow;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t ia = i0 + lcol + e, jb = j0 + lcol + e;
            ra[e] = (t < d && ia < n) ? x[t * n + ia] : 0.0;
            rb[e] = (t < d && jb < n) ? x[t * n + jb] : 0.0;
        }
    };
    fetch(0);
    for (int64_t t0 = 0; t0 < d; t0 += GK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            As[lrow][lcol + e] = ra[e];
            Bs[lrow][lcol + e] = rb[e];
        }
        __syncthreads();
        if (t0 + GK < d) fetch(t0 + GK);
#pragma unroll
        for (int kk = 0; kk < GK / 4; ++kk) {
            const int kr = kk * 4 + (lane >> 4);
            double a[2], b[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                a[m] = As[kr][wm * 32 + m * 16 + (lane & 15)];
                b[m] = Bs[kr][wn * 32 + m * 16 + (lane & 15)];
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();
    }

    // C/D map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t i = i0 + wm * 32 + mi * 16 + (lane >> 4) + 4 * r;
                const int64_t j = j0 + wn * 32 + ni * 16 + (lane & 15);
                if (i < n && j < n) G[i * n + j] = acc[mi][ni][r];
            }
    if (I == J) return;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Ts[wave][(lane >> 4) + 4 * r][lane & 15] = acc[mi][ni][r];
            // the patch is exchanged between the lanes of ONE wave: its stores are ordered before the transposed loads (and
            // those before the next block's stores) explicitly, not by what the compiler happens to assume about aliasing
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double tv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) tv[r] = Ts[wave][lane & 15][(lane >> 4) + 4 * r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // row of the mirror image = a column of the block, column = one of its rows
                const int64_t j = j0 + wn * 32 + ni * 16 + (lane >> 4) + 4 * r;
                const int64_t i = i0 + wm * 32 + mi * 16 + (lane & 15);
                if (i < n && j < n) G[j * n + i] = tv[r];
            }
        }
}

__global__ void k_take_diag(const double *__restrict__ G, int64_t n, double *__restrict__ diag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) diag[i] = G[i * n + i];
}

// in place: G_ij <- max(G_ii + G_jj - 2 G_ij, +0).  The Gram form cancels for near-coincident columns
// (absolute error ~ eps (G_ii + G_jj)), and those nearest-neighbour distances are exactly the ones theta is
// built from; the reference takes direct differences (pdist, smooth_learning.py:278).  Where the result is
// below 2^-10 of G_ii + G_jj it is therefore recomputed as sum_t (x_ti - x_tj)^2 from the ORIGINAL signals, as
// pdist does (rare: d strided loads per such entry), which restores full relative accuracy.
__global__ void k_gram_to_dist(double *__restrict__ G, int64_t n, const double *__restrict__ diag,
                               const double *__restrict__ x, int64_t d) {
    const int64_t i = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) {
        const double s = diag[i] + diag[j];
        double z = s - 2 * G[i * n + j];
        if (i != j && z < s * 0x1p-10) {
            double acc = 0;
            for (int64_t t = 0; t < d; ++t) {
                const double df = x[t * n + i] - x[t * n + j];
                acc += df * df;
            }
            z = acc;
        }
        G[i * n + j] = (i != j && z > 0) ? z : 0.0;
    }
}

}  // namespace

// x_dev (d x n) stays as it is, xc_dev (d x n) receives the centred copy; zsq_dev receives the n x n squared
// distances; diag_dev is n doubles of scratch.  Stream-ordered: no allocation, no synchronisation.
int gl_gram_distances(gl_ctx *ctx, const double *x_dev, double *xc_dev, int64_t d, int64_t n, double *zsq_dev,
                      double *diag_dev) {
    k_center_rows<<<dim3((unsigned)d), dim3(256), 0, ctx->stream>>>(x_dev, xc_dev, d, n);
    const unsigned nb = (unsigned)((n + GT - 1) / GT);
    k_gram<<<dim3(nb * (nb + 1) / 2), dim3(256), 0, ctx->stream>>>(xc_dev, d, n, zsq_dev, (int64_t)nb);
    k_take_diag<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream>>>(zsq_dev, n, diag_dev);
    k_gram_to_dist<<<dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, ctx->stream>>>(zsq_dev, n, diag_dev, x_dev,
                                                                                                  d);
    GL_HIP(ctx, hipGetLastError());
    return GL_OK;
}

// Brackets of the largest eigenvalue of a small Gram matrix on the device (the coupled stopping rule of the batched
// solver, smooth_learning.py:158-159: np.linalg.norm(M, ord=2)^2 of an (E x B) matrix is lambda_max of its B x B Gram matrix).
#pragma once

#include "gl_lanes.h"

// One wave: rigorous bracket [lo, hi] of the largest eigenvalue of the symmetric positive semi-definite
// B x B matrix A (LDS, destroyed), B <= 32.  With A_0 = A / tr A and A_{k+1} = A_k^2 / t_k, t_k = tr A_k^2:
// mu_k = lambda_max(A_k) obeys mu_k = sqrt(t_k mu_{k+1}), and a trace-1 PSD matrix has tr A^2 <= mu <= sqrt(tr A^2).
// So log mu_0 = sum_k 2^-(k+1) log t_k + 2^-m log mu_m with mu_m in [t_m, sqrt t_m]: the width of the bracket
// is 2^-(m+1) |log t_m| <= 2^-(m+1) log B in the logarithm.  Stops early once A_k is rank one (t_k -> 1).
// The same in stages (the bracket halves its logarithmic width with every squaring, and most decisions need a few):
// sm_bracket_begin, then sm_bracket_advance (up to `steps` more squarings) and sm_bracket_bounds (reads A only) as often
// as needed.
struct SmBracket {
    double tr, L, wgt;
    int k;        // squarings done
    bool done;    // rank one reached (or the 34 squarings of the full bracket): no further step changes anything
};
template <int EPL>  // matrix entries per lane: 4 (B <= 16) or 16 (B <= 32)
__device__ __forceinline__ void sm_bracket_begin(double *A, int B, int lane, SmBracket &S) {
    const int nn = B * B;
    double tr = 0.0;
    for (int i = lane; i < B; i += 64) tr += A[i * B + i];
    tr = wave_sum(tr);
    S.tr = tr, S.L = 0.0, S.wgt = 0.5, S.k = 0, S.done = !(tr > 0.0);
    if (S.done) return;
    for (int idx = lane; idx < nn; idx += 64) A[idx] = A[idx] / tr;
}
template <int EPL>
__device__ __forceinline__ void sm_bracket_advance(double *A, int B, int lane, SmBracket &S, int steps) {
    const int nn = B * B;
    for (int q = 0; q < steps && !S.done; ++q) {
        double sv[EPL], dsum = 0.0;
#pragma unroll
        for (int r = 0; r < EPL; ++r) {
            const int idx = lane + 64 * r;
            sv[r] = 0.0;
            if (idx < nn) {
                const int i = idx / B, j = idx - i * B;
                double acc = 0.0;
                for (int l = 0; l < B; ++l) acc += A[i * B + l] * A[l * B + j];
                sv[r] = acc;
                if (i == j) dsum += acc;
            }
        }
        const double t = wave_sum(dsum);  // tr A_k^2, in (1/B, 1]
#pragma unroll
        for (int r = 0; r < EPL; ++r)
            if (lane + 64 * r < nn) A[lane + 64 * r] = sv[r] / t;
        S.L += S.wgt * log(t);
        S.wgt *= 0.5;
        ++S.k;
        if (t > 1.0 - 1e-13 || S.k >= 34) S.done = true;
    }
}
template <int EPL>
__device__ __forceinline__ void sm_bracket_bounds(const double *A, int B, int lane, const SmBracket &S, double &lo, double &hi) {
    if (!(S.tr > 0.0)) {
        lo = hi = 0.0;
        return;
    }
    const int nn = B * B;
    double fro = 0.0;
    for (int idx = lane; idx < nn; idx += 64) fro += A[idx] * A[idx];
    const double tm = fmin(wave_sum(fro), 1.0);  // tr A_m^2 (A_m symmetric)
    const double lt = log(tm);                   // <= 0
    lo = S.tr * exp(S.L + 2.0 * S.wgt * lt) * (1.0 - 1e-12);
    hi = S.tr * exp(S.L + S.wgt * lt) * (1.0 + 1e-12);
}
template <int EPL>
__device__ __forceinline__ void sm_lambda_bracket(double *A, int B, int lane, double &lo, double &hi) {
    SmBracket S;
    sm_bracket_begin<EPL>(A, B, lane, S);
    sm_bracket_advance<EPL>(A, B, lane, S, 34);
    sm_bracket_bounds<EPL>(A, B, lane, S, lo, hi);
}

// The same stages with the squaring on the matrix cores (one wave, v_mfma_f64_16x16x4_f64): the matrix padded with zeros to
// 16 NB16 rows of LD = 16 NB16 + 4 doubles (operand reads then meet 2-way bank conflicts at most).  A squaring is 4 NB16^3
// instructions of 16 x 16 x 4 -- 1 us at B = 20, 0.15 us at B <= 16 -- where the entry-by-entry form above takes 3.5 / 0.7.
template <int NB16>
struct SmMfma {
    static constexpr int N = 16 * NB16, LD = N + 4, SIZE = N * LD;
    typedef double d4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ d4 mfma(double a, double b, d4 c) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
#else
        return c;
#endif
    }
    static __device__ __forceinline__ void begin(double *A, int lane, SmBracket &S) {
        double tr = 0.0;
        for (int i = lane; i < N; i += 64) tr += A[i * LD + i];
        tr = wave_sum(tr);
        S.tr = tr, S.L = 0.0, S.wgt = 0.5, S.k = 0, S.done = !(tr > 0.0);
        if (S.done) return;
        // (a reciprocal instead of SIZE / 64 divisions per lane: the scaling is off by an ulp, twelve orders below the
        //  bracket's safety factors, and so is the one by 1 / t below)
        const double rt = 1.0 / tr;
        for (int idx = lane; idx < SIZE; idx += 64) A[idx] = A[idx] * rt;
    }
    static __device__ __forceinline__ void advance(double *A, int lane, SmBracket &S, int steps) {
        const int r = lane & 15, q4 = lane >> 4;
        for (int q = 0; q < steps && !S.done; ++q) {
            d4 acc[NB16][NB16];
#pragma unroll
            for (int bi = 0; bi < NB16; ++bi)
#pragma unroll
                for (int bj = 0; bj < NB16; ++bj) acc[bi][bj] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
            for (int ks = 0; ks < N / 4; ++ks) {
                double a[NB16], b[NB16];
#pragma unroll
                for (int bi = 0; bi < NB16; ++bi) a[bi] = A[(16 * bi + r) * LD + 4 * ks + q4], b[bi] = A[(4 * ks + q4) * LD + 16 * bi + r];
#pragma unroll
                for (int bi = 0; bi < NB16; ++bi)
#pragma unroll
                    for (int bj = 0; bj < NB16; ++bj) acc[bi][bj] = mfma(a[bi], b[bj], acc[bi][bj]);
            }
            double dsum = 0.0;  // the trace of A^2: entries (row (lane >> 4) + 4 e, column lane & 15) of the diagonal blocks
#pragma unroll
            for (int bi = 0; bi < NB16; ++bi)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (q4 + 4 * e == r) dsum += acc[bi][bi][e];
            const double t = wave_sum(dsum);  // in (1/B, 1]
            const double rt = 1.0 / t;
#pragma unroll
            for (int bi = 0; bi < NB16; ++bi)
#pragma unroll
                for (int bj = 0; bj < NB16; ++bj)
#pragma unroll
                    for (int e = 0; e < 4; ++e) A[(16 * bi + q4 + 4 * e) * LD + 16 * bj + r] = acc[bi][bj][e] * rt;
            S.L += S.wgt * log(t);
            S.wgt *= 0.5;
            ++S.k;
            if (t > 1.0 - 1e-13 || S.k >= 34) S.done = true;
        }
    }
    static __device__ __forceinline__ void bounds(const double *A, int lane, const SmBracket &S, double &lo, double &hi) {
        if (!(S.tr > 0.0)) {
            lo = hi = 0.0;
            return;
        }
        double fro = 0.0;
        for (int idx = lane; idx < SIZE; idx += 64) fro += A[idx] * A[idx];
        const double tm = fmin(wave_sum(fro), 1.0);  // tr A_m^2 (A_m symmetric; the padding is zero)
        const double lt = log(tm);                   // <= 0
        lo = S.tr * exp(S.L + 2.0 * S.wgt * lt) * (1.0 - 1e-12);
        hi = S.tr * exp(S.L + S.wgt * lt) * (1.0 + 1e-12);
    }
};

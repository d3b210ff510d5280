// Graph-dictionary model with log-barrier degrees (GraphDictLog), the "next" row f-1 of SURVEY.md
// section 8.  Reference: graph_learn/dictionary/base_model.py:355-527 (_update_coefficients,
// _update_weights, _fit_step, _single_fit), dictionary/log_model.py:27-50 (adjoints, dual prox).
//
// T samples on n nodes, K atoms.  The reference materialises the (T x E) matrix of squared
// pairwise differences D[t,e] = (x_ti - x_tj)^2 (17 GB at n = 1024, T = 4096) and runs two GEMMs
// per step against it:  W @ D' (K x T, coefficient gradient) and C @ D (K x E, weight gradient).
// With K <= 16 these are not GEMM-shaped work: forming D[t,e] on the fly costs 2 flops against
// 2 K flops of use, and nothing has to be stored.  Two kernels do that:
//
//   k_gd_grad_c   lane <-> sample (four per lane), wave walks pairs (i, j): x_i, x_j come from the
//                 transposed signal matrix XT (n x T, coalesced), the K weights of the pair are
//                 wave-uniform -> scalar loads from the pair-major weight array WS[i][j][K].
//   k_gd_grad_w   lane <-> column j, wave holds a strip of rows i: x_j from X (T x n, coalesced),
//                 x_i and the K coefficients of sample t are wave-uniform -> scalar loads.
//
// Both are bound by the fp64 vector rate (2 + 2K lane-ops per pair-sample and kernel), not by
// HBM: X / XT (33 MB at the BASELINE size) stay in L2 / Infinity Cache.  Everything else of a
// step (adjoint terms through the closed form of S / S', projections, dual prox, relative
// changes) is O(K E + K T + T n) and runs in a handful of small kernels.
//
// Weights live in WS: n_pad x n_pad x KP doubles, entry (i, j, k) for i < j < n, zero elsewhere
// (the zero lower triangle doubles as the mask of the pair loops).  KP = K rounded up to 4/8/16.
#include "gl_common.h"
#include "gl_graphdict.h"
#include "gl_lanes.h"

#include <algorithm>
#include <limits>
#include <new>
#include <vector>

#ifndef GD_UPD_EXP
#define GD_UPD_EXP 0  // tools/build_variant_gd.sh: ablations of k_gd_update2 (timing only)
#endif
#ifndef GD_GC_EXP
#define GD_GC_EXP 0  // tools/build_variant_gd.sh: ablations of k_gd_grad_c (timing only): 1 = x_j always from the strip's first two columns, 2 = the weights, 3 = both
#endif
#ifndef GD_XD32_DEFAULT
#define GD_XD32_DEFAULT 4  // k_gd_grad_c<float, 8, 2>: columns of x_j in flight (GRAPHLEARN_AMD_GD_XD = 2 | 4 overrides; fp64: 2)
#endif

namespace {

__device__ __forceinline__ double gd_wave_sum(double v) { return wave_sum(v); }  // gl_lanes.h
__device__ __forceinline__ double gd_wave_max(double v) { return wave_max(v); }  // gl_lanes.h (DPP)

// ------------------------------------------------------------------------------------------
// coefficient gradient: SC[g][k][t] = sum over the pairs of strip group g of W[k,(i,j)] (x_ti - x_tj)^2
// (base_model.py:375 `self.weights_ @ sq_pdiffs.T`).  grid (Tp / 128, NG), one wave per workgroup.
// T = double, or float for the fp32 form of BASELINE configs[4] (gl_graphdict_set_precision): operands and the
// running sums of a (strip group, sample) cell in fp32 -- the compiler packs the K-loop into v_pk_fma_f32, two
// fused multiply-adds per lane and instruction -- the cell's result leaves as fp64 and every fold after it is fp64.
template <typename T>
__device__ __forceinline__ T gd_fma(T a, T b, T c);
template <>
__device__ __forceinline__ double gd_fma<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <>
__device__ __forceinline__ float gd_fma<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// squared differences of a pair: one packed subtraction and one packed multiplication in fp32 (the compiler packs the
// fused multiply-adds by itself, not these two), the plain operations in fp64 -- the same values either way
typedef float gd_v2f __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ void gd_sqdiff2(const T a0, const T a1, const T b0, const T b1, T &s0, T &s1) {
    const T d0 = a0 - b0, d1 = a1 - b1;
    s0 = d0 * d0, s1 = d1 * d1;
}
template <>
__device__ __forceinline__ void gd_sqdiff2<float>(const float a0, const float a1, const float b0, const float b1, float &s0, float &s1) {
    const gd_v2f d = gd_v2f{a0, a1} - gd_v2f{b0, b1};
    const gd_v2f q = d * d;
    s0 = q.x, s1 = q.y;
}

// XD: columns of x_j in flight -- 2: one operand set ahead, together with the weights; 4: the x_j of three columns ahead.
// A column's accumulation is ~0.1-0.3 us of a SIMD that 4-7 waves share, and X' (16 / 32 MB) does not fit an XCD's L2: with one
// column of lead the loads are waited for.  Measured at K = 8, n = 1024, T = 4096, same GPU (tools/gd_gradc_ab.sh): fp32
// 0.435-0.447 -> 0.405-0.412 ms although the 16 registers cost two waves of occupancy (7 -> 5); fp64 (4 waves either way)
// 0.728 -> 0.743: there the loads are bandwidth, not latency -- without them (-DGD_GC_EXP=1) the kernel takes 0.59 ms in fp64,
// 0.365 in fp32.  Same sums in the same order for every XD.
template <typename T, int KP, int RC, int XD = 2>
__global__ __launch_bounds__(64) void k_gd_grad_c(const T *__restrict__ XT, const T *__restrict__ WS,
                                                  double *__restrict__ SC, int n, int n_pad, int Tp, int NG,
                                                  const double *__restrict__ done) {
    if (done[0] != 0.0) return;  // the stopping rule was met earlier in this chunk of steps
    const int lane = threadIdx.x;
    const int g = blockIdx.y;
    const int64_t t0 = (int64_t)blockIdx.x * GD_SB + lane;  // this lane's samples: t0 + 64 q, q < GD_SPL
    T acc[GD_SPL][KP];
#pragma unroll
    for (int q = 0; q < GD_SPL; ++q)
#pragma unroll
        for (int k = 0; k < KP; ++k) acc[q][k] = (T)0;
    const int nstrips = (n + RC - 1) / RC;
    // strip s has n - RC s - 1 columns to walk: dealing the strips out in serpentine order
    // (g, 2 NG - 1 - g, 2 NG + g, ...) gives every group the same amount of work
    for (int r = 0; r * NG < nstrips; ++r) {
        const int s = r * NG + ((r & 1) ? NG - 1 - g : g);
        if (s >= nstrips) continue;
        const int i0 = RC * s;
        if (i0 + 1 >= n) continue;
        T xi[RC][GD_SPL];
#pragma unroll
        for (int a = 0; a < RC; ++a)
#pragma unroll
            for (int q = 0; q < GD_SPL; ++q) xi[a][q] = XT[(int64_t)(i0 + a) * Tp + t0 + 64 * q];
        // software pipeline over the columns j, two operand sets (see k_gd_grad_w); rows i0 + a >= j
        // carry zeros in WS (lower triangle): no branch needed
        T wA[RC][KP], xjA[GD_SPL], wB[RC][KP], xjB[GD_SPL];
        auto request = [&](int j, T (&w)[RC][KP], T (&xj)[GD_SPL]) {
            const int jc = j < n ? j : n - 1;
            const int jf = i0 + 1 + (jc & 1) < n ? i0 + 1 + (jc & 1) : n - 1;  // (ablations: rows / columns inside the matrices)
            const int jw = (GD_GC_EXP & 2) ? jf : jc, jx = (GD_GC_EXP & 1) ? jf : jc;
#pragma unroll
            for (int a = 0; a < RC; ++a) {
                const T *__restrict__ wp = WS + ((int64_t)(i0 + a) * n_pad + jw) * KP;  // wave-uniform
#pragma unroll
                for (int k = 0; k < KP; ++k) w[a][k] = wp[k];
            }
#pragma unroll
            for (int q = 0; q < GD_SPL; ++q) xj[q] = XT[(int64_t)jx * Tp + t0 + 64 * q];
        };
        auto accumulate = [&](const T (&w)[RC][KP], const T (&xj)[GD_SPL]) {
            static_assert(GD_SPL % 2 == 0, "samples per lane are taken in pairs");
#pragma unroll
            for (int a = 0; a < RC; ++a)
#pragma unroll
                for (int q = 0; q < GD_SPL; q += 2) {
                    T sq0, sq1;
                    gd_sqdiff2<T>(xi[a][q], xi[a][q + 1], xj[q], xj[q + 1], sq0, sq1);
#pragma unroll
                    for (int k = 0; k < KP; ++k) acc[q][k] = gd_fma<T>(w[a][k], sq0, acc[q][k]);
#pragma unroll
                    for (int k = 0; k < KP; ++k) acc[q + 1][k] = gd_fma<T>(w[a][k], sq1, acc[q + 1][k]);
                }
        };
        int j = i0 + 1;
        if constexpr (XD == 4) {
            // x_j three columns ahead in four rotating sets, the weights one column ahead as below (scalar cache)
            T xq[4][GD_SPL];
            auto request_w = [&](int jj, T (&w)[RC][KP]) {
                const int jc = jj < n ? jj : n - 1;
#pragma unroll
                for (int a = 0; a < RC; ++a) {
                    const T *__restrict__ wp = WS + ((int64_t)(i0 + a) * n_pad + jc) * KP;  // wave-uniform
#pragma unroll
                    for (int k = 0; k < KP; ++k) w[a][k] = wp[k];
                }
            };
            auto request_x = [&](int jj, T (&xj)[GD_SPL]) {
                const int jc = jj < n ? jj : n - 1;
#pragma unroll
                for (int q = 0; q < GD_SPL; ++q) xj[q] = XT[(int64_t)jc * Tp + t0 + 64 * q];
            };
            request_x(j, xq[0]);
            request_x(j + 1, xq[1]);
            request_x(j + 2, xq[2]);
            request_w(j, wA);
            for (; j + 3 < n; j += 4) {
                __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): weights A are in
                request_w(j + 1, wB);
                request_x(j + 3, xq[3]);
                __builtin_amdgcn_sched_barrier(0);
                accumulate(wA, xq[0]);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC07F);
                request_w(j + 2, wA);
                request_x(j + 4, xq[0]);
                __builtin_amdgcn_sched_barrier(0);
                accumulate(wB, xq[1]);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC07F);
                request_w(j + 3, wB);
                request_x(j + 5, xq[1]);
                __builtin_amdgcn_sched_barrier(0);
                accumulate(wA, xq[2]);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC07F);
                request_w(j + 4, wA);
                request_x(j + 6, xq[2]);
                __builtin_amdgcn_sched_barrier(0);
                accumulate(wB, xq[3]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // up to three columns left: the x of j, j + 1, j + 2 in xq[0 .. 2], the weights of j in A
            if (j < n) {
                __builtin_amdgcn_s_waitcnt(0xC07F);
                request_w(j + 1, wB);
                accumulate(wA, xq[0]);
            }
            if (j + 1 < n) {
                __builtin_amdgcn_s_waitcnt(0xC07F);
                request_w(j + 2, wA);
                accumulate(wB, xq[1]);
            }
            if (j + 2 < n) {
                __builtin_amdgcn_s_waitcnt(0xC07F);
                accumulate(wA, xq[2]);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);  // (nothing scalar pending when the next strip's sets are requested)
            continue;
        }
        request(j, wA, xjA);
        for (; j + 1 < n; j += 2) {
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): set A is in
            request(j + 1, wB, xjB);
            __builtin_amdgcn_sched_barrier(0);
            accumulate(wA, xjA);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);  // set B is in
            request(j + 2, wA, xjA);
            __builtin_amdgcn_sched_barrier(0);
            accumulate(wB, xjB);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (j < n) accumulate(wA, xjA);
    }
#pragma unroll
    for (int q = 0; q < GD_SPL; ++q)
#pragma unroll
        for (int k = 0; k < KP; ++k) SC[((int64_t)g * KP + k) * Tp + t0 + 64 * q] = (double)acc[q][k];
}

// ------------------------------------------------------------------------------------------
// weight gradient: GS[ts][i][j][k] = sum over the samples of split ts of C[k,t] (x_ti - x_tj)^2
// (base_model.py:447 `self.coefficients_ @ sq_pdiffs`).  1-D grid over the compact list of
// (row strip, column block) tasks that contain a pair i < j, times TS sample splits: workgroup ids
// go round-robin to the 8 XCDs, so a 3-D grid with the dead below-diagonal blocks left in gives
// XCD 7 2.4x the work of XCD 0 (measured: the kernel then runs as long as its slowest XCD).
template <typename T, int KP, int SW>
__global__ __launch_bounds__(256, 3) void k_gd_grad_w(const T *__restrict__ X, const T *__restrict__ CT,
                                                   double *__restrict__ GS, const int2 *__restrict__ tasks, int ntasks,
                                                   int n_pad, int Tp, int tchunk, double *dbg,
                                                   const double *__restrict__ done) {
    if (done[0] != 0.0) return;
    // the four waves of a workgroup take a quarter of the split's samples each and add their
    // partial sums through LDS: finer tasks (even load per SIMD) without more slabs in HBM
    __shared__ double2 s_red[3][KP / 2][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: keeps the operand loads scalar
    const int2 task = tasks[blockIdx.x % ntasks];
    const int jb = task.y, i0 = task.x * SW, ts = blockIdx.x / ntasks;
    const int j = jb * 64 + lane;
    T acc[SW][KP];
#pragma unroll
    for (int a = 0; a < SW; ++a)
#pragma unroll
        for (int k = 0; k < KP; ++k) acc[a][k] = (T)0;
    const int tA0 = ts * tchunk, tB0 = min(Tp, tA0 + tchunk);
    const int sub = (max(tB0 - tA0, 0) + 3) / 4;
    const int tA = tA0 + wave * sub, tB = min(tB0, tA + sub);
    const unsigned long long rt0 = dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // Software pipeline over the samples: the wave-uniform operands of sample t + 1 (SW node values,
    // KP coefficients: scalar loads, which return out of order and can only be waited for all
    // together) are requested right after the wait for those of sample t and land while t computes.
    // Two operand sets alternate (no register copies); sched_barrier keeps the requests ahead of
    // the arithmetic (left alone, the scheduler sinks them into the middle of the FMA block).
    if (tA < tB) {
        T xiA[SW], cfA[KP], xjA, xiB[SW], cfB[KP], xjB;
        auto request = [&](int t, T (&xi)[SW], T (&cf)[KP], T &xj) {
            const int tc = t < tB ? t : tB - 1;
            const T *__restrict__ xr = X + (int64_t)tc * n_pad + i0;  // wave-uniform
            const T *__restrict__ cr = CT + (int64_t)tc * KP;         // wave-uniform
#pragma unroll
            for (int a = 0; a < SW; ++a) xi[a] = xr[a];
#pragma unroll
            for (int k = 0; k < KP; ++k) cf[k] = cr[k];
            xj = X[(int64_t)tc * n_pad + j];
        };
        auto accumulate = [&](const T (&xi)[SW], const T (&cf)[KP], const T xj) {
            static_assert(SW % 2 == 0, "rows of a strip are taken in pairs");
#pragma unroll
            for (int a = 0; a < SW; a += 2) {
                T sq0, sq1;
                gd_sqdiff2<T>(xi[a], xi[a + 1], xj, xj, sq0, sq1);
#pragma unroll
                for (int k = 0; k < KP; ++k) acc[a][k] = gd_fma<T>(cf[k], sq0, acc[a][k]);
#pragma unroll
                for (int k = 0; k < KP; ++k) acc[a + 1][k] = gd_fma<T>(cf[k], sq1, acc[a + 1][k]);
            }
        };
        request(tA, xiA, cfA, xjA);
        int t = tA;
        for (; t + 1 < tB; t += 2) {
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): set A is in
            request(t + 1, xiB, cfB, xjB);
            __builtin_amdgcn_sched_barrier(0);
            accumulate(xiA, cfA, xjA);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);  // set B is in
            request(t + 2, xiA, cfA, xjA);
            __builtin_amdgcn_sched_barrier(0);
            accumulate(xiB, cfB, xjB);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (t < tB) accumulate(xiA, cfA, xjA);
    }
    if (dbg && lane == 0) {  // tuning aid (GRAPHLEARN_AMD_GD_PROBE=1): per-wave (start, end) in 10 ns ticks
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
        const int64_t w = (int64_t)blockIdx.x * 4 + wave;
        dbg[4 * w + 0] = (double)rt0;
        dbg[4 * w + 1] = (double)r1;
        dbg[4 * w + 2] = (double)__builtin_amdgcn_s_getreg(63492);          // HW_REG_HW_ID
        dbg[4 * w + 3] = (double)(__builtin_amdgcn_s_getreg(63508) & 15u);  // XCC_ID
    }
    // ((wave 0 + wave 1) + wave 2) + wave 3 in fp64, one row of the strip at a time (12 KB of LDS)
#pragma unroll
    for (int a = 0; a < SW; ++a) {
        if (wave > 0) {
#pragma unroll
            for (int k = 0; k < KP; k += 2) s_red[wave - 1][k / 2][lane] = double2{(double)acc[a][k], (double)acc[a][k + 1]};
        }
        __syncthreads();
        if (wave == 0) {
            double fold[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) fold[k] = (double)acc[a][k];
#pragma unroll
            for (int v = 0; v < 3; ++v)
#pragma unroll
                for (int k = 0; k < KP; k += 2) {
                    const double2 p = s_red[v][k / 2][lane];
                    fold[k] += p.x;
                    fold[k + 1] += p.y;
                }
            // GS[ts][i][column block][k / 2][lane][2]: a store instruction writes 1 KB in one piece (round 4: pair-major, every
            // instruction wrote 16 bytes into each of 64 lines -- a slab's 67 MB went out, and came back in, as quarter lines)
            double *dst = GS + ((((int64_t)ts * n_pad + (i0 + a)) * (n_pad / 64) + jb) * (KP / 2)) * 128 + 2 * lane;
#pragma unroll
            for (int k = 0; k < KP; k += 2) *reinterpret_cast<double2 *>(dst + (k / 2) * 128) = double2{fold[k], fold[k + 1]};
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// log_c reset (base_model.py:385-387): columns of the strided view whose sum is < 1e-8
__global__ void k_gd_logc_reset(GdDev P) {
    if (P.stat[5] != 0.0) return;
    const int tw = blockIdx.x * blockDim.x + threadIdx.x;
    if (tw >= P.Tw) return;
    const int64_t t = (int64_t)tw * P.W;
    double s = 0;
    for (int k = 0; k < P.K; ++k) s += P.C[(int64_t)k * P.T + t];
    if (s < 1e-8)
        for (int k = 0; k < P.K; ++k) {
            P.C[(int64_t)k * P.T + t] = P.log_c;
            P.CT[t * P.KP + k] = P.log_c;
            if (P.CTf) P.CTf[t * P.KP + k] = (float)P.log_c;
        }
}

// sum of the strip-group partials of k_gd_grad_c, fixed order: 4 interleaved partial sums per
// element (threadIdx.y), then ((p0 + p1) + p2) + p3
__device__ __forceinline__ void gd_sc_reduce_body(const GdDev &P, const int bx) {
    const double done_flag = P.stat[5];  // ("done" is looked at where the first store would be: tested up front, every wave of the tail spent a round trip on it before its first load)
    __shared__ double s_p[4][64];
    const int64_t q = (int64_t)bx * 64 + threadIdx.x;  // (k, t)
    const int gy = threadIdx.y;
    double v = 0.0;
    if (q < (int64_t)P.KP * P.Tp) {
#pragma unroll 16
        for (int g = gy; g < P.NG; g += 4) v += P.SC[(int64_t)g * P.KP * P.Tp + q];
    }
    s_p[gy][threadIdx.x] = v;
    __syncthreads();
    if (done_flag != 0.0) return;
    if (gy == 0 && q < (int64_t)P.KP * P.Tp)
        P.SCR[q] = ((s_p[0][threadIdx.x] + s_p[1][threadIdx.x]) + s_p[2][threadIdx.x]) + s_p[3][threadIdx.x];
}

// ------------------------------------------------------------------------------------------
// coefficient update (base_model.py:373-401), one wave per window
// maximum over an aligned group of KP lanes (KP = 4, 8, 16), in every lane of the group: DPP
template <int KP>
__device__ __forceinline__ double gd_group_max(double v) {
    v = fmax(v, dpp_mov<DPP_QUAD_XOR1>(v));
    v = fmax(v, dpp_mov<DPP_QUAD_XOR2>(v));
    if (KP >= 8) v = fmax(v, dpp_mov<DPP_HALF_MIRROR>(v));
    if (KP >= 16) v = fmax(v, dpp_mov<DPP_ROW_ROR8>(v));
    return v;
}

// NW windows per wave (tw0 .. tw0 + NW - 1): the degrees, which every window's adjoint term reads in full (KP x n: 64 KB at
// n = 1024 -- one window per wave, that was 295 MB through L2 for a step's 4096 windows, 23 us of k_gd_update2 by ablation),
// are fetched once for the NW of them; every sum keeps its order (a lane adds its nodes in ascending order).
template <int KP, int NW>
__device__ __forceinline__ void gd_coef_update_body(const GdDev &P, const int tw0, const int lane) {
    static_assert(NW * KP <= 64, "a lane per (window, atom) finishes the update");
    const double done_flag = P.stat[5];  // ("done" is looked at where the first store would be: tested up front, every wave of the tail spent a round trip on it before its first load)
    // adjoint term  W @ (S' dual')  ==  <S W_k, dual_tw>   (log_model.py:33-34)
    double dt[NW][KP];
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int k = 0; k < KP; ++k) dt[w][k] = 0.0;
    // (the loads of G x 64 nodes in flight: one node group per round trip it was n / 64 dependent round trips a wave)
    constexpr int G = KP * NW >= 32 ? 2 : 4;
    for (int i0 = 0; i0 < P.n; i0 += 64 * G) {
        double dv[G][NW], dg[G][KP];
#pragma unroll
        for (int r = 0; r < G; ++r) {
            const int i = i0 + r * 64 + lane, ic = min(i, P.n - 1);
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const double v = P.dual[(int64_t)min(tw0 + w, P.Tw - 1) * P.n + ic];
                dv[r][w] = (i < P.n && tw0 + w < P.Tw) ? v : 0.0;  // (past the last node / window: exact zeros, added to sums that x + 0.0 leaves alone)
            }
#pragma unroll
            for (int k = 0; k < KP; ++k) dg[r][k] = P.deg[(int64_t)k * P.n_pad + ic];  // rows k >= K are zero
        }
#pragma unroll
        for (int r = 0; r < G; ++r)
#pragma unroll
            for (int w = 0; w < NW; ++w)
#pragma unroll
                for (int k = 0; k < KP; ++k) dt[w][k] += dg[r][k] * dv[r][w];
    }
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int k = 0; k < KP; ++k) dt[w][k] = gd_wave_sum(dt[w][k]);
    if (done_flag != 0.0) return;
    // lane w * KP + k finishes atom k of window tw0 + w
    double dterm = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int k = 0; k < KP; ++k) dterm = lane == w * KP + k ? dt[w][k] : dterm;
    const int k = lane % KP, tw = tw0 + lane / KP;
    const bool mine = lane < NW * KP && tw < P.Tw;
    double cmax = 0.0;
    if (mine && k < P.K) {
        // smoothness term: mean over the window's samples (:375-379) of the summed strip-group partials
        double tot = 0.0;
        for (int q = 0; q < P.W; ++q) tot += P.SCR[(int64_t)k * P.Tp + (int64_t)tw * P.W + q];
        double grad = P.W > 1 ? tot / (double)P.W : tot;
        const double *c = P.C + (int64_t)k * P.T;  // the window view is c[m * W]
        grad += P.l1_c * (double)P.T;
        if (P.l1_diff_c > 0) {  // :389-394, diff(sign(diff(c)), prepend=0, append=0)[tw]
            auto sgn = [](double v) { return v > 0 ? 1.0 : (v < 0 ? -1.0 : 0.0); };
            const double right = tw + 1 < P.Tw ? sgn(c[(int64_t)(tw + 1) * P.W] - c[(int64_t)tw * P.W]) : 0.0;
            const double left = tw > 0 ? sgn(c[(int64_t)tw * P.W] - c[(int64_t)(tw - 1) * P.W]) : 0.0;
            grad -= P.l1_diff_c * (double)P.T * (right - left);
        }
        double cn = c[(int64_t)tw * P.W] - P.step_c / (double)P.T * (dterm + grad);  // :400
        cn = cn < 0 ? 0.0 : cn;
        cn = cn > 1 ? 1.0 : cn;
        P.Cw[(int64_t)k * P.Tw + tw] = cn;
        cmax = fabs(cn);
    }
    cmax = gd_group_max<KP>(cmax);  // over the window's KP lanes (those past K hold 0)
    if (mine && k == 0) P.cpart[tw] = cmax;
}

// all-zero guard (:406-408); one workgroup
__device__ __forceinline__ void gd_coef_finish_body(const GdDev &P) {
    if (P.stat[5] != 0.0) return;
    __shared__ double s_a[256];
    const int tid = threadIdx.x;
    double mx = 0.0;
    for (int tw = tid; tw < P.Tw; tw += 256) mx = fmax(mx, P.cpart[tw]);
    s_a[tid] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) s_a[tid] = fmax(s_a[tid], s_a[tid + o]);
        __syncthreads();
    }
    if (s_a[0] <= 1e-8) {  // np.allclose(c, 0)
        for (int64_t q = tid; q < (int64_t)P.K * P.Tw; q += 256) P.Cw[q] = P.log_c;
        if (tid == 0) P.stat[2] = 1.0;
    }
}

// ------------------------------------------------------------------------------------------
// M[k][i] = sum_t C[k,t] dual[t / W, i]  (log_model.py:36-37: C @ (S' dual_rep')' = S'(C @ dual_rep) pairwise)
template <int KP>
__global__ __launch_bounds__(256) void k_gd_m(GdDev P, int tchunk) {
    const double done_flag = P.stat[5];
    // one wave per (64 nodes, sample chunk), four chunks per workgroup.  Per 32 samples a wave stages their coefficients
    // (sample-major: one contiguous piece of CT) in its own corner of LDS and reads them back as broadcasts; eight
    // samples' loads of the dual variable in flight, the K sums in registers, no branch inside.  (Round 4: before, a
    // runtime K indexed the sums in scratch; a first templated form came out as a branch and a vmcnt(0) per
    // (sample, atom): 35 us for 33.5 MB either way.)
    __shared__ double s_ct[4][32 * KP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane, ms = blockIdx.y * 4 + wv;
    if (ms >= P.MS) return;  // (whole waves; no workgroup barrier below)
    const int ii = min(i, P.n - 1);
    double acc[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) acc[k] = 0.0;
    const int tA = min(P.T, ms * tchunk), tB = min(P.T, tA + tchunk);
    const double *__restrict__ dual = P.dual;
    const double *__restrict__ CT = P.CT;
    double *ct = s_ct[wv];
    for (int t0 = tA; t0 < tB; t0 += 32) {
        const int cnt = min(32, tB - t0) * KP;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < KP / 2; ++r) {
            const int idx = r * 64 + lane;
            const double v = idx < cnt ? CT[(int64_t)t0 * KP + idx] : 0.0;
            ct[idx] = (idx & (KP - 1)) < P.K ? v : 0.0;  // samples past the chunk and atoms past K add exact zeros
        }
        __builtin_amdgcn_wave_barrier();
        for (int j0 = 0; j0 < 32 && t0 + j0 < tB; j0 += 8) {
            double dl[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = t0 + j0 + j;
                const double v = dual[(int64_t)(min(t, tB - 1) / P.W) * P.n + ii];
                dl[j] = t < tB ? v : 0.0;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int k = 0; k < KP; ++k) acc[k] += ct[(j0 + j) * KP + k] * dl[j];
            }
        }
    }
    if (done_flag != 0.0) return;
#pragma unroll
    for (int k = 0; k < KP; ++k) P.MSL[((int64_t)ms * KP + k) * P.n_pad + i] = i < P.n ? acc[k] : 0.0;
}
__device__ __forceinline__ void gd_m_reduce_body(const GdDev &P, const int bx) {
    const double done_flag = P.stat[5];
    __shared__ double s_p[4][64];
    const int64_t q = (int64_t)bx * 64 + threadIdx.x;
    const int gy = threadIdx.y;
    double v = 0.0;
    if (q < (int64_t)P.KP * P.n_pad) {
#pragma unroll 8
        for (int ms = gy; ms < P.MS; ms += 4) v += P.MSL[(int64_t)ms * P.KP * P.n_pad + q];
    }
    s_p[gy][threadIdx.x] = v;
    __syncthreads();
    if (done_flag != 0.0) return;
    if (gy == 0 && q < (int64_t)P.KP * P.n_pad)
        P.M[q] = ((s_p[0][threadIdx.x] + s_p[1][threadIdx.x]) + s_p[2][threadIdx.x]) + s_p[3][threadIdx.x];
}

// ------------------------------------------------------------------------------------------
// weight update (base_model.py:444-461): one wave per (row i, column block); in place
template <int KP>
__device__ __forceinline__ void gd_weight_update_body(const GdDev &P, const int jb, const int i, const int nbj, const int lane,
                                                      double *__restrict__ s_t) {
    // s_t: this wave's corner of LDS, 64 rows of KP + 2 doubles.  The weights are pair-major (a pair's KP atoms contiguous: what
    // the scalar loads of k_gd_grad_c need), so a lane that read and wrote ITS pair touched 16 bytes in each of 64 lines per
    // instruction; now a wave moves the column block's 64 x KP doubles as whole kilobytes and the rows change hands in LDS
    // (row stride KP * 8 + 16 bytes: the 16-byte accesses of both views fall on distinct banks).
    if (jb * 64 + 63 <= i) {  // (a column block on or below the diagonal's row: nothing to update, nothing to load)
        if (lane == 0) {
            double *o = P.wpart + ((int64_t)i * nbj + jb) * 3;
            o[0] = o[1] = o[2] = 0.0;
        }
        return;
    }
    const double done_flag = P.stat[5];
    constexpr int CPP = KP / 2, RS = KP + 2;  // 16-byte pieces per pair; row stride in doubles
    const int j = jb * 64 + lane;
    const bool live = j > i && j < P.n;
    double *wrow = P.WS + ((int64_t)i * P.n_pad + jb * 64) * KP;  // the block's 64 pairs (all inside the padded array)
    double wo[KP], g[KP], mi[KP], mj[KP];
    // Everything the update reads is requested before anything is used (round 4: the slabs came one round trip
    // after the other and the two dual terms of every atom one after the other: twelve dependent round trips a wave).
    double2 wv[CPP];
#pragma unroll
    for (int r = 0; r < CPP; ++r) wv[r] = *reinterpret_cast<const double2 *>(wrow + 2 * (r * 64 + lane));
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        mi[k] = P.M[(int64_t)k * P.n_pad + i];
        mj[k] = P.M[(int64_t)k * P.n_pad + j];
        g[k] = 0.0;
    }
    for (int ts0 = 0; ts0 < (GD_UPD_EXP == 4 ? 0 : P.TS); ts0 += 4) {  // four slabs in flight; added in the order of the slabs
        double2 gv[4][KP / 2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (ts0 + u < P.TS) {
                const double *gp = P.GS + ((((int64_t)(ts0 + u) * P.n_pad + i) * nbj + jb) * (KP / 2)) * 128 + 2 * lane;
#pragma unroll
                for (int k = 0; k < KP; k += 2) gv[u][k / 2] = *reinterpret_cast<const double2 *>(gp + (k / 2) * 128);
            } else {
#pragma unroll
                for (int k = 0; k < KP; k += 2) gv[u][k / 2] = double2{0.0, 0.0};  // (g is never -0.0: adding +0.0 changes nothing)
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < KP; k += 2) g[k] += gv[u][k / 2].x, g[k + 1] += gv[u][k / 2].y;
    }
    // piece c = r * 64 + lane of the block belongs to pair c / CPP, atoms 2 (c % CPP), 2 (c % CPP) + 1
#pragma unroll
    for (int r = 0; r < CPP; ++r) {
        const int c = r * 64 + lane;
        *reinterpret_cast<double2 *>(s_t + (c / CPP) * RS + 2 * (c % CPP)) = wv[r];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < CPP; ++q) {
        const double2 v = *reinterpret_cast<const double2 *>(s_t + lane * RS + 2 * q);
        wo[2 * q] = v.x, wo[2 * q + 1] = v.y;
    }
    double num = 0, den = 0, wmax = 0;
    double wsum = 0.0;
#pragma unroll
    for (int k = 0; k < KP; ++k) wsum += k < P.K ? wo[k] : 0.0;
    const double Tn = (double)P.T;
    double wn[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        double gk = g[k] + P.l1_w * Tn;                                       // :448
        if (P.ortho_w > 0) gk += P.ortho_w * Tn * (wsum - wo[k]);             // :450-451
        const double dterm = mi[k] + mj[k];
        double v = wo[k] - P.step_w / Tn * (dterm + gk);                      // :456
        v = v < 0 ? 0.0 : v;                                                  // :459
        wn[k] = k < P.K ? v : 0.0;
        if (live && k < P.K) {
            num += (v - wo[k]) * (v - wo[k]);
            den += wo[k] * wo[k];
            wmax = fmax(wmax, fabs(wo[k]));
        }
    }
    if (done_flag != 0.0) return;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < CPP; ++q) *reinterpret_cast<double2 *>(s_t + lane * RS + 2 * q) = double2{wn[2 * q], wn[2 * q + 1]};
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < CPP; ++r) {  // (pairs on or below the diagonal and past the last node keep their zeros: not written)
        const int c = r * 64 + lane, jp = jb * 64 + c / CPP;
        const double2 v = *reinterpret_cast<const double2 *>(s_t + (c / CPP) * RS + 2 * (c % CPP));
        if (GD_UPD_EXP != 5 && jp > i && jp < P.n) *reinterpret_cast<double2 *>(wrow + 2 * c) = v;
    }
    if (GD_UPD_EXP != 3 && GD_UPD_EXP != 5 && P.WSf) {  // the fp32 mirror the next step's contractions read (round 4: a pass of its own over all of WS before)
        constexpr int FPP = KP / 4 > 0 ? KP / 4 : 1, FS = KP + 4;  // 16-byte pieces per pair; row stride in floats
        float *s_f = reinterpret_cast<float *>(s_t);  // (64 x (KP + 4) floats fit the same corner)
        float *frow = P.WSf + ((int64_t)i * P.n_pad + jb * 64) * KP;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < KP; k += 4)
            *reinterpret_cast<float4 *>(s_f + lane * FS + k) = float4{(float)wn[k], (float)wn[k + 1], (float)wn[k + 2], (float)wn[k + 3]};
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < FPP; ++r) {
            const int c = r * 64 + lane, jp = jb * 64 + c / FPP;
            const float4 v = *reinterpret_cast<const float4 *>(s_f + (c / FPP) * FS + 4 * (c % FPP));
            if (jp > i && jp < P.n) *reinterpret_cast<float4 *>(frow + 4 * c) = v;
        }
    }
    num = gd_wave_sum(num);
    den = gd_wave_sum(den);
    wmax = gd_wave_max(wmax);
    if (lane == 0) {
        double *o = P.wpart + ((int64_t)i * nbj + jb) * 3;
        o[0] = num;
        o[1] = den;
        o[2] = wmax;
    }
}

template <int NT>
__device__ __forceinline__ void gd_weight_finish_body(const GdDev &P, const int64_t nparts) {
    const double done_flag = P.stat[5];
    __shared__ double s_a[NT], s_b[NT], s_c[NT];
    const int tid = threadIdx.x;
    double num = 0, den = 0, mx = 0;
    // (one workgroup folds n x n / 64 partials: eight entries' loads in flight, added in the order they always were)
    const double *__restrict__ wp = P.wpart;
    for (int64_t q0 = tid; q0 < nparts; q0 += 8 * NT) {
        double a[8], b[8], c[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t q = q0 + (int64_t)j * NT;
            const bool in = q < nparts;
            a[j] = in ? wp[3 * q] : 0.0;
            b[j] = in ? wp[3 * q + 1] : 0.0;
            c[j] = in ? wp[3 * q + 2] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (q0 + (int64_t)j * NT < nparts) {
                num += a[j];
                den += b[j];
                mx = fmax(mx, c[j]);
            }
    }
    s_a[tid] = num;
    s_b[tid] = den;
    s_c[tid] = mx;
    __syncthreads();
    for (int o = NT / 2; o > 0; o >>= 1) {
        if (tid < o) {
            s_a[tid] += s_a[tid + o];
            s_b[tid] += s_b[tid + o];
            s_c[tid] = fmax(s_c[tid], s_c[tid + o]);
        }
        __syncthreads();
    }
    if (tid == 0 && done_flag == 0.0) P.stat[1] = s_c[0] <= 1e-8 ? sqrt(s_a[0]) : sqrt(s_a[0]) / sqrt(s_b[0]);
}

// degrees S W_k' of the weights in WS: one wave per node
template <int KP>
__device__ __forceinline__ void gd_degree_body(const GdDev &P, double *__restrict__ out, const int chk_done, const int v, const int lane) {
    const double done_flag = chk_done ? P.stat[5] : 0.0;
    if (v >= P.n_pad) return;  // a wave past the last node (four nodes per block in k_gd_finish3)
    // (KP is the compile-time stride of an entry's atoms: with the runtime K the sums were indexed in scratch)
    double acc[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) acc[k] = 0.0;
    // (a gather of n 64-byte pieces, 64 KB apart below the diagonal: more of them in flight -- four neighbour groups per round
    //  trip -- measured slower, 18.2 against 17.3 us at n = 1024: the pieces themselves are the cost, not the round trips)
    if (v < P.n)
        for (int u = lane; u < P.n; u += 64) {
            if (u == v) continue;
            const double *p = u < v ? P.WS + ((int64_t)u * P.n_pad + v) * KP : P.WS + ((int64_t)v * P.n_pad + u) * KP;
#pragma unroll
            for (int k = 0; k < KP; k += 2) {  // (16-byte accesses: an entry's KP atoms are contiguous and 16-byte aligned)
                const double2 w = *reinterpret_cast<const double2 *>(p + k);
                acc[k] += w.x, acc[k + 1] += w.y;
            }
        }
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const double s = gd_wave_sum(acc[k]);
        if (lane == 0 && k < P.K && done_flag == 0.0) out[(int64_t)k * P.n_pad + v] = s;
    }
}


// ---- launch wrappers.  The pieces of a step that do not depend on each other share a launch (the step is
// launch-bound below ~1e8 pair-sample products: 13 dependent kernels cost 64 us whatever the size):
// the two partial-sum folds; the coefficient and weight updates; the two finishers and the new degrees.
__global__ __launch_bounds__(256) void k_gd_sc_reduce(GdDev P) { gd_sc_reduce_body(P, (int)blockIdx.x); }
__global__ __launch_bounds__(256) void k_gd_reduce2(GdDev P, int nsc) {  // blocks [0, nsc): SC, then M
    if ((int)blockIdx.x < nsc)
        gd_sc_reduce_body(P, (int)blockIdx.x);
    else
        gd_m_reduce_body(P, (int)blockIdx.x - nsc);
}
template <int KP>
__global__ __launch_bounds__(64) void k_gd_coef_update(GdDev P) {
    gd_coef_update_body<KP, 1>(P, (int)blockIdx.x, (int)threadIdx.x);
}
// four waves per workgroup, NW windows or a (row, column block) each: the first blocks take the windows, the others
// (row i, column blocks 4 q .. 4 q + 3)
template <int KP>
__global__ __launch_bounds__(256) void k_gd_update2(GdDev P, int nbj, int cblocks) {
    __shared__ double s_t[4][64 * (KP + 2)];  // the weight update's row exchange, a corner per wave
    const int b = (int)blockIdx.x, wv = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
#if GD_UPD_EXP == 1  // (tools/build_variant_gd.sh: what the parts of this kernel cost; results are wrong in these builds)
    if (b < cblocks) return;
#elif GD_UPD_EXP == 2
    if (b >= cblocks) return;
#endif
    if (b < cblocks) {
        constexpr int NW = KP >= 16 ? 2 : 4;  // windows per wave
        const int tw0 = (b * 4 + wv) * NW;
        if (tw0 < P.Tw) gd_coef_update_body<KP, NW>(P, tw0, lane);
    } else {
        const int nq = (nbj + 3) / 4, r = b - cblocks;
        const int jb = (r % nq) * 4 + wv, i = r / nq;
        if (jb < nbj) gd_weight_update_body<KP>(P, jb, i, nbj, lane, s_t[wv]);
    }
}
__global__ __launch_bounds__(256) void k_gd_coef_finish(GdDev P) { gd_coef_finish_body(P); }
template <int KP>
__global__ __launch_bounds__(64) void k_gd_degree(GdDev P, double *__restrict__ out, int chk_done) {
    gd_degree_body<KP>(P, out, chk_done, (int)blockIdx.x, (int)threadIdx.x);
}
// block 0: all-zero guard of the coefficients; block 1: relative change of the weights; then four nodes'
// degrees per block
template <int KP>
__global__ __launch_bounds__(256) void k_gd_finish3(GdDev P, int64_t nparts, double *__restrict__ deg_out) {
    const int b = (int)blockIdx.x;
    if (b == 0)
        gd_coef_finish_body(P);
    else if (b == 1)
        gd_weight_finish_body<256>(P, nparts);
    else
        gd_degree_body<KP>(P, deg_out, 1, (b - 2) * 4 + ((int)threadIdx.x >> 6), (int)threadIdx.x & 63);
}

// ------------------------------------------------------------------------------------------
// dual update (log_model.py:39-50) with the overshoot 2 new - old of weights and coefficients
// (base_model.py:509-514).  fixed_weights: predict() passes the weights themselves (:665-669).
template <int KP>
__global__ __launch_bounds__(256) void k_gd_dual_update(GdDev P, int fixed_weights) {
    const double done_flag = P.stat[5];
    // the overshoot coefficients 2 new - old of this workgroup's 8 windows, once per workgroup (round 4: every thread used
    // to fetch the 2 x K of every window itself, window after window, each window's load of the dual variable behind the
    // previous window's store: 46 us for 67 MB of traffic)
    __shared__ double s_cf[8][KP];
    const int tw0 = blockIdx.y * 8;
    if (threadIdx.x < 8 * KP) {
        const int w = threadIdx.x / KP, k = threadIdx.x % KP, tw = tw0 + w;
        s_cf[w][k] = (k < P.K && tw < P.Tw) ? 2.0 * P.Cw[(int64_t)k * P.Tw + tw] - P.C[(int64_t)k * P.T + (int64_t)tw * P.W] : 0.0;
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n) return;
    double dv[KP];  // overshoot degrees of node i, shared by the 8 windows of this thread
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const double dn = P.deg_new[(int64_t)k * P.n_pad + i], dold = P.deg[(int64_t)k * P.n_pad + i];
        dv[k] = fixed_weights ? dold : 2.0 * dn - dold;
    }
    const double sigma = P.step_dual;
    double *__restrict__ dual = P.dual;
    double dl[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) dl[w] = tw0 + w < P.Tw ? dual[(int64_t)(tw0 + w) * P.n + i] : 0.0;  // all eight in flight
    if (done_flag != 0.0) return;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        if (tw0 + w >= P.Tw) break;
        double step = 0.0;
#pragma unroll
        for (int k = 0; k < KP; ++k)
            if (k < P.K) step += dv[k] * s_cf[w][k];
        const double d = dl[w] + sigma * step;
        dual[(int64_t)(tw0 + w) * P.n + i] = (d - sqrt(d * d + 4.0 * sigma)) / 2.0;
    }
}

// coefficients_ <- repeat(new window coefficients) (base_model.py:495-501,519) and the partial sums
// of relative_error(coefficients_.ravel(), new.ravel()) (:516); grid of <= 256 workgroups
__global__ __launch_bounds__(256) void k_gd_coef_commit(GdDev P) {
    if (P.stat[5] != 0.0) return;
    __shared__ double s_a[256], s_b[256], s_c[256];
    const int tid = threadIdx.x;
    double num = 0.0, den = 0.0, omax = 0.0;
    for (int64_t q = (int64_t)blockIdx.x * 256 + tid; q < (int64_t)P.K * P.T; q += (int64_t)gridDim.x * 256) {
        const int64_t k = q / P.T, t = q % P.T;
        const double nw = P.Cw[k * P.Tw + t / P.W], o = P.C[q];
        num += (nw - o) * (nw - o);
        den += o * o;
        omax = fmax(omax, fabs(o));
        P.C[q] = nw;
        P.CT[t * P.KP + k] = nw;
        if (P.CTf) P.CTf[t * P.KP + k] = (float)nw;
    }
    s_a[tid] = num;
    s_b[tid] = den;
    s_c[tid] = omax;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            s_a[tid] += s_a[tid + o];
            s_b[tid] += s_b[tid + o];
            s_c[tid] = fmax(s_c[tid], s_c[tid + o]);
        }
        __syncthreads();
    }
    if (tid == 0) {
        P.cnorm[3 * blockIdx.x] = s_a[0];
        P.cnorm[3 * blockIdx.x + 1] = s_b[0];
        P.cnorm[3 * blockIdx.x + 2] = s_c[0];
    }
}
// stat[0] = relative change of the coefficients (plain norm if the old ones were ~ 0, evaluation.py:21-23),
// stat[3] = ||new - old|| (predict's stopping norm, base_model.py:671).  fit_mode: the bookkeeping of
// _single_fit (base_model.py:552-581) -- history row, converged_, and the "done" word that turns every
// kernel of the steps already enqueued behind this one into a no-op (the host looks once per chunk).
__global__ __launch_bounds__(64) void k_gd_coef_stat(GdDev P, int nblocks, int fit_mode) {
    if (P.stat[5] != 0.0) return;
    const int lane = threadIdx.x;
    double num = 0, den = 0, omax = 0;
    for (int b = lane; b < nblocks; b += 64) {
        num += P.cnorm[3 * b];
        den += P.cnorm[3 * b + 1];
        omax = fmax(omax, P.cnorm[3 * b + 2]);
    }
    num = gd_wave_sum(num);
    den = gd_wave_sum(den);
    omax = gd_wave_max(omax);
    if (lane == 0) {
        const double dc = omax <= 1e-8 ? sqrt(num) : sqrt(num) / sqrt(den);
        P.stat[0] = dc;
        P.stat[3] = sqrt(num);
        if (fit_mode) {
            const double dw = P.stat[1], tol = P.stat[6];
            const long long i = (long long)P.stat[4];
            if (i < P.hist_cap) {
                P.hist[2 * i] = dc;
                P.hist[2 * i + 1] = dw;
            }
            if (dc < tol && dw < tol) P.stat[7] = (double)i;
            P.stat[4] = (double)(i + 1);
            if (P.stat[7] > 0.0) P.stat[5] = 1.0;  // sic: a stop at step 0 is not taken (:578)
        }
    }
}

// fp32 copies of the gradient kernels' operands (gl_graphdict_set_precision(s, 32))
__global__ void k_gd_to_f32(const double *__restrict__ src, float *__restrict__ dst, int64_t count, const double *__restrict__ done) {
    if (done && done[0] != 0.0) return;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < count; q += (int64_t)gridDim.x * blockDim.x) dst[q] = (float)src[q];
}

// ---- layout kernels --------------------------------------------------------------------------
__global__ void k_gd_pack_x(const double *__restrict__ x, GdDev P) {  // x: T x n row-major
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (int64_t)P.T * P.n) return;
    const int64_t t = q / P.n, i = q % P.n;
    const double v = x[q];
    P.X[t * P.n_pad + i] = v;
    P.XT[i * P.Tp + t] = v;
}
__global__ void k_gd_pack_w(const double *__restrict__ w, GdDev P) {  // w: K x E condensed
    const int64_t E = (int64_t)P.n * (P.n - 1) / 2;
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (int64_t)P.n * P.n) return;
    const int64_t i = q / P.n, j = q % P.n;
    if (j <= i) return;
    const int64_t e = gl_condensed_index(i, j, P.n);
    for (int k = 0; k < P.K; ++k) P.WS[(i * P.n_pad + j) * P.KP + k] = w[(int64_t)k * E + e];
}
__global__ void k_gd_unpack_w(double *__restrict__ w, GdDev P) {
    const int64_t E = (int64_t)P.n * (P.n - 1) / 2;
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (int64_t)P.n * P.n) return;
    const int64_t i = q / P.n, j = q % P.n;
    if (j <= i) return;
    const int64_t e = gl_condensed_index(i, j, P.n);
    for (int k = 0; k < P.K; ++k) w[(int64_t)k * E + e] = P.WS[(i * P.n_pad + j) * P.KP + k];
}
__global__ void k_gd_pack_c(GdDev P) {  // CT from C
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (int64_t)P.K * P.T) return;
    const int64_t k = q / P.T, t = q % P.T;
    P.CT[t * P.KP + k] = P.C[q];
}

}  // namespace


namespace {

void gd_drop_graphs(gl_graphdict *s) {
    for (hipGraphExec_t &g : s->step_graph) {
        if (g) (void)hipGraphExecDestroy(g);
        g = nullptr;
    }
}

int gd_check(gl_graphdict *s) {
    if (!s || !s->ctx) return GL_E_INVALID;
    return GL_OK;
}

template <typename T>
int gd_alloc(gl_graphdict *s, T **p, size_t count) {
    gl_ctx *ctx = s->ctx;
    GL_MALLOC(ctx, (void **)p, std::max<size_t>(count, 1) * sizeof(T), "graphdict");
    s->allocs.push_back(*p);
    GL_HIP(ctx, hipMemsetAsync(*p, 0, std::max<size_t>(count, 1) * sizeof(T), ctx->stream));
    return GL_OK;
}

// fp32 mirrors of the state from scratch (a new state, a switch of precision); the steps keep them from there
void gd_refresh_f32(gl_graphdict *s) {
    const GdDev &P = s->dev;
    if (!s->f32 || !s->have_state) return;
    hipStream_t st = s->ctx->stream;
    const int64_t cw = (int64_t)P.n_pad * P.n_pad * P.KP, cc = (int64_t)P.Tp * P.KP;
    k_gd_to_f32<<<dim3((unsigned)std::min<int64_t>(4096, (cw + 255) / 256)), 256, 0, st>>>(P.WS, s->WSf, cw, nullptr);
    k_gd_to_f32<<<dim3((unsigned)std::min<int64_t>(1024, (cc + 255) / 256)), 256, 0, st>>>(P.CT, s->CTf, cc, nullptr);
}

int gd_launch_grad_c(gl_graphdict *s) {
    const GdDev &P = s->dev;
    dim3 grid((unsigned)(P.Tp / GD_SB), (unsigned)P.NG);
    hipStream_t st = s->ctx->stream;
    const int xd_env = getenv("GRAPHLEARN_AMD_GD_XD") ? atoi(getenv("GRAPHLEARN_AMD_GD_XD")) : -1;  // (read per launch: a test switches it)
    if (s->f32) {  // (WSf is kept by the weight update)
        const int xd = xd_env >= 0 ? xd_env : GD_XD32_DEFAULT;
        switch (P.KP) {
            case 4: k_gd_grad_c<float, 4, 2><<<grid, 64, 0, st>>>(s->XTf, s->WSf, P.SC, P.n, P.n_pad, P.Tp, P.NG, P.stat + 5); break;
            case 8:
                if (s->rc32 == 4) k_gd_grad_c<float, 8, 4><<<grid, 64, 0, st>>>(s->XTf, s->WSf, P.SC, P.n, P.n_pad, P.Tp, P.NG, P.stat + 5);
                else if (s->rc32 == 3) k_gd_grad_c<float, 8, 3><<<grid, 64, 0, st>>>(s->XTf, s->WSf, P.SC, P.n, P.n_pad, P.Tp, P.NG, P.stat + 5);
                else if (xd == 4) k_gd_grad_c<float, 8, 2, 4><<<grid, 64, 0, st>>>(s->XTf, s->WSf, P.SC, P.n, P.n_pad, P.Tp, P.NG, P.stat + 5);
                else k_gd_grad_c<float, 8, 2><<<grid, 64, 0, st>>>(s->XTf, s->WSf, P.SC, P.n, P.n_pad, P.Tp, P.NG, P.stat + 5);
                break;
            default: k_gd_grad_c<float, 16, 1><<<grid, 64, 0, st>>>(s->XTf, s->WSf, P.SC, P.n, P.n_pad, P.Tp, P.NG, P.stat + 5); break;
        }
        GL_HIP(s->ctx, hipGetLastError());
        return GL_OK;
    }
    switch (P.KP) {
        case 4: k_gd_grad_c<double, 4, 2><<<grid, 64, 0, st>>>(P.XT, P.WS, P.SC, P.n, P.n_pad, P.Tp, P.NG, P.stat + 5); break;
        case 8:
            if (xd_env == 4) k_gd_grad_c<double, 8, 2, 4><<<grid, 64, 0, st>>>(P.XT, P.WS, P.SC, P.n, P.n_pad, P.Tp, P.NG, P.stat + 5);
            else k_gd_grad_c<double, 8, 2><<<grid, 64, 0, st>>>(P.XT, P.WS, P.SC, P.n, P.n_pad, P.Tp, P.NG, P.stat + 5);
            break;
        default: k_gd_grad_c<double, 16, 1><<<grid, 64, 0, st>>>(P.XT, P.WS, P.SC, P.n, P.n_pad, P.Tp, P.NG, P.stat + 5); break;
    }
    GL_HIP(s->ctx, hipGetLastError());
    return GL_OK;
}

int gd_strip_w(int KP) { return KP <= 8 ? 8 : 4; }
void gd_launch_dual_update(const GdDev &P, hipStream_t st, int fixed) {
    const dim3 grid((unsigned)((P.n + 255) / 256), (unsigned)((P.Tw + 7) / 8));
    switch (P.KP) {
        case 4: k_gd_dual_update<4><<<grid, 256, 0, st>>>(P, fixed); break;
        case 8: k_gd_dual_update<8><<<grid, 256, 0, st>>>(P, fixed); break;
        default: k_gd_dual_update<16><<<grid, 256, 0, st>>>(P, fixed); break;
    }
}
void gd_launch_coef_update(const GdDev &P, hipStream_t st) {
    switch (P.KP) {
        case 4: k_gd_coef_update<4><<<dim3((unsigned)P.Tw), 64, 0, st>>>(P); break;
        case 8: k_gd_coef_update<8><<<dim3((unsigned)P.Tw), 64, 0, st>>>(P); break;
        default: k_gd_coef_update<16><<<dim3((unsigned)P.Tw), 64, 0, st>>>(P); break;
    }
}
int gd_commit_blocks(const GdDev &P) { return (int)std::max<int64_t>(1, std::min<int64_t>(256, ((int64_t)P.K * P.T + 1023) / 1024)); }

int gd_launch_grad_w(gl_graphdict *s) {
    const GdDev &P = s->dev;
    const int tchunk = (P.Tp + P.TS - 1) / P.TS;
    hipStream_t st = s->ctx->stream;
    dim3 grid((unsigned)(P.nwtasks * P.TS));
    if (s->f32) {  // (CTf is kept by k_gd_coef_commit)
        switch (P.KP) {
            case 4: k_gd_grad_w<float, 4, 8><<<grid, 256, 0, st>>>(s->Xf, s->CTf, P.GS, P.wtasks, P.nwtasks, P.n_pad, P.Tp, tchunk, s->dbgbuf, P.stat + 5); break;
            case 8: k_gd_grad_w<float, 8, 8><<<grid, 256, 0, st>>>(s->Xf, s->CTf, P.GS, P.wtasks, P.nwtasks, P.n_pad, P.Tp, tchunk, s->dbgbuf, P.stat + 5); break;
            default: k_gd_grad_w<float, 16, 4><<<grid, 256, 0, st>>>(s->Xf, s->CTf, P.GS, P.wtasks, P.nwtasks, P.n_pad, P.Tp, tchunk, s->dbgbuf, P.stat + 5); break;
        }
        GL_HIP(s->ctx, hipGetLastError());
        return GL_OK;
    }
    switch (P.KP) {
        case 4: k_gd_grad_w<double, 4, 8><<<grid, 256, 0, st>>>(P.X, P.CT, P.GS, P.wtasks, P.nwtasks, P.n_pad, P.Tp, tchunk, s->dbgbuf, P.stat + 5); break;
        case 8: k_gd_grad_w<double, 8, 8><<<grid, 256, 0, st>>>(P.X, P.CT, P.GS, P.wtasks, P.nwtasks, P.n_pad, P.Tp, tchunk, s->dbgbuf, P.stat + 5); break;
        default: k_gd_grad_w<double, 16, 4><<<grid, 256, 0, st>>>(P.X, P.CT, P.GS, P.wtasks, P.nwtasks, P.n_pad, P.Tp, tchunk, s->dbgbuf, P.stat + 5); break;
    }
    GL_HIP(s->ctx, hipGetLastError());
    return GL_OK;
}

int gd_degrees(gl_graphdict *s, double *out, int chk_done = 0) {
    switch (s->dev.KP) {
        case 4: k_gd_degree<4><<<dim3((unsigned)s->dev.n), 64, 0, s->ctx->stream>>>(s->dev, out, chk_done); break;
        case 8: k_gd_degree<8><<<dim3((unsigned)s->dev.n), 64, 0, s->ctx->stream>>>(s->dev, out, chk_done); break;
        default: k_gd_degree<16><<<dim3((unsigned)s->dev.n), 64, 0, s->ctx->stream>>>(s->dev, out, chk_done); break;
    }
    GL_HIP(s->ctx, hipGetLastError());
    return GL_OK;
}

// one primal-dual step (base_model.py:485-527); the relative changes land in stat[0..1]
int gd_side_branch(gl_graphdict *s) {  // (outside any capture)
    // (measured at K = 8, n = 1024, T = 4096: 1.006 ms per step with the branch, 1.007 without -- the folds stretch to the
    //  length of the kernel beside them and that one slows by as much: opt-in)
    if (s->side || !getenv("GRAPHLEARN_AMD_GD_FORK") || atoi(getenv("GRAPHLEARN_AMD_GD_FORK")) == 0) return GL_OK;
    gl_ctx *ctx = s->ctx;
    GL_HIP(ctx, hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
    GL_HIP(ctx, hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
    GL_HIP(ctx, hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
    return GL_OK;
}

int gd_step(gl_graphdict *s, int fit_mode = 0) {
    gl_ctx *ctx = s->ctx;
    GdDev &P = s->dev;
    hipStream_t st = ctx->stream;
    if (P.log_c > 0) k_gd_logc_reset<<<dim3((unsigned)((P.Tw + 255) / 256)), 256, 0, st>>>(P);
    GL_TRY(gd_launch_grad_c(s));
    // C @ dual and the two folds need the coefficient gradient's partials, the coefficients and the dual variable -- not the
    // weight gradient: they run beside it on a branch of their own (bandwidth- and latency-bound beside a kernel that
    // is neither).  Inside a capture the events become the graph's edges.
    hipStream_t sb = st;
    if (s->side) {
        GL_HIP(ctx, hipEventRecord(s->ev_fork, st));
        GL_HIP(ctx, hipStreamWaitEvent(s->side, s->ev_fork, 0));
        sb = s->side;
    }
    const int mchunk = (P.T + P.MS - 1) / P.MS;
    {
        const dim3 gm((unsigned)(P.n_pad / 64), (unsigned)((P.MS + 3) / 4));
        switch (P.KP) {
            case 4: k_gd_m<4><<<gm, 256, 0, sb>>>(P, mchunk); break;
            case 8: k_gd_m<8><<<gm, 256, 0, sb>>>(P, mchunk); break;
            default: k_gd_m<16><<<gm, 256, 0, sb>>>(P, mchunk); break;
        }
    }
    const int nsc = (int)(((int64_t)P.KP * P.Tp + 63) / 64), nmr = (int)(((int64_t)P.KP * P.n_pad + 63) / 64);
    k_gd_reduce2<<<dim3((unsigned)(nsc + nmr)), dim3(64, 4), 0, sb>>>(P, nsc);
    if (s->side) GL_HIP(ctx, hipEventRecord(s->ev_join, s->side));
    GL_TRY(gd_launch_grad_w(s));
    if (s->side) GL_HIP(ctx, hipStreamWaitEvent(st, s->ev_join, 0));
    const int nbj = P.n_pad / 64;
    const int wpb = 4 * (P.KP >= 16 ? 2 : 4);  // windows per workgroup: four waves, NW each (k_gd_update2)
    const int cblocks = (P.Tw + wpb - 1) / wpb;
    const dim3 gu((unsigned)(cblocks + (nbj + 3) / 4 * P.n));
    switch (P.KP) {
        case 4: k_gd_update2<4><<<gu, 256, 0, st>>>(P, nbj, cblocks); break;
        case 8: k_gd_update2<8><<<gu, 256, 0, st>>>(P, nbj, cblocks); break;
        default: k_gd_update2<16><<<gu, 256, 0, st>>>(P, nbj, cblocks); break;
    }
    {
        const dim3 gf((unsigned)(2 + (P.n + 3) / 4));
        switch (P.KP) {
            case 4: k_gd_finish3<4><<<gf, 256, 0, st>>>(P, (int64_t)P.n * nbj, P.deg_new); break;
            case 8: k_gd_finish3<8><<<gf, 256, 0, st>>>(P, (int64_t)P.n * nbj, P.deg_new); break;
            default: k_gd_finish3<16><<<gf, 256, 0, st>>>(P, (int64_t)P.n * nbj, P.deg_new); break;
        }
    }
    gd_launch_dual_update(P, st, 0);
    const int cb = gd_commit_blocks(P);
    k_gd_coef_commit<<<dim3((unsigned)cb), 256, 0, st>>>(P);
    k_gd_coef_stat<<<1, 64, 0, st>>>(P, cb, fit_mode);
    GL_HIP(ctx, hipGetLastError());
    std::swap(P.deg, P.deg_new);
    s->step_parity ^= 1;
    return GL_OK;
}

// one step through the captured graph of the current parity (captured on first use; the kernels'
// arguments -- pointers, regularisers, step sizes -- are frozen in it, so set_params, the plateau rule and
// a new history buffer drop the graphs).  fit_mode steps keep the books of _single_fit on the device.
int gd_step_replayed(gl_graphdict *s, int fit_mode) {
    gl_ctx *ctx = s->ctx;
    GdDev &P = s->dev;
    GL_TRY(gd_side_branch(s));
    if (getenv("GRAPHLEARN_AMD_NO_GRAPH")) return gd_step(s, fit_mode);
    const int par = s->step_parity;
    hipGraphExec_t &exec = s->step_graph[2 * fit_mode + par];
    if (!exec) {
        hipGraph_t graph = nullptr;
        GL_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
        const int rc = gd_step(s, fit_mode);  // flips the parity and the degree buffers: undone below, replay redoes it
        hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
        std::swap(P.deg, P.deg_new);
        s->step_parity ^= 1;
        if (rc != GL_OK) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc;
        }
        if (e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        if (e != hipSuccess) return gl_set_error(ctx, GL_E_HIP, "step graph capture failed: %s", hipGetErrorString(e));
    }
    GL_HIP(ctx, hipGraphLaunch(exec, ctx->stream));
    std::swap(P.deg, P.deg_new);
    s->step_parity ^= 1;
    return GL_OK;
}

}  // namespace

// ==============================================================================================
GL_API int gl_graphdict_create(gl_ctx *ctx, int64_t n_nodes, int64_t n_samples, int n_atoms, int window_size,
                               gl_graphdict **out) {
    if (!ctx) return GL_E_INVALID;
    if (!out) return gl_set_error(ctx, GL_E_INVALID, "out is NULL");
    *out = nullptr;
    if (n_nodes < 2 || n_nodes > 16384) return gl_set_error(ctx, GL_E_INVALID, "n_nodes must be in [2, 16384]");
    if (n_atoms < 1 || n_atoms > GD_MAXK)
        return gl_set_error(ctx, GL_E_INVALID, "n_atoms must be in [1, %d], got %d", GD_MAXK, n_atoms);
    if (window_size < 1 || n_samples < 1 || n_samples % window_size != 0)
        return gl_set_error(ctx, GL_E_INVALID, "n_samples must be a positive multiple of window_size");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    gl_graphdict *s = new (std::nothrow) gl_graphdict();
    if (!s) return GL_E_NOMEM;
    s->ctx = ctx;
    GdDev &P = s->dev;
    P.n = (int)n_nodes;
    P.n_pad = (int)((n_nodes + 63) / 64 * 64);
    P.T = (int)n_samples;
    P.Tp = (int)((n_samples + GD_SB - 1) / GD_SB * GD_SB);
    P.W = window_size;
    P.Tw = P.T / P.W;
    P.K = n_atoms;
    P.KP = n_atoms <= 4 ? 4 : (n_atoms <= 8 ? 8 : 16);
    // enough waves to fill 256 CUs x 4 SIMDs a few times over, slabs kept small
    // Rows per strip of k_gd_grad_c: 2 at K <= 8 (two operand sets of rows x K weights must fit the scalar file; in fp64 a weight
    // is two scalar registers).  GRAPHLEARN_AMD_GD_RC32=3 / 4: three / four rows in the fp32 form at K = 8 -- fewer loads of x_j
    // per fused multiply-add.  Measured on one GPU, 100-step fits: three rows are faster back to back (0.430 against 0.440 ms)
    // and slower inside a step (450 against 435 us under rocprofv3), their fewer strip groups fold faster (16 against 22 us),
    // and the step comes out the same (0.959 ms both); four rows spill 53 scalar registers (0.563 ms).  Default: 2.
    s->rc32 = 2;
    if (const char *e = getenv("GRAPHLEARN_AMD_GD_RC32"))
        if (P.KP == 8 && (atoi(e) == 2 || atoi(e) == 3 || atoi(e) == 4)) s->rc32 = atoi(e);
    const int nstrips = (P.n + gd_rows_c(P.KP) - 1) / gd_rows_c(P.KP);
    const int sblocks = P.Tp / GD_SB;
    P.NG = std::max(1, std::min(nstrips, std::min(512, (8192 + sblocks - 1) / sblocks)));
    // (the strip groups of either precision: one strip per group where there are enough of them; the partials are sized for
    //  the larger count)
    s->ng64 = P.NG;
    s->ng32 = (P.KP == 8 && s->rc32 != 2) ? std::max(1, std::min((P.n + s->rc32 - 1) / s->rc32, P.NG)) : P.NG;
    std::vector<int2> tasks;
    {
        const int sw = gd_strip_w(P.KP);
        for (int st = 0; st < P.n_pad / sw; ++st)
            for (int jb = 0; jb < P.n_pad / 64; ++jb)
                if (jb * 64 + 63 > st * sw && st * sw < P.n) tasks.push_back(int2{st, jb});
    }
    P.nwtasks = (int)tasks.size();
    const int64_t wtasks = P.nwtasks;
    P.TS = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, (P.Tp + 511) / 512), (3072 + wtasks - 1) / wtasks));
    if (const char *e = getenv("GRAPHLEARN_AMD_GD_TS")) P.TS = std::max(1, std::min(64, atoi(e)));  // tuning aid
    if (const char *e = getenv("GRAPHLEARN_AMD_GD_NG")) P.NG = s->ng64 = s->ng32 = std::max(1, std::min(nstrips, atoi(e)));
    P.MS = std::max(1, std::min(128, P.T / 32));
    const size_t np2 = (size_t)P.n_pad * P.n_pad;
    // every buffer of the session in ONE zeroed allocation (21 hipMalloc + memset pairs cost more than a
    // hundred steps of the small engine)
    s->stage_elems = std::max((size_t)P.K * ((size_t)P.n * (P.n - 1) / 2), (size_t)P.T * P.n);
    struct Part {
        void **p;
        size_t bytes;
    };
    const Part parts[] = {
        {(void **)&P.X, (size_t)P.Tp * P.n_pad * 8}, {(void **)&P.XT, (size_t)P.n_pad * P.Tp * 8}, {(void **)&P.WS, np2 * P.KP * 8},
        {(void **)&P.C, (size_t)P.K * P.T * 8}, {(void **)&P.CT, (size_t)P.Tp * P.KP * 8}, {(void **)&P.Cw, (size_t)P.KP * P.Tw * 8},
        {(void **)&P.dual, (size_t)P.Tw * P.n * 8}, {(void **)&P.deg, (size_t)P.KP * P.n_pad * 8},
        {(void **)&P.deg_new, (size_t)P.KP * P.n_pad * 8}, {(void **)&P.SC, (size_t)P.NG * P.KP * P.Tp * 8},
        {(void **)&P.SCR, (size_t)P.KP * P.Tp * 8}, {(void **)&P.wtasks, tasks.size() * sizeof(int2)}, {(void **)&P.cnorm, 256 * 3 * 8},
        {(void **)&P.GS, (size_t)P.TS * np2 * P.KP * 8}, {(void **)&P.MSL, (size_t)P.MS * P.KP * P.n_pad * 8},
        {(void **)&P.M, (size_t)P.KP * P.n_pad * 8}, {(void **)&P.cpart, (size_t)P.Tw * 8},
        {(void **)&P.wpart, (size_t)P.n * (P.n_pad / 64) * 3 * 8}, {(void **)&P.stat, GD_NORM_SLOTS * 8},
        {(void **)&s->stage, s->stage_elems * 8}};
    size_t total = 0;
    for (const Part &q : parts) total += (std::max<size_t>(q.bytes, 8) + 255) / 256 * 256;
    void *arena = nullptr;
    if (gl_ctx_malloc(ctx, &arena, total, "graphdict") != hipSuccess) {
        delete s;
        return gl_set_error(ctx, GL_E_NOMEM, "hipMalloc(%zu) failed", total);
    }
    s->allocs.push_back(arena);
    {
        char *at = (char *)arena;
        for (const Part &q : parts) {
            *q.p = at;
            at += (std::max<size_t>(q.bytes, 8) + 255) / 256 * 256;
        }
    }
    GL_HIP(ctx, hipMemsetAsync(arena, 0, total, ctx->stream));
    GL_TRY(gl_h2d(ctx, P.wtasks, tasks.data(), tasks.size() * sizeof(int2)));
    if (getenv("GRAPHLEARN_AMD_GD_PROBE"))
        (void)gd_alloc(s, &s->dbgbuf, (size_t)16 * P.nwtasks * P.TS);
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = s;
    return GL_OK;
}

GL_API void gl_graphdict_destroy(gl_graphdict *s) {
    if (!s) return;
    if (s->ctx) {
        (void)hipSetDevice(s->ctx->device);
        (void)hipStreamSynchronize(s->ctx->stream);
    }
    gd_drop_graphs(s);
    if (s->side) {
        (void)hipStreamSynchronize(s->side);
        (void)hipEventDestroy(s->ev_fork);
        (void)hipEventDestroy(s->ev_join);
        (void)hipStreamDestroy(s->side);
    }
    for (void *p : s->allocs) (void)hipFree(p);
    delete s;
}

GL_API int gl_graphdict_set_signals(gl_graphdict *s, const double *x) {
    GL_TRY(gd_check(s));
    gl_ctx *ctx = s->ctx;
    if (!x) return gl_set_error(ctx, GL_E_INVALID, "x is NULL");
    GdDev &P = s->dev;
    GL_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cnt = (size_t)P.T * P.n;
    for (size_t q = 0; q < cnt; ++q)
        if (!std::isfinite(x[q])) return gl_set_error(ctx, GL_E_NONFINITE, "x contains NaN or Inf");
    GL_TRY(gl_h2d(ctx, s->stage, x, cnt * 8));
    k_gd_pack_x<<<dim3((unsigned)((cnt + 255) / 256)), 256, 0, ctx->stream>>>(s->stage, P);
    if (s->f32) {
        const int64_t pc = (int64_t)P.Tp * P.n_pad;
        k_gd_to_f32<<<dim3(2048), 256, 0, ctx->stream>>>(P.X, s->Xf, pc, nullptr);
        k_gd_to_f32<<<dim3(2048), 256, 0, ctx->stream>>>(P.XT, s->XTf, pc, nullptr);
    }
    GL_HIP(ctx, hipGetLastError());
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    s->have_x = true;
    return GL_OK;
}

GL_API int gl_graphdict_set_state(gl_graphdict *s, const double *coefficients, const double *weights, const double *dual) {
    GL_TRY(gd_check(s));
    gl_ctx *ctx = s->ctx;
    GdDev &P = s->dev;
    if (!coefficients || !weights) return gl_set_error(ctx, GL_E_INVALID, "coefficients / weights are NULL");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    const size_t E = (size_t)P.n * (P.n - 1) / 2;
    GL_TRY(gl_h2d(ctx, P.C, coefficients, (size_t)P.K * P.T * 8));
    GL_HIP(ctx, hipMemsetAsync(P.CT, 0, (size_t)P.Tp * P.KP * 8, ctx->stream));
    k_gd_pack_c<<<dim3((unsigned)(((size_t)P.K * P.T + 255) / 256)), 256, 0, ctx->stream>>>(P);
    GL_TRY(gl_h2d(ctx, s->stage, weights, (size_t)P.K * E * 8));
    GL_HIP(ctx, hipMemsetAsync(P.WS, 0, (size_t)P.n_pad * P.n_pad * P.KP * 8, ctx->stream));
    k_gd_pack_w<<<dim3((unsigned)(((size_t)P.n * P.n + 255) / 256)), 256, 0, ctx->stream>>>(s->stage, P);
    if (dual)
        GL_TRY(gl_h2d(ctx, P.dual, dual, (size_t)P.Tw * P.n * 8));
    else
        GL_HIP(ctx, hipMemsetAsync(P.dual, 0, (size_t)P.Tw * P.n * 8, ctx->stream));
    GL_TRY(gd_degrees(s, P.deg));
    s->have_state = true;
    gd_refresh_f32(s);
    GL_HIP(ctx, hipGetLastError());
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GL_OK;
}

GL_API int gl_graphdict_get_state(gl_graphdict *s, double *coefficients, double *weights, double *dual) {
    GL_TRY(gd_check(s));
    gl_ctx *ctx = s->ctx;
    GdDev &P = s->dev;
    if (!s->have_state) return gl_set_error(ctx, GL_E_STATE, "no state set");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    const size_t E = (size_t)P.n * (P.n - 1) / 2;
    if (coefficients)
        GL_TRY(gl_d2h(ctx, coefficients, P.C, (size_t)P.K * P.T * 8));
    if (weights) {
        k_gd_unpack_w<<<dim3((unsigned)(((size_t)P.n * P.n + 255) / 256)), 256, 0, ctx->stream>>>(s->stage, P);
        GL_HIP(ctx, hipGetLastError());
        GL_TRY(gl_d2h(ctx, weights, s->stage, (size_t)P.K * E * 8));
    }
    if (dual) GL_TRY(gl_d2h(ctx, dual, P.dual, (size_t)P.Tw * P.n * 8));
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GL_OK;
}

GL_API int gl_graphdict_set_params(gl_graphdict *s, double l1_w, double ortho_w, double l1_c, double log_c,
                                   double l1_diff_c, double step_c, double step_w, double step_dual) {
    GL_TRY(gd_check(s));
    if (!(step_c > 0) || !(step_w > 0) || !(step_dual > 0))
        return gl_set_error(s->ctx, GL_E_INVALID, "step sizes must be positive");
    GdDev &P = s->dev;
    const bool same = P.l1_w == l1_w && P.ortho_w == ortho_w && P.l1_c == l1_c && P.log_c == log_c &&
                      P.l1_diff_c == l1_diff_c && P.step_c == step_c && P.step_w == step_w && P.step_dual == step_dual;
    P.l1_w = l1_w, P.ortho_w = ortho_w, P.l1_c = l1_c, P.log_c = log_c, P.l1_diff_c = l1_diff_c;
    P.step_c = step_c, P.step_w = step_w, P.step_dual = step_dual;
    if (!same) gd_drop_graphs(s);  // the captured steps carry the old values in their kernel arguments
    return GL_OK;
}

// Precision of the two O(K T E) contractions of a step (base_model.py:375, :447): 64 (default: what the reference
// computes) or 32 (BASELINE configs[4] as written: fp32 operands and cell sums, packed two per instruction; every fold,
// the projections, the dual prox and the state itself stay fp64).  The single-workgroup engine of small problems
// ignores it (fp64).
GL_API int gl_graphdict_set_precision(gl_graphdict *s, int bits) {
    GL_TRY(gd_check(s));
    gl_ctx *ctx = s->ctx;
    if (bits != 32 && bits != 64) return gl_set_error(ctx, GL_E_INVALID, "precision must be 32 or 64, got %d", bits);
    GdDev &P = s->dev;
    GL_HIP(ctx, hipSetDevice(ctx->device));
    if (bits == 32 && !s->Xf) {
        GL_TRY(gd_alloc(s, &s->Xf, (size_t)P.Tp * P.n_pad));
        GL_TRY(gd_alloc(s, &s->XTf, (size_t)P.n_pad * P.Tp));
        GL_TRY(gd_alloc(s, &s->WSf, (size_t)P.n_pad * P.n_pad * P.KP));
        GL_TRY(gd_alloc(s, &s->CTf, (size_t)P.Tp * P.KP));
    }
    if (bits == 32 && s->have_x) {
        const int64_t cnt = (int64_t)P.Tp * P.n_pad;
        k_gd_to_f32<<<dim3(2048), 256, 0, ctx->stream>>>(P.X, s->Xf, cnt, nullptr);
        k_gd_to_f32<<<dim3(2048), 256, 0, ctx->stream>>>(P.XT, s->XTf, cnt, nullptr);
        GL_HIP(ctx, hipGetLastError());
    }
    if (s->f32 != (bits == 32)) gd_drop_graphs(s);  // the captured steps launch the other instantiations
    s->f32 = bits == 32;
    P.WSf = s->f32 ? s->WSf : nullptr;
    P.CTf = s->f32 ? s->CTf : nullptr;
    P.NG = s->f32 ? s->ng32 : s->ng64;
    gd_refresh_f32(s);
    GL_HIP(ctx, hipGetLastError());
    return GL_OK;
}

GL_API int gl_graphdict_get_steps(gl_graphdict *s, double *steps3_out) {
    GL_TRY(gd_check(s));
    if (!steps3_out) return gl_set_error(s->ctx, GL_E_INVALID, "out is NULL");
    steps3_out[0] = s->dev.step_c, steps3_out[1] = s->dev.step_w, steps3_out[2] = s->dev.step_dual;
    return GL_OK;
}

GL_API int gl_graphdict_engine(gl_graphdict *s) {
    if (gd_check(s) != GL_OK) return -1;
    return gd_small_applicable(s) ? 1 : 0;
}

// The loop of _single_fit (base_model.py:552-581).
GL_API int gl_graphdict_fit(gl_graphdict *s, int64_t max_iter, double tol, int reduce_step_on_plateau,
                            double *history_out, int64_t *steps_run_out, int64_t *converged_out, int *flags_out) {
    GL_TRY(gd_check(s));
    gl_ctx *ctx = s->ctx;
    if (!s->have_x || !s->have_state) return gl_set_error(ctx, GL_E_STATE, "fit needs set_signals and set_state first");
    if (max_iter < 0 || !history_out) return gl_set_error(ctx, GL_E_INVALID, "bad argument");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    GdDev &P = s->dev;
    if (gd_small_applicable(s)) {  // small problems: the whole loop in one launch of one workgroup
        const double sc = P.step_c;
        const int rc = gd_small_fit(s, max_iter, tol, reduce_step_on_plateau, history_out, steps_run_out, converged_out, flags_out);
        if (P.step_c != sc) gd_drop_graphs(s);  // the plateau rule changed the steps frozen in the captured graphs
        return rc;
    }
    double *st = (double *)ctx->pinned;
    int64_t converged = -1, i = 0;
    int flags = 0;
    for (int64_t q = 0; q < 2 * max_iter; ++q) history_out[q] = -1.0;
    if (P.hist_cap < max_iter) {  // history of the run on the device; the captured steps hold its address
        double *h = nullptr;
        GL_MALLOC(ctx, (void **)&h, (size_t)std::max<int64_t>(max_iter, 1) * 2 * sizeof(double), "graphdict");
        if (P.hist) {
            auto itp = std::find(s->allocs.begin(), s->allocs.end(), (void *)P.hist);
            if (itp != s->allocs.end()) s->allocs.erase(itp);
            GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
            (void)hipFree(P.hist);
        }
        P.hist = h;
        P.hist_cap = max_iter;
        s->allocs.push_back(h);
        gd_drop_graphs(s);
    }
    // status words: [2] all-zero flag (sticky over the run), [4] steps done, [5] done, [6] tol, [7] converged_
    const double init[6] = {0.0, 0.0, 0.0, 0.0, tol, -1.0};
    GL_TRY(gl_h2d(ctx, P.stat + 2, init, sizeof init));
    const int par0 = s->step_parity;
    if (!reduce_step_on_plateau) {
        // steps are enqueued a chunk at a time; the device keeps the books (k_gd_coef_stat) and, once the rule
        // is met, the rest of the chunk falls through -- no host round trip per step
        int64_t enq = 0;
        bool done = false;
        while (!done && enq < max_iter) {
            const int64_t chunk = std::min<int64_t>(max_iter - enq, enq == 0 ? 4 : 16);
            for (int64_t c = 0; c < chunk; ++c) GL_TRY(gd_step_replayed(s, 1));
            enq += chunk;
            GL_HIP(ctx, hipMemcpyAsync(st, P.stat, 8 * 8, hipMemcpyDeviceToHost, ctx->stream));
            GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
            done = st[5] != 0.0;
        }
        i = (int64_t)st[4];
        converged = (int64_t)st[7];
        if (st[2] != 0.0) flags |= 1;  // "All coefficients dropped to 0"
        if (((enq - i) & 1) != 0) {  // the steps that fell through flipped the host's view of the degree ping-pong only
            std::swap(P.deg, P.deg_new);
            s->step_parity ^= 1;
        }
        if (i > 0) GL_TRY(gl_d2h(ctx, history_out, P.hist, (size_t)(2 * i) * 8));
    } else {
        for (i = 0; i < max_iter; ++i) {
            GL_TRY(gd_step_replayed(s, 1));
            GL_HIP(ctx, hipMemcpyAsync(st, P.stat, 8 * 8, hipMemcpyDeviceToHost, ctx->stream));
            GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
            const double dc = st[0], dw = st[1];
            history_out[2 * i] = dc;
            history_out[2 * i + 1] = dw;
            if (dc < tol && dw < tol) converged = i;
            if (i > 100) {  // :563-574
                double mc = 0, mw = 0;
                for (int64_t q = i - 10; q < i; ++q) {
                    mc += history_out[2 * q];
                    mw += history_out[2 * q + 1];
                }
                if (dc >= mc / 10 && dw >= mw / 10) {
                    P.step_c *= 0.7, P.step_w *= 0.7, P.step_dual *= 0.7;
                    gd_drop_graphs(s);  // the step sizes are frozen in the captured kernels' arguments
                    flags |= 2;  // "Divergence detected ..., reducing step size"
                }
            }
            if (converged > 0) {  // sic: a stop at step 0 is not taken (:578)
                ++i;
                break;
            }
        }
        if (st[2] != 0.0) flags |= 1;
    }
    (void)par0;
    GL_HIP(ctx, hipMemsetAsync(P.stat + 5, 0, 8, ctx->stream));  // the kernels are usable again (set_state, predict, probes)
    if (steps_run_out) *steps_run_out = i;
    if (converged_out) *converged_out = converged;
    if (flags_out) *flags_out = flags;
    return GL_OK;
}

// predict (base_model.py:647-676): coefficients and dual only, weights fixed.  The session's
// coefficients are the start (K x T), its dual is reset to zero.  Returns in coefficients_out
// what the reference returns (the update that met the tolerance, or the last iterate).
GL_API int gl_graphdict_predict(gl_graphdict *s, int64_t max_iter, double tol, double *coefficients_out,
                                int64_t *steps_run_out, int *flags_out) {
    GL_TRY(gd_check(s));
    gl_ctx *ctx = s->ctx;
    if (!s->have_x || !s->have_state) return gl_set_error(ctx, GL_E_STATE, "predict needs set_signals and set_state first");
    if (max_iter < 0 || !coefficients_out) return gl_set_error(ctx, GL_E_INVALID, "bad argument");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    GdDev &P = s->dev;
    hipStream_t st_ = ctx->stream;
    double *st = (double *)ctx->pinned;
    GL_HIP(ctx, hipMemsetAsync(P.dual, 0, (size_t)P.Tw * P.n * 8, st_));
    GL_HIP(ctx, hipMemsetAsync(P.stat + 2, 0, 6 * 8, st_));
    if (gd_small_applicable(s)) {
        GL_TRY(gd_small_predict(s, max_iter, tol, steps_run_out, flags_out));
        GL_TRY(gl_d2h(ctx, coefficients_out, P.C, (size_t)P.K * P.T * 8));
        return GL_OK;
    }
    GL_TRY(gd_launch_grad_c(s));  // the weights never change: one pass over the pairs
    k_gd_sc_reduce<<<dim3((unsigned)(((int64_t)P.KP * P.Tp + 63) / 64)), dim3(64, 4), 0, st_>>>(P);
    int flags = 0;
    int64_t i = 0;
    for (i = 0; i < max_iter; ++i) {
        GL_HIP(ctx, hipMemsetAsync(P.stat + 2, 0, 8, st_));
        if (P.log_c > 0) k_gd_logc_reset<<<dim3((unsigned)((P.Tw + 255) / 256)), 256, 0, st_>>>(P);
        gd_launch_coef_update(P, st_);
        k_gd_coef_finish<<<1, 256, 0, st_>>>(P);
        gd_launch_dual_update(P, st_, 1);
        const int cb = gd_commit_blocks(P);
        k_gd_coef_commit<<<dim3((unsigned)cb), 256, 0, st_>>>(P);
        k_gd_coef_stat<<<1, 64, 0, st_>>>(P, cb, 0);
        GL_HIP(ctx, hipGetLastError());
        GL_HIP(ctx, hipMemcpyAsync(st, P.stat, 4 * 8, hipMemcpyDeviceToHost, st_));
        GL_HIP(ctx, hipStreamSynchronize(st_));
        if (st[2] != 0.0) flags |= 1;
        if (st[3] < tol) {
            ++i;
            break;
        }
    }
    GL_HIP(ctx, hipMemcpyAsync(coefficients_out, P.C, (size_t)P.K * P.T * 8, hipMemcpyDeviceToHost, st_));
    GL_HIP(ctx, hipStreamSynchronize(st_));
    if (steps_run_out) *steps_run_out = i;
    if (flags_out) *flags_out = flags;
    return GL_OK;
}

// Measurement facility (tools / bench): average duration in microseconds of the two gradient
// kernels alone, HIP events on the session's stream; state unchanged.
GL_API int gl_graphdict_time_gradients(gl_graphdict *s, int reps, double *grad_c_us_out, double *grad_w_us_out) {
    GL_TRY(gd_check(s));
    gl_ctx *ctx = s->ctx;
    if (!s->have_x || !s->have_state || reps < 1) return gl_set_error(ctx, GL_E_STATE, "needs signals, state, reps >= 1");
    GL_HIP(ctx, hipSetDevice(ctx->device));
    float ms = 0;
    for (int which = 0; which < 2; ++which) {
        GL_TRY(which ? gd_launch_grad_w(s) : gd_launch_grad_c(s));
        GL_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        for (int r = 0; r < reps; ++r) GL_TRY(which ? gd_launch_grad_w(s) : gd_launch_grad_c(s));
        GL_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        GL_HIP(ctx, hipEventSynchronize(ctx->ev1));
        GL_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        double *o = which ? grad_w_us_out : grad_c_us_out;
        if (o) *o = (double)ms * 1e3 / reps;
        if (which && s->dbgbuf) {
            const GdDev &P = s->dev;
            const size_t nw = (size_t)P.nwtasks * P.TS * 4;
            std::vector<double> d(4 * nw);
            (void)hipMemcpy(d.data(), s->dbgbuf, d.size() * 8, hipMemcpyDeviceToHost);
            double t0 = 1e300, t1 = 0, dur = 0;
            size_t cnt = 0;
            for (size_t w = 0; w < nw; ++w)
                if (d[4 * w + 1] > 0) t0 = std::min(t0, d[4 * w]), t1 = std::max(t1, d[4 * w + 1]), dur += d[4 * w + 1] - d[4 * w], ++cnt;
            int nb8 = 0, nbc8 = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb8, k_gd_grad_w<double, 8, 8>, 256, 0);
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbc8, k_gd_grad_c<double, 8, 2>, 64, 0);
            fprintf(stderr, "[grad_w probe] runtime occupancy: grad_w<8,8> %d workgroups/CU, grad_c<8> %d\n", nb8, nbc8);
            fprintf(stderr, "[grad_w probe] %zu waves, span %.1f us, mean wave %.1f us\n", cnt, (t1 - t0) / 100, dur / cnt / 100);
            // concurrency over time (20 buckets) and waves per (xcc, cu, simd) slot
            int hist[20] = {0};
            for (size_t w = 0; w < nw; ++w)
                if (d[4 * w + 1] > 0)
                    for (int b = 0; b < 20; ++b) {
                        const double tb = t0 + (t1 - t0) * (b + 0.5) / 20;
                        if (d[4 * w] <= tb && tb < d[4 * w + 1]) ++hist[b];
                    }
            fprintf(stderr, "[grad_w probe] resident waves over time:");
            for (int b = 0; b < 20; ++b) fprintf(stderr, " %d", hist[b]);
            fprintf(stderr, "\n");
            std::vector<int> per(8 * 4096, 0);
            for (size_t w = 0; w < nw; ++w)
                if (d[4 * w + 1] > 0) {
                    const unsigned hw = (unsigned)d[4 * w + 2];
                    const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                    per[(size_t)d[4 * w + 3] * 4096 + ((se * 2 + sh) * 16 + cu) * 4 + simd]++;
                }
            int used = 0, mx = 0, mn = 1 << 30;
            for (int v : per)
                if (v) ++used, mx = std::max(mx, v), mn = std::min(mn, v);
            fprintf(stderr, "[grad_w probe] distinct (xcc, se, sh, cu, simd) slots used %d, waves per slot min %d max %d\n", used, mn, mx);
        }
    }
    return GL_OK;
}

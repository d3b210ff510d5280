// Cross-lane reductions on the VALU for gfx950 (wave64).
//
// __shfl / __shfl_xor compile to ds_bpermute_b32: every exchange is a round trip through the LDS
// crossbar (~100+ cycles before the dependent add can issue), which leaves reduction butterflies
// latency-bound in kernels that run at 1-2 waves per SIMD.  DPP modifiers (row_ror, row_half_mirror,
// quad_perm) and the gfx950 v_permlane16_swap / v_permlane32_swap do the same exchanges in a few
// cycles.  Lane maps verified on hardware with tools/dpp_probe.hip.
#pragma once

#include <hip/hip_runtime.h>

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const long long b = __double_as_longlong(v);
    // bound_ctrl with full row / bank masks: every lane has a valid source for the controls used here, and the
    // compiler does not have to materialise the "old" operand (a v_mov 0 per half otherwise)
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
constexpr int DPP_ROW_ROR8 = 0x128;     // lane i <- lane i ^ 8 (rows of 16 lanes)
constexpr int DPP_HALF_MIRROR = 0x141;  // lane i <- lane 7 - i (groups of 8 lanes)
constexpr int DPP_QUAD_XOR2 = 0x4E;     // quad_perm [2,3,0,1]
constexpr int DPP_QUAD_XOR1 = 0xB1;     // quad_perm [1,0,3,2]

// v_permlane16_swap(X, Y): X's odd 16-lane rows <-> Y's even rows.  The sum of its two results is
//   even-row lane l:  x[l] + x[l + 16]        odd-row lane l:  y[l - 16] + y[l]
// i.e. one halving step of the butterfly (even rows keep x, odd rows keep y) with no select at all.
__device__ __forceinline__ double swap16_halve(double x, double y) {
    const long long xb = __double_as_longlong(x), yb = __double_as_longlong(y);
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)xb, (unsigned)yb, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(xb >> 32), (unsigned)(yb >> 32), false, false);
    const double r0 = __longlong_as_double(((long long)hi[0] << 32) | lo[0]);
    const double r1 = __longlong_as_double(((long long)hi[1] << 32) | lo[1]);
    return r0 + r1;
}
// v + (v of lane ^ 32) in every lane, through v_permlane32_swap (lower / upper half exchange)
__device__ __forceinline__ double add_xor32(double v) {
    const long long b = __double_as_longlong(v);
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    const double r0 = __longlong_as_double(((long long)hi[0] << 32) | lo[0]);
    const double r1 = __longlong_as_double(((long long)hi[1] << 32) | lo[1]);
    return r0 + r1;
}

// sum over the 32 lanes of each half-wave (result in every lane of the half): quad, 8, 16, 32 lanes
__device__ __forceinline__ double half_sum(double v) {
    v += dpp_mov<DPP_QUAD_XOR1>(v);
    v += dpp_mov<DPP_QUAD_XOR2>(v);
    v += dpp_mov<DPP_HALF_MIRROR>(v);
    v += dpp_mov<DPP_ROW_ROR8>(v);
    return swap16_halve(v, v);  // x = y: v[l] + v[l ^ 16]
}
// sum over all 64 lanes (result in every lane)
__device__ __forceinline__ double wave_sum(double v) { return add_xor32(half_sum(v)); }

// maximum over all 64 lanes (result in every lane): the butterfly of wave_sum with fmax
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_mov<DPP_QUAD_XOR1>(v));
    v = fmax(v, dpp_mov<DPP_QUAD_XOR2>(v));
    v = fmax(v, dpp_mov<DPP_HALF_MIRROR>(v));
    v = fmax(v, dpp_mov<DPP_ROW_ROR8>(v));
    {
        const long long b = __double_as_longlong(v);
        const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
        v = fmax(__longlong_as_double(((long long)hi[0] << 32) | lo[0]), __longlong_as_double(((long long)hi[1] << 32) | lo[1]));
    }
    {
        const long long b = __double_as_longlong(v);
        const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
        v = fmax(__longlong_as_double(((long long)hi[0] << 32) | lo[0]), __longlong_as_double(((long long)hi[1] << 32) | lo[1]));
    }
    return v;
}

// Sums of 8 per-lane values over the 32 lanes of a half-wave, "halving" butterfly: every step
// exchanges half of what is left, so 4 + 2 + 1 + 1 + 1 = 9 exchanges instead of 8 x 5.  Afterwards
// lane l holds the total of value ((l >> 2) & 7) of its half-wave.
__device__ __forceinline__ double halfwave_rows8(const double (&v)[8], int lane) {
    const bool b3 = lane & 8, b2 = lane & 4;
    double a[4], b[2], c;
#pragma unroll
    for (int m = 0; m < 4; ++m) a[m] = swap16_halve(v[m], v[m + 4]);  // even rows keep m, odd rows m + 4
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const double keep = b3 ? a[m + 2] : a[m];
        const double send = b3 ? a[m] : a[m + 2];
        b[m] = keep + dpp_mov<DPP_ROW_ROR8>(send);
    }
    {
        const double keep = b2 ? b[1] : b[0];
        const double send = b2 ? b[0] : b[1];
        c = keep + dpp_mov<DPP_HALF_MIRROR>(send);  // partner 7 - i instead of i ^ 4: also across bit 2
    }
    c += dpp_mov<DPP_QUAD_XOR2>(c);
    c += dpp_mov<DPP_QUAD_XOR1>(c);
    return c;
}

// Same for 4 values per half-wave: 2 + 1 halving exchanges, then 3 full ones.  Afterwards lane l holds
// the total of value ((l >> 3) & 3).
__device__ __forceinline__ double halfwave_rows4(const double (&v)[4], int lane) {
    const bool b3 = lane & 8;
    double a[2], c;
#pragma unroll
    for (int m = 0; m < 2; ++m) a[m] = swap16_halve(v[m], v[m + 2]);
    {
        const double keep = b3 ? a[1] : a[0];
        const double send = b3 ? a[0] : a[1];
        c = keep + dpp_mov<DPP_ROW_ROR8>(send);
    }
    c += dpp_mov<DPP_HALF_MIRROR>(c);
    c += dpp_mov<DPP_QUAD_XOR2>(c);
    c += dpp_mov<DPP_QUAD_XOR1>(c);
    return c;
}

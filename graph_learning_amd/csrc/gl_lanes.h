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

// t where the lane's bit of the CONSTANT 64-lane mask M is set, else f.  The masks of the butterflies below are lane
// patterns ((lane & 8) != 0, (lane & 4) != 0); derived from the lane number they are live SGPR pairs that the compiler,
// short of scalar registers in the tile engine's pass, spills to VGPR lanes and reads back with v_readlane + s_nop before
// every select (34 + 27 instructions per pass, measured from the assembly).  As literals they cost an s_mov_b64 each.
template <unsigned long long M>
__device__ __forceinline__ double lane_select(double f, double t) {
    const long long fb = __double_as_longlong(f), tb = __double_as_longlong(t);
    int lo, hi;
    const unsigned long long m = M;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"((int)fb), "v"((int)tb), "s"(m));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"((int)(fb >> 32)), "v"((int)(tb >> 32)), "s"(m));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
constexpr unsigned long long LANES_BIT3 = 0xFF00FF00FF00FF00ull;  // (lane & 8) != 0
constexpr unsigned long long LANES_BIT2 = 0xF0F0F0F0F0F0F0F0ull;  // (lane & 4) != 0

// Sums of 8 per-lane values over the 32 lanes of a half-wave, "halving" butterfly: every step
// exchanges half of what is left, so 4 + 2 + 1 + 1 + 1 = 9 exchanges instead of 8 x 5.  Afterwards
// lane l holds the total of value ((l >> 2) & 7) of its half-wave.
__device__ __forceinline__ double halfwave_rows8(const double (&v)[8], int lane) {
    (void)lane;
    double a[4], b[2], c;
#pragma unroll
    for (int m = 0; m < 4; ++m) a[m] = swap16_halve(v[m], v[m + 4]);  // even rows keep m, odd rows m + 4
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const double keep = lane_select<LANES_BIT3>(a[m], a[m + 2]);
        const double send = lane_select<LANES_BIT3>(a[m + 2], a[m]);
        b[m] = keep + dpp_mov<DPP_ROW_ROR8>(send);
    }
    {
        const double keep = lane_select<LANES_BIT2>(b[0], b[1]);
        const double send = lane_select<LANES_BIT2>(b[1], b[0]);
        c = keep + dpp_mov<DPP_HALF_MIRROR>(send);  // partner 7 - i instead of i ^ 4: also across bit 2
    }
    c += dpp_mov<DPP_QUAD_XOR2>(c);
    c += dpp_mov<DPP_QUAD_XOR1>(c);
    return c;
}

// Same for 4 values per half-wave: 2 + 1 halving exchanges, then 3 full ones.  Afterwards lane l holds
// the total of value ((l >> 3) & 3).
__device__ __forceinline__ double halfwave_rows4(const double (&v)[4], int lane) {
    (void)lane;
    double a[2], c;
#pragma unroll
    for (int m = 0; m < 2; ++m) a[m] = swap16_halve(v[m], v[m + 2]);
    {
        const double keep = lane_select<LANES_BIT3>(a[0], a[1]);
        const double send = lane_select<LANES_BIT3>(a[1], a[0]);
        c = keep + dpp_mov<DPP_ROW_ROR8>(send);
    }
    c += dpp_mov<DPP_HALF_MIRROR>(c);
    c += dpp_mov<DPP_QUAD_XOR2>(c);
    c += dpp_mov<DPP_QUAD_XOR1>(c);
    return c;
}

// Tile engine of the log-degree loop: ALL iterations of a call in one persistent launch, the edge iterate w
// in REGISTERS, the distances z in LDS (and a short register ring for what does not fit), and every
// inter-workgroup dependency carried by self-validating tagged records instead of grid barriers.
//
// Reference path: graph_learn/smooth_learning.py:146-167 (same per-edge arithmetic as gl_logdeg.hip).
//
// Why (measured on the first resident engine, gl_resident.hip, n = 4096: 23.4 us per iteration): its edge
// phase (11.6 us) ran at the pace of the z stream (67 MB per iteration = 5.8 TB/s), and the other half of
// an iteration was synchronisation: two grid barriers, a node phase on 64 of 252 workgroups, a stopping rule
// behind a third round trip.  Here
//   * one workgroup of 8 waves per CU (two per SIMD: a lone wave issues fp64 at half rate -- measured, 410
//     cycles per row pair against 200), 256 registers each.  The unit of work is a CHUNK: U = 32, 16 or 8
//     consecutive row pairs (2 x 64 edges each) of one 64 x 64 tile; a wave owns one chunk and keeps its w in
//     4 U registers for the whole call.  The workgroup's LDS holds 128 row pairs of z: everything when
//     U <= 16 (nothing streams at all), 16 of a tile's 32 row pairs when U = 32 (n up to 4096) -- the other 16
//     pass through a 4-deep register ring that is refilled during the first half of the pass, so the memory
//     pipe is quiet when the hand-offs start.  LDS-resident z is stored pre-multiplied (2 gamma z);
//   * the diagonal tiles ("fringe") are dealt out as groups of <= 2 row pairs to waves that already work on
//     the same node block; their w and z stay in registers, their degree partials are folded into the wave's
//     own partial vectors;
//   * degree partials are pre-reduced inside the workgroup (LDS, fixed order) to ONE record per node block
//     the workgroup touches (6-8 instead of 16), published as 16-byte granules {low word, tag}{high word, tag}
//     with write-through stores; the owner of a node block (workgroup b owns block b) re-reads its records
//     until every tag matches, updates the dual state -- which lives in the registers of its wave 0 -- and
//     publishes {d, pbar} of its 64 nodes the same way; edge workgroups re-read the 6-8 blocks they need.
//     One memory round trip per hand-off, no barrier, no flag ("the data is the flag", recipe R2 of the
//     CDNA4 guide; the LGMRF ADMM of gl_lgmrf.hip uses the same records);
//   * the stopping rule of iteration k is evaluated by every workgroup at the top of iteration k + 1, from
//     records that arrive in the same round trip as the node values, before anything is modified -- and not
//     at all while tol = 0 (fixed-iteration calls), except once at the end for the values reported; the norms
//     themselves are only formed in the iterations whose rule is evaluated.
//   * past n = 4096 the lattice no longer fits the register files: a workgroup then takes a rectangle of tiles and works
//     through it in ROUNDS, one tile per wave from registers and LDS as above, the others streamed through register
//     rings (see k_tile, BIGV); the records, the owners and the rule are the same.
// Every sum has a fixed order (no atomics): bit-reproducible run to run.
//
// Arithmetic of the pass.  The reference forms, per edge (smooth_learning.py:147-163),
//     y = c w - g (d_i + d_j),  p = max(0, y - 2 g z),  q = c p - g (pbar_i + pbar_j),  w' = w - y + q
// and per node the degree sums S p and S w'.  Substituting y and q into w' gives the same real number as
//     w' = (1 - c) w + c p + g (u_i + u_j),        u = d - pbar,
// so the node values that travel are {g d, g u}; and summing THAT over the edges of node i gives
//     S w'_i = (1 - c) S w_i + c S p_i + (n - 2) u_i + sum_j u_j:
// the degree of the new iterate follows from node quantities alone.  The pass therefore forms one degree sum (S p), not
// two, and 7 fp64 operations per edge instead of 8: 18.5 instead of 26 fp64 instructions per row pair and lane, half the
// records.  The recurrence contracts ((1 - c) = 2 gamma beta ~ 0.01), so its rounding does not accumulate; iterates stay
// within ~1e-14 of the oracle (1e-10 is the parity contract).  The distances arrive pre-multiplied
// (zg = 2 gamma z, formed once per gl_logdeg_init) with +1e300 at every entry outside i < j < n: p is exactly 0 there
// without a mask.
//
// Nothing is written back before the end of the call (w lives in registers, the dual state in registers of
// the owners), so a launch whose workgroups are not all co-resident (start-up rendezvous, bounded by the wall
// clock) leaves the session untouched and the host redoes the call on the streaming engine.
#include <algorithm>
#include <type_traits>
#include <vector>

#include "gl_gridbar.h"
#include "gl_lanes.h"
#include "gl_logdeg.h"
#include "gl_tile_plan.h"

#ifndef GL_TILE_LDS_PREFETCH
#define GL_TILE_LDS_PREFETCH 1  // request the LDS operands of row pair s + 1 before row pair s is computed
#endif
#ifndef GL_TILE_PRIO_EIGHTHS
#define GL_TILE_PRIO_EIGHTHS 4  // the younger wave of a SIMD runs this many eighths of its pass at priority 1
#endif
#ifndef GL_TILE_PIN
#define GL_TILE_PIN 2  // row pairs between two scheduling boundaries of the pass (independent work for the fp64 pipe)
#endif
#ifndef GL_TILE_PROBES
#define GL_TILE_PROBES 1      // probes a consumer of node values / an owner keeps in flight (2: see the acquire)
#endif
#ifndef GL_TILE_PROBE_GAP
#define GL_TILE_PROBE_GAP 12  // s_sleep units (64 cycles) between the two: about half a round trip
#endif
#ifndef GL_TILE_ROWGROUP
#define GL_TILE_ROWGROUP 8     // row pairs whose row sums share one halving butterfly (4: 1.5 adds per row pair, 8: 1.125)
#endif
#ifndef GL_TILE_ROUNDS_ZEV
#define GL_TILE_ROUNDS_ZEV 4         // rounds: every 4th row pair of the resident tile keeps its z in LDS (25 row pairs of w in registers) ...
                                     // (8: 21 row pairs in registers; measured 15.1 / 28.3 / 105.8 us per iteration at n = 4160 / 5000 / 8192 against 13.8 / 27.0 / 104.2)
#endif
#ifndef GL_TILE_ROUNDS_ZEV_NORMS
#define GL_TILE_ROUNDS_ZEV_NORMS 64  // ... none in the variants with the norms (more row pairs of w in LDS instead)
#endif
#ifndef GL_TILE_EARLY_STORE
#define GL_TILE_EARLY_STORE 1  // fixed-iteration calls store w right after their last pass, underneath the last hand-off
#endif
#ifndef GL_TILE_EXACT_HEAD
#define GL_TILE_EXACT_HEAD 3  // the stopping rule is evaluated exactly (no lag) for this many iterations at the start of a call
#endif

namespace {

using namespace gl_tile_plan;
static_assert(TB == GL_TB, "tile edge");

typedef unsigned int tl_u4 __attribute__((ext_vector_type(4)));

constexpr double ZG_OUTSIDE = 1e300;  // zg at entries outside i < j < n: p = max(0, y - zg) is exactly 0 there

struct TileDev {
    double *w;        // tiled iterate buffer: read at the start (rounds: updated in place)
    double *wout;     // ... where the registers are written at the end (w itself, or the other ping-pong buffer)
    const double *zg; // tiled 2 gamma z, ZG_OUTSIDE outside i < j < n
    double *dvar, *ybar, *pbar, *dprev;
    double *bnorm;    // the streaming engine's [2][nb][4] norm partials (kept consistent at the end)
    double *dstat;
    long long *ctr;
    const WaveDesc *wd;
    const WgDesc *gd;
    tl_u4 *part;   // [n_entries][64] granules: degree partials S p of one workgroup for one node block
    tl_u4 *nval;   // [nb][2][64] granules: {gamma d, gamma (d - pbar)} of a node block
    tl_u4 *usum;   // [2][nb] granules: sum over a node block of gamma (d - pbar) (ring by tag parity)
    tl_u4 *wnorm;  // [4][NWG][2] granules: {sum (q-y)^2, sum w^2} of a workgroup (ring by iteration)
    tl_u4 *bnrm;   // [4][nb][4] granules: norm partials of a node block and of its slice of wnorm
    tl_u4 *look;   // [4][4] granules: the rule's early look of one workgroup for all ({code, ratio, ratio, -}; ring by iteration)
    int rule_wg;   // the workgroup that takes it (one without a node block of its own where there is one)
    unsigned *bar;  // rendezvous words (gl_gridbar.h layout)
    unsigned long long *dbg;  // development only (GRAPHLEARN_AMD_RES_STAMPS=1): per-workgroup phase stamps
    int64_t n, nb;
    int NWG, n_entries;
    double gamma, c, omc, nm2, a4g, tol;  // omc = 1 - c (exact: c is in [1/2, 2]), nm2 = n - 2
    // The variants WITHOUT the norms, tol > 0 (gl_logdeg_iterate's middle phase): the launch runs while the DUAL ratio alone
    // rules the stopping rule out -- ||qbar - ybar|| / ||d|| >= gate (= a small multiple of tol; smooth_learning.py:159,166:
    // both ratios must be below tol) -- and ends (ctr[0] = 2) once it comes below it; the primal norms, a third of the pass,
    // are not formed at all.  0: no gate (a fixed number of iterations).
    double gate;
    long long it0;
    int maxit;
    unsigned tag0;
    unsigned rendezvous;  // value the arrival counter reaches when every workgroup of this launch has arrived
    int check_rendezvous; // wait for that at the start also below the rounds (test hook)
    // BATCH (k_tile<..., BATCH = true>): the lattice is block-diagonal -- B graphs of n nodes, graph g = node blocks
    // g nb1 .. g nb1 + nb1 - 1, tiles g ntiles1 .., workgroups g NWG1 ..; n, nm2 are those of ONE graph; nb, NWG the totals
    int nb1, NWG1;
    int64_t ntiles1;
    double t2lo;          // the coupled rule's screen: tol^2 (1 + 1e-9)
    // ... and, where the exact coupled rule is judged afterwards (gl_logdeg.hip: the deciding chunks): every iteration's new
    // iterate is also written out, iteration it of the call to slot it + 1 of a ring of whole tiled batches (slot 0: the
    // call's start, by the host) and its dual state {d, ybar, pbar, dprev, norm partials} to the same slot of dring; null: not
    double *ring, *dring;
    int64_t ring_stride, dring_stride;  // doubles per slot
};

__device__ __forceinline__ double tl_val(tl_u4 g) {
    return __longlong_as_double((long long)(((unsigned long long)g.z << 32) | g.x));
}
__device__ __forceinline__ tl_u4 tl_gran(double v, unsigned tag) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    tl_u4 g;
    g.x = (unsigned)b, g.y = tag, g.z = (unsigned)(b >> 32), g.w = tag;
    return g;
}
__device__ __forceinline__ bool tl_ok(tl_u4 g, unsigned tag) { return g.y == tag && g.w == tag; }
// v when bit 0 of `bits` is set, else +0.0 -- by an AND, so that a wave-uniform mask costs no branch
__device__ __forceinline__ double tl_keep(double v, int bits) {
    return __longlong_as_double(__double_as_longlong(v) & -(long long)(bits & 1));
}
__device__ __forceinline__ double lane_bcast(double v, int l) {  // lane l's value in every lane
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// a batch's screen of the coupled rule on s = {max_g ||dW_g||^2, sum_g ||W_g||^2, max_g ||dD_g||^2, sum_g ||D_g||^2}: is the rule
// surely not met; the larger lower bound over tol, squared (0: degenerate).  (Plain functions: as lambdas that capture the
// kernel's argument block inside other lambdas they made the compiler keep that block -- 330 bytes -- in scratch.)
__device__ __forceinline__ bool tl_screen_sure(const double (&v)[4], const double t2lo) {
    return v[1] > 0 && v[3] > 0 && (v[0] > t2lo * v[1] || v[2] > t2lo * v[3]);
}
__device__ __forceinline__ double tl_screen_ahead(const double (&v)[4], const double t2lo) {
    return (v[1] > 0 && v[3] > 0) ? fmax(v[0] / v[1], v[2] / v[3]) / t2lo : 0.0;
}

// false: give up (error word raised by somebody, or one second without progress)
__device__ __forceinline__ bool tl_spin(unsigned &spins, unsigned long long t0, unsigned *err) {
    __builtin_amdgcn_s_sleep(2);
    if ((++spins & 63u) == 0u) {
        if (gl_ld_u32(err) != 0u) return false;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}

// 16 bytes per lane from global memory straight into LDS (LDS-DMA: lane l's bytes land at lds + 16 l, no register in
// between).  The builtin exists in the device pass only; the host pass, which parses kernel bodies too, sees an empty body.
__device__ __forceinline__ void tl_dma16(const double *g, double2 *lds) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(g, lds, 16, 0, 0);
#endif
}

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() also waits for every vector-memory operation of
// the wave (its workgroup-scope fences cover global memory: s_waitcnt vmcnt(0)): here that would drain the bulk loads of the
// prologue in front of the first pass, and the z ring at every hand-off.  Nothing in this kernel passes data between the
// waves of a workgroup through global memory.
__device__ __forceinline__ void tl_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool NT = false>
__device__ __forceinline__ double2 tl_ldz(const double *base, unsigned byte_off) {
    if (NT) {  // (rounds past the Infinity Cache: z leaves the cache to the streamed w, see tl_ld_stream)
        typedef double v2d __attribute__((ext_vector_type(2)));
        const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(reinterpret_cast<const char *>(base) + byte_off));
        return double2{v.x, v.y};
    }
    return *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(base) + byte_off);
}

// One edge (see the head of the file): ai, bi = {gamma d, gamma u} of the row node, aj, bj of the column node, gz = 2 gamma z
// (ZG_OUTSIDE outside i < j < n: then p = 0 exactly).  Returns p and t = c p + gamma (u_i + u_j); the new iterate is
// w' = (1 - c) w + t and the update q - y = w' - w = t - c w: both are formed from the OLD w and t, so that w' can take the
// old w's register (formed as w' - w, the difference keeps both alive and the register allocator rotates the whole iterate
// through a second set of registers: measured, 13 us per pass instead of 4.5).  5 + 1 (+ 1 with the norms) fp64 instructions.
__device__ __forceinline__ void tl_edge(const double c, const double w, const double gz, const double ai, const double bi,
                                        const double aj, const double bj, double &pr, double &t) {
    const double y = __builtin_fma(c, w, -(ai + aj));   // :147
    const double tt = y - gz;                           // :150
    pr = tt > 0 ? tt : 0.0;
    t = __builtin_fma(c, pr, bi + bj);                  // :153 without its c p ... see above
}
// w <- w' = (1 - c) w + t (:163), IN PLACE, and the sums of squares of the stopping rule (:158-159) from the old w and t
// (fused multiply-adds: these sums are compared with tol, not with NumPy bit for bit; its norm sums in another order anyway).
// The update is written as the three-address instruction itself: the compiler's choice, v_fmac with t as the accumulator,
// leaves w' in t's register and pays a 64-bit move per edge to bring it home (and, in the variants with the norms, runs out
// of registers over it).  volatile: left alone the scheduler sinks these to the end of the unrolled pass.
template <bool NORMS>
__device__ __forceinline__ void tl_new_w(const double c, const double omc, double &w, const double t, double &nqy, double &nw) {
    if (NORMS) {
        const double dq = __builtin_fma(-c, w, t);      // q - y
        nqy = __builtin_fma(dq, dq, nqy);
    }
    asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(w) : "s"(omc), "v"(t));
    if (NORMS) nw = __builtin_fma(w, w, nw);            // of the NEW iterate: next iteration's denominator (see s_wcarry)
}

// One chunk: U row pairs of a bulk tile, w in registers.  With U = 32 the even row pairs come through the
// register ring (refilled RING streamed row pairs = 2 RING row pairs ahead: the odd ones in between double the
// lead time a register buys; the last refills of a pass are next iteration's first row pairs: z never changes),
// the odd ones -- and everything when U <= 16 -- from the chunk's LDS slice.  Row sums of p by a halving butterfly per 8
// row pairs, column sums in registers; everything leaves through LDS (row p, column p vectors of 64; rows are those of
// the whole tile, the chunk fills its own 2 U of them).
// The LDS operands of row pair s + 1 are requested before row pair s is computed: the empty asm after every
// row pair (it pins the running sums, which LLVM otherwise sinks to the end of the unrolled body, spilling every
// addend on the way) is a scheduling boundary.
// MASKW (variants with the norms on a ragged lattice): the new w of my two columns is multiplied by m0 / m1 (0 outside
// j < n), so that the sums of squares see zeros there; without the norms such entries simply carry a bounded value
// nobody reads (p is 0 there anyway, and the write-back masks them).
// WRAP = false (rounds): the ring is filled by the caller right before the pass and is dead after it.
template <int U, bool MASKW, bool NORMS, bool WRAP = true, bool ZNT = false>
__device__ __forceinline__ void tl_chunk_pass(const TileDev &P, double2 (&wreg)[U], double2 (&zr)[RING], const double2 *zl,
                                              const unsigned zb, const double2 *s_rows, const double aj0, const double aj1,
                                              const double bj0, const double bj1, const double m0, const double m1,
                                              double *s_row, double (&col)[2], const int lane, const bool younger, double &nqy,
                                              double &nw) {
    constexpr int NS = U == SLOTS ? SLOTS / 2 : 0;  // streamed row pairs
    constexpr int GR = GL_TILE_ROWGROUP;            // row pairs per butterfly (4 or 8)
    const int half = lane >> 5;
    const double c = P.c, omc = P.omc;
    double colp0 = col[0], colp1 = col[1];
    // not loop-invariant for the compiler: otherwise it hoists one 64-bit address per streamed row pair (and the
    // LDS addresses) out of the iteration loop, for each instantiation of this pass, and spills them all
    unsigned zbv = zb;
    asm volatile("" : "+v"(zbv));
    double2 rv_n = double2{0.0, 0.0}, zl_n = double2{0.0, 0.0};
    if (GL_TILE_LDS_PREFETCH && WRAP) {
        rv_n = s_rows[half];
        if (NS == 0) zl_n = zl[0];  // (U = 32: row pair 0 streams)
    }
    // Two waves share a SIMD (waves v and v + 4); both at priority 0 the older one takes the issue slots it wants and
    // the younger the rest, then finishes alone at half the fp64 rate (measured: 5.8 against 8.1 us at n = 4096).  The
    // younger half therefore runs the first half of its pass at priority 1: both end together.
    if (younger) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int gq = 0; gq < U / GR; ++gq) {
        double rowp[GR];
#pragma unroll
        for (int k = 0; k < GR; ++k) {
            const int s = GR * gq + k;
            if (s == U * GL_TILE_PRIO_EIGHTHS / 8) __builtin_amdgcn_s_setprio(0);
            double2 gz;  // 2 gamma z of this row pair (:150)
            constexpr bool STREAMS = NS > 0;
            const bool streamed = STREAMS && (s & 1) == 0;  // U = 32: even row pairs stream, odd ones live in LDS
            if (streamed) {
                const int pos = (s / 2) % RING;
                gz = zr[pos];
                // refill with the RING-th next streamed row pair (2 RING row pairs ahead; wraps into the next iteration)
                if (WRAP || s + 2 * RING < SLOTS) zr[pos] = tl_ldz<ZNT>(P.zg, zbv + (unsigned)(((s + 2 * RING) % SLOTS) * (2 * TB) * 8));
            } else {
                gz = zl_n;
            }
            double2 rv;
            if (GL_TILE_LDS_PREFETCH && WRAP) {
                rv = rv_n;  // {gamma d_i, gamma u_i}: one address per half-wave (broadcast)
                if (s + 1 < U) {
                    rv_n = s_rows[2 * (s + 1) + half];
                    if (!STREAMS) zl_n = zl[(s + 1) * 64];
                    else if (((s + 1) & 1) == 1) zl_n = zl[((s + 1) / 2) * 64];
                }
            } else {  // (rounds: the registers of the look-ahead are worth more than its 5 %)
                rv = s_rows[2 * s + half];
                if (!streamed) gz = zl[(STREAMS ? s / 2 : s) * 64];
            }
            double pp[2], t[2];
            tl_edge(c, wreg[s].x, gz.x, rv.x, rv.y, aj0, bj0, pp[0], t[0]);
            tl_edge(c, wreg[s].y, gz.y, rv.x, rv.y, aj1, bj1, pp[1], t[1]);
            if (MASKW) t[0] *= m0, t[1] *= m1;  // (w is 0 outside j < n and stays 0)
            tl_new_w<NORMS>(c, omc, wreg[s].x, t[0], nqy, nw);
            tl_new_w<NORMS>(c, omc, wreg[s].y, t[1], nqy, nw);
            colp0 += pp[0];
            colp1 += pp[1];
            rowp[k] = pp[0] + pp[1];
            if (k % GL_TILE_PIN == GL_TILE_PIN - 1) {
                if (NORMS)
                    asm volatile("" : "+v"(nqy), "+v"(nw), "+v"(colp0), "+v"(colp1));
                else
                    asm volatile("" : "+v"(colp0), "+v"(colp1));
            }
        }
#if GL_TILE_ROWGROUP == 8
        const double rp = halfwave_rows8(rowp, lane);  // lane l: the total of row pair (l >> 2) & 7 of the group
        if ((lane & 3) == 0) s_row[2 * (8 * gq + ((lane >> 2) & 7)) + half] = rp;  // (s_row points at the chunk's first row inside the tile's row vector)
#else
        const double rp = halfwave_rows4(rowp, lane);  // lane l: the total of row pair (l >> 3) & 3 of the group
        if ((lane & 7) == 0) s_row[2 * (4 * gq + ((lane >> 3) & 3)) + half] = rp;
#endif
    }
    col[0] = colp0, col[1] = colp1;
}

// The resident pass of the ROUNDS (n > 4096): as tl_chunk_pass<32>, but only WRG of the tile's 32 row pairs keep their
// w in registers; the last ones live in the wave's LDS slice, which in exchange holds z for every ZEV-th row pair only
// (the others come through the ring).  The rounds need registers beside the resident pass -- the streamed pass, its rings,
// the accumulators of the fold -- and a row pair of w that the compiler parks in scratch stalls the pass for a memory
// round trip per access (its reload sits in the same vmcnt queue as the z ring: n = 5000 with norms, 46 against 32 us).
// LDS slice of a wave (zw, 16 x 64 double2): [0, ZLN) 2 gamma z of row pairs ZEV - 1, 2 ZEV - 1, ..; [ZLN, 16) w of row pairs WRG .. 31.
constexpr int WRG_NONE = SLOTS;
template <bool NORMS>
struct RoundsSplit {
    static constexpr int SLICE = 15;                  // row pairs of LDS per wave (one less than below n = 4096: s_node holds 16 blocks)
    static constexpr int ZEV = NORMS ? GL_TILE_ROUNDS_ZEV_NORMS : GL_TILE_ROUNDS_ZEV;  // every ZEV-th row pair keeps its z in LDS (64: none)
    static constexpr int ZLN = SLOTS / ZEV;           // ... that many of them
    static constexpr int WRG = SLOTS - (SLICE - ZLN); // row pairs of w in registers; the others in LDS behind the z
    static constexpr int slot(int t) { return t + t / (ZEV - 1); }  // the t-th streamed row pair
};
// WRAPR (variants without the norms: they have the 16 registers): the ring's last refills are next iteration's first row
// pairs, so the ring is full when the pass starts (filled right before the pass it costs a memory round trip per iteration).
template <bool MASKW, bool NORMS, bool ZNT, bool WRAPR>
__device__ __forceinline__ void tl_resident_pass_rounds(const TileDev &P, double2 (&wreg)[SLOTS], double2 (&zr)[RING], double2 *zw,
                                                        const unsigned zb, const double2 *s_rows, const double aj0, const double aj1,
                                                        const double bj0, const double bj1, const double m0, const double m1,
                                                        double *s_row, double (&col)[2], const int lane, const bool younger,
                                                        double &nqy, double &nw) {
    constexpr int ZEV = RoundsSplit<NORMS>::ZEV, ZLN = RoundsSplit<NORMS>::ZLN, WRG = RoundsSplit<NORMS>::WRG;
    constexpr int NSTR = SLOTS - ZLN;
    const int half = lane >> 5;
    const double c = P.c, omc = P.omc;
    double colp0 = col[0], colp1 = col[1];
    unsigned zbv = zb;  // (not loop-invariant for the compiler: see tl_chunk_pass)
    asm volatile("" : "+v"(zbv));
    if (younger) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int gq = 0; gq < SLOTS / 4; ++gq) {
        double rowp[4];
        if (gq == (SLOTS / 4) * GL_TILE_PRIO_EIGHTHS / 8) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int s = 4 * gq + k;
            double2 gz;
            if (s % ZEV != ZEV - 1) {
                const int t = s - (s + 1) / ZEV;  // streamed row pairs before this one
                const int pos = t % RING;
                gz = zr[pos];
                if (WRAPR || t + RING < NSTR) zr[pos] = tl_ldz<ZNT>(P.zg, zbv + (unsigned)(RoundsSplit<NORMS>::slot((t + RING) % NSTR) * (2 * TB) * 8));
            } else {
                gz = zw[(s / ZEV) * 64];
            }
            const double2 rv = s_rows[2 * s + half];
            double2 wl = double2{0.0, 0.0};
            if (s >= WRG) wl = zw[(ZLN + s - WRG) * 64];
            double2 &wv = s < WRG ? wreg[s < WRG ? s : 0] : wl;
            double pp[2], t[2];
            tl_edge(c, wv.x, gz.x, rv.x, rv.y, aj0, bj0, pp[0], t[0]);
            tl_edge(c, wv.y, gz.y, rv.x, rv.y, aj1, bj1, pp[1], t[1]);
            if (MASKW) t[0] *= m0, t[1] *= m1;
            tl_new_w<NORMS>(c, omc, wv.x, t[0], nqy, nw);
            tl_new_w<NORMS>(c, omc, wv.y, t[1], nqy, nw);
            colp0 += pp[0];
            colp1 += pp[1];
            rowp[k] = pp[0] + pp[1];
            if (s >= WRG) zw[(ZLN + s - WRG) * 64] = wl;
            if (k % GL_TILE_PIN == GL_TILE_PIN - 1) {
                if (NORMS)
                    asm volatile("" : "+v"(nqy), "+v"(nw), "+v"(colp0), "+v"(colp1));
                else
                    asm volatile("" : "+v"(colp0), "+v"(colp1));
            }
        }
        const double rp = halfwave_rows4(rowp, lane);
        // the address of the row entries is formed anew every time: kept in a register over the pass it is the first thing
        // the compiler parks in scratch, and every reload waits for the whole z ring (s_waitcnt vmcnt(0))
        int lr = lane;
        asm volatile("" : "+v"(lr));
        if ((lr & 7) == 0) s_row[2 * (4 * gq + ((lr >> 3) & 3)) + (lr >> 5)] = rp;
    }
    col[0] = colp0, col[1] = colp1;
}

// column sums of a wave (each lane: its two columns over the rows of its half) -> the wave's column vector in LDS
__device__ __forceinline__ void tl_put_cols(const double (&col)[2], double *s_col, const int lane) {
    const int half = lane >> 5, c0 = 2 * (lane & 31);
    const double colp0 = add_xor32(col[0]), colp1 = add_xor32(col[1]);
    if (half == 0) {
        s_col[c0] = colp0;
        s_col[c0 + 1] = colp1;
    }
}

// One STREAMED tile (n > 4096: what the register files do not hold): w and 2 gamma z come from memory through two
// 4-deep register rings (row pair s + 4 is requested while row pair s is computed; the last four requests of a
// tile are the first row pairs of the wave's next streamed tile), the new w goes straight back.  Same per-edge
// arithmetic as the chunk pass; the new w is multiplied by its validity (i < j < n: diagonal tiles, ragged last block
// column), since it returns to memory, where every engine expects zeros outside the lattice.
#ifndef GL_TILE_STREAM_RING
#define GL_TILE_STREAM_RING 4
#endif
constexpr int RS = GL_TILE_STREAM_RING;  // 4 or 8
#ifndef GL_TILE_NT_W
#define GL_TILE_NT_W 0  // 1: w carries the hint too (measured slower: 137 against 114 us per iteration at n = 8192)
#endif
// NT: the streamed z passes through the caches with the non-temporal hint, so that it does not push the streamed w --
// rewritten and re-read every iteration, 201 MB at n = 8192 -- out of the 256 MB Infinity Cache.  us per iteration
// without / with: n = 5000: 32.8 / 38.2, 6144: 61.2 / 61.2, 7000: 88.9 / 82.7, 7800: 116.3 / 97.3, 8192: 141-143 / 114.0,
// 8704: 161 / 127.4.  (The hint on w as well: 8192: 137, 8704: 150.)
template <bool NT>
__device__ __forceinline__ double2 tl_ld_stream(const char *p) {
    if (NT) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(p));
        return double2{v.x, v.y};
    }
    return *reinterpret_cast<const double2 *>(p);
}
template <bool NT>
__device__ __forceinline__ void tl_st_stream(char *p, const double2 v) {
    if (NT) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store((v2d){v.x, v.y}, reinterpret_cast<v2d *>(p));
    } else {
        *reinterpret_cast<double2 *>(p) = v;
    }
}
#ifndef GL_TILE_STAGGER
#define GL_TILE_STAGGER 1
#endif
#ifndef GL_TILE_STREAM_SCHED
#define GL_TILE_STREAM_SCHED 1
#endif
template <bool NORMS, bool NT>
__device__ __forceinline__ void tl_stream_pass(const TileDev &P, double2 (&wq)[RS], double2 (&zq)[RS], const unsigned tb,
                                               const unsigned nxt, const double2 *s_rows, const double aj0, const double aj1,
                                               const double bj0, const double bj1, const bool diag, const bool jv0, const bool jv1,
                                               double *s_row, double (&col)[2], const int lane, double &nqy, double &nw,
                                               const int g0, const int ng = SLOTS / RS) {
    // (ng < SLOTS / RS: only the groups g0 .. g0 + ng - 1 of the tile -- a tile split between the waves of a workgroup)
    const int half = lane >> 5, c0 = 2 * (lane & 31);
    const double c = P.c, omc = P.omc;
    char *wbase = reinterpret_cast<char *>(P.w);
    const char *zbase = reinterpret_cast<const char *>(P.zg);
    double colp0 = col[0], colp1 = col[1];
#pragma unroll 1
    for (int gi = 0; gi < ng; ++gi) {
        // the waves of the chip start their tiles at different row pairs (g0): tiles are 32 KiB apart, and waves that walk
        // them in step would all be on the same few memory channels
        const int gr = (gi + g0) & (SLOTS / RS - 1);
        // what the ring requests next: the following group of this tile, at the end the first group of the next tile
        // (`nxt`; the caller passes this tile again when there is none: a harmless re-read keeps the loop free of branches)
        const unsigned nbase = gi < ng - 1 ? tb + (unsigned)((((gr + 1) & (SLOTS / RS - 1)) * RS) * (2 * TB) * 8)
                                           : nxt + (unsigned)((g0 * RS) * (2 * TB) * 8);
#pragma unroll
        for (int gh = 0; gh < RS / 4; ++gh) {
            const int gq = gr * (RS / 4) + gh;
            double rowp[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int s = 4 * gq + k, pos = 4 * gh + k;
                double2 wv = wq[pos];
                const double2 zv = zq[pos];
                const double2 rv = s_rows[2 * s + half];
                double pp[2], t[2];
                tl_edge(c, wv.x, zv.x, rv.x, rv.y, aj0, bj0, pp[0], t[0]);
                tl_edge(c, wv.y, zv.y, rv.x, rv.y, aj1, bj1, pp[1], t[1]);
                // (entries outside i < j < n hold w = 0 and zg = ZG_OUTSIDE: p is 0 there by itself, w' stays 0 by this factor)
                t[0] *= (jv0 && (!diag || 2 * s + half < c0)) ? 1.0 : 0.0;
                t[1] *= (jv1 && (!diag || 2 * s + half < c0 + 1)) ? 1.0 : 0.0;
                tl_new_w<NORMS>(c, omc, wv.x, t[0], nqy, nw);
                tl_new_w<NORMS>(c, omc, wv.y, t[1], nqy, nw);
                tl_st_stream<NT && GL_TILE_NT_W>(wbase + tb + (unsigned)(s * (2 * TB) * 8), wv);
                // the ring's registers are requested again only now that their values have been used (requested earlier, the
                // compiler copies the whole ring aside and waits for all of it at the top of the loop)
                asm volatile("" ::: "memory");
                wq[pos] = tl_ld_stream<NT && GL_TILE_NT_W>(wbase + nbase + (unsigned)(pos * (2 * TB) * 8));
                zq[pos] = tl_ld_stream<NT>(zbase + nbase + (unsigned)(pos * (2 * TB) * 8));
                colp0 += pp[0];
                colp1 += pp[1];
                rowp[k] = pp[0] + pp[1];
#if GL_TILE_STREAM_SCHED
                __builtin_amdgcn_sched_barrier(0);  // one row pair at a time: the registers belong to the resident tile
#endif
            }
            const double rp = halfwave_rows4(rowp, lane);
            if ((lane & 7) == 0) s_row[2 * (4 * gq + ((lane >> 3) & 3)) + half] = rp;
        }
    }
    col[0] = colp0, col[1] = colp1;
}

// NORMS: the norms of the stopping rule are formed every iteration (calls with tol > 0, and synchronous calls, which
// report the ratios of their last iteration); without, the pass is a fifth shorter (asynchronous fixed-iteration calls).
// One instantiation of the pass per kernel: two of them inside one loop cost 150 registers (measured).
//
// BIG (4096 < n <= ~11400, U = 32): the lattice no longer fits the register files.  A workgroup takes a rectangle of
// tiles, 8 block columns (one per wave) x up to 8 block rows, and works through it in ROUNDS: in round r wave v
// handles tile (I0 + r, J0 + v) -- the first one from its registers as before, the others streamed (tl_stream_pass);
// diagonal tiles are streamed tiles with the validity mask i < j.  A wave's tiles share their block column, so its
// column sums stay in registers over the rounds; its row vectors are folded after every round (two workgroup
// barriers) into the accumulators of the waves that publish the records.  Everything else -- records, owners, rule --
// is unchanged.  Streamed tiles are updated in place: a launch that fails after the rendezvous is an error as before,
// one whose workgroups never met has still written nothing.
// BATCH: a batch of graphs of one size as ONE block-diagonal lattice (TileDev), stopped by the screen of the coupled rule
template <int U, bool RAGGED, bool NORMS, int BIGV = 0, bool BATCH = false>  // BIGV: 0, 1 = rounds, 2 = rounds with non-temporal streams
__global__ __launch_bounds__(TT, 2) void k_tile(TileDev P) {
    constexpr bool BIG = BIGV != 0, NT = BIGV == 2;
    static_assert(!BATCH || (NORMS && !BIG), "batches: below the rounds, with the norms");
    constexpr bool MASKW = RAGGED && NORMS;  // (see tl_chunk_pass)
    constexpr int ZEV = RoundsSplit<NORMS>::ZEV, ZLN = RoundsSplit<NORMS>::ZLN, WRG = BIG ? RoundsSplit<NORMS>::WRG : WRG_NONE;
    static_assert(!BIG || U == SLOTS, "rounds work on whole tiles");
    // LDS-resident row pairs of z per chunk (rounds: a slice of RoundsSplit::SLICE row pairs shared by z and w)
    constexpr int ZRES = BIGV != 0 ? RoundsSplit<NORMS>::SLICE : (U == SLOTS ? SLOTS / 2 : U);
    constexpr int NS = U - ZRES;                      // streamed ones
    __shared__ double2 s_z[TW][ZRES][64];             // up to 128 KiB
    __shared__ double2 s_node[BIGV != 0 ? NBW : NBW_FIT][TB];  // {gamma d, gamma u} of the node blocks this workgroup touches
    __shared__ double s_part[TW][BIGV != 0 ? 4 : 2][TB];  // per chunk: row sums of p, column sums of p (owner phase: s_own); rounds: the same for a split extra tile
    __shared__ double s_dual[5][TB];                  // owner workgroups: d, ybar, pbar, d of the previous iteration, gamma S w
    __shared__ double s_bn[4];                        // ... and the norm partials of its last update
    __shared__ double s_gu;                           // ... and gamma sum_j u_j over all nodes (inputs of the current iteration)
    __shared__ double s_wn[TW][2];
    __shared__ double s_wcarry[TW];                   // per wave: sum of squares of its current w (the rule's denominator)
    __shared__ int s_flag[4];                         // {rendezvous ok, failed, stop, exact}
    __shared__ double s_scr[4];                       // a batch's look at the bottom of an iteration (wave RW's own: have, room, sure, figure)
    __shared__ double s_rel[2];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x;
    const int half = lane >> 5, c0 = 2 * (lane & 31);
    if (P.ctr[4] != 0) return;  // an earlier launch of this session failed and has not been looked at yet
    // development (GRAPHLEARN_AMD_RES_STAMPS): where a CALL spends its time -- entry, prologue issued, rendezvous, first
    // iteration, loop, write-back
    auto call_stamp = [&](int q) {  // (written at once: kept in registers the stamps would cost the pass a dozen of them)
        if (P.dbg && (threadIdx.x & 63) == 0)
            P.dbg[(size_t)P.NWG * 16 * (6 + 2 * TW) + ((size_t)blockIdx.x * TW + (threadIdx.x >> 6)) * 8 + q] = __builtin_amdgcn_s_memrealtime();
    };
    call_stamp(0);
    // ---- start-up rendezvous, arrival: every workgroup of the grid is on the chip, or nobody touches anything.  The
    //      arrival counter is monotonic over the launches of the session (target = arrivals so far + grid), so nothing has
    //      to be reset between launches: the launch is the only thing on the stream.  The arrival travels while the
    //      prologue below is issued; wave 0 waits for the others before it issues its own bulk loads (its poll would
    //      otherwise wait behind them: vector memory operations of a wave complete in order).
    if (tid == 0) __hip_atomic_fetch_add(P.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    const WaveDesc wdv = P.wd[wg * TW + wave];
    // the workgroup's descriptor: lane l keeps words l and 64 + l, wave-uniform reads go through v_readlane
    const int gdw0 = reinterpret_cast<const int *>(P.gd + wg)[lane], gdw1 = reinterpret_cast<const int *>(P.gd + wg)[64 + lane];
    auto gd_int = [&](int idx) { return idx < 64 ? __builtin_amdgcn_readlane(gdw0, idx) : __builtin_amdgcn_readlane(gdw1, idx - 64); };
    constexpr int GD_BLK = 4, GD_ENT = GD_BLK + NBW, GD_ROW = GD_ENT + NBW, GD_COL = GD_ROW + NBW;
    constexpr int GD_SLICE = GD_COL + NBW, GD_ACTIVE = GD_SLICE + 2, GD_NRND = GD_ACTIVE + 1, GD_ROW2 = GD_NRND + 1;
    constexpr int GD_NX = GD_ROW2 + NBW, GD_XT = GD_NX + 1, GD_XS = GD_XT + 4;
    const int nblk = gd_int(0), own = gd_int(1), own_e0 = gd_int(2), own_e1 = gd_int(3);
    // a node block's index inside its graph (what lies outside i < j < n is decided there)
    const int nb1 = BATCH ? P.nb1 : (int)P.nb;
    auto in_graph = [&](int b) { return BATCH ? b % nb1 : b; };
    const unsigned long long slice_mask = ((unsigned long long)(unsigned)gd_int(GD_SLICE + 1) << 32) | (unsigned)gd_int(GD_SLICE);
    const bool active = gd_int(GD_ACTIVE) != 0;
    unsigned *err = P.bar + 32;
    const auto r_part = __builtin_amdgcn_make_buffer_rsrc((void *)P.part, 0, (int)((int64_t)P.n_entries * TB * 16), 0x00020000);
    const auto r_nval = __builtin_amdgcn_make_buffer_rsrc((void *)P.nval, 0, (int)(P.nb * 2 * TB * 16), 0x00020000);
    const auto r_us = __builtin_amdgcn_make_buffer_rsrc((void *)P.usum, 0, (int)(2 * P.nb * 16), 0x00020000);
    const auto r_wn = __builtin_amdgcn_make_buffer_rsrc((void *)P.wnorm, 0, (int)(4 * P.NWG * 2 * 16), 0x00020000);
    const auto r_bn = __builtin_amdgcn_make_buffer_rsrc((void *)P.bnrm, 0, (int)(4 * P.nb * 4 * 16), 0x00020000);
    const auto r_lk = __builtin_amdgcn_make_buffer_rsrc((void *)P.look, 0, 4 * 4 * 16, 0x00020000);

    // ---- the node values this call starts from (what the previous call, or init, left in memory): requested first, so
    //      that waiting for them does not wait for the bulk loads behind them
    double nv_d[NACQ], nv_p[NACQ];
#pragma unroll
    for (int j = 0; j < NACQ; ++j) {
        const int k = wave + TW * j;
        nv_d[j] = nv_p[j] = 0.0;
        if (k < nblk) {
            const int64_t i = (int64_t)gd_int(GD_BLK + k) * TB + lane;
            nv_d[j] = P.dvar[i];
            nv_p[j] = P.pbar[i];
        }
    }
    // ---- dual state of my node block (owner workgroups, wave 0, lane = node): in LDS, not in registers that
    //      every wave of the grid would have to set aside.  gamma S w of the current iterate is ybar - d (:148); the sum of
    //      my block's gamma u is published for every owner (tag of this call's input state).
    const int64_t own_i = (int64_t)own * TB + lane;
    if (own >= 0 && wave == 0) {
        const double d0 = P.dvar[own_i], y0 = P.ybar[own_i], p0 = P.pbar[own_i];
        s_dual[0][lane] = d0;
        s_dual[1][lane] = y0;
        s_dual[2][lane] = p0;
        s_dual[3][lane] = P.dprev[own_i];
        s_dual[4][lane] = y0 - d0;
        if (lane < 4) s_bn[lane] = 0.0;
        const double us = wave_sum(P.gamma * (d0 - p0));
        if (lane == 0) __builtin_amdgcn_raw_buffer_store_b128(tl_gran(us, P.tag0), r_us, (int)(((P.tag0 & 1u) * (unsigned)P.nb + (unsigned)own) * 16u), 0, 16);
    }

    // ---- static work of this wave
    const bool has = wdv.tile >= 0;
    const int a0 = wdv.a0;
    unsigned zb = 0u;
    double2 wreg[U];
    double2 zr[RING];
#pragma unroll
    for (int k = 0; k < RING; ++k) zr[k] = double2{0.0, 0.0};
#pragma unroll
    for (int s = 0; s < U; ++s) wreg[s] = double2{0.0, 0.0};
    // fringe group: row pairs fa0 .. fa0 + fcnt - 1 of diagonal tile (fB, fB), w and 2 gamma z in registers
    const int fcnt = BIG ? 0 : wdv.fcnt, fa0 = wdv.fa0, fB = wdv.fB, fs = wdv.fs, ftgt = wdv.ftgt;
    double2 fw[FRMAX], fgz[FRMAX];
#pragma unroll
    for (int f = 0; f < FRMAX; ++f) fw[f] = fgz[f] = double2{0.0, 0.0};
    // element offsets of this lane's first pair in my chunk and in my fringe tile: formed where they are used (prologue and
    // write-back), from the lane number of that moment -- over the pass they would only hold registers
    auto chunk_off = [&](int ln) -> int64_t { return (int64_t)wdv.tile * GL_TILE_ELEMS + (int64_t)a0 * (2 * TB) + (ln >> 5) * TB + 2 * (ln & 31); };
    const int fBl = in_graph(fB >= 0 ? fB : 0);
    auto fringe_off = [&](int ln) -> int64_t {
        const int64_t t = BATCH ? (int64_t)((fB >= 0 ? fB : 0) / nb1) * P.ntiles1 + gl_tile_id(fBl, fBl, nb1) : gl_tile_id(fB, fB, P.nb);
        return t * GL_TILE_ELEMS + (ln >> 5) * TB + 2 * (ln & 31);
    };
    if (has) zb = (unsigned)(chunk_off(lane) * 8);
    // The bulk loads of the prologue, in the order the first pass wants them: the LDS-resident part of z (straight into
    // LDS: zg is stored pre-multiplied, so nothing has to pass through registers) -- waited for at once: the compiler
    // treats an LDS-DMA as a FLAT access that may touch both memories, and while one is pending every vmcnt wait of the wave
    // becomes vmcnt(0) --, then the head of the z ring, the fringe and w, which nothing waits for: the first pass consumes w
    // row pair by row pair as it arrives.
    auto bulk_loads = [&]() {
        const int64_t tb0 = chunk_off(lane), ftb = fringe_off(lane);
        if (has) {
#pragma unroll
            for (int s = 0; s < U; ++s) {  // LDS-resident part of 2 gamma z (U = 32: the odd row pairs; rounds: every ZEV-th)
                if (BIG ? s % ZEV != ZEV - 1 : (NS > 0 && (s & 1) == 0)) continue;
                tl_dma16(P.zg + tb0 + s * (2 * TB), &s_z[wave][BIG ? s / ZEV : (NS > 0 ? s / 2 : s)][0]);
            }
            if (BIG) {
#pragma unroll
                for (int s = WRG; s < U; ++s)  // (rounds: the last row pairs of w live in LDS)
                    tl_dma16(P.w + tb0 + s * (2 * TB), &s_z[wave][ZLN + s - WRG][0]);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): the DMAs have landed
        // The ring and w in ONE straight-line block, in the order of use, the fringe (used after the pass) behind them: the
        // compiler counts pending loads per static path, and with the ring in a block of its own there is a path on which
        // the ring loads are the youngest -- the first row pair would wait for everything (vmcnt(3)).  The empty asm
        // statements keep the scheduler from re-ordering the loads.
        if (has) {
            if (NS > 0 && !BIG) {
#pragma unroll
                for (int k = 0; k < RING; ++k) zr[k] = tl_ldz(P.zg, zb + (unsigned)(2 * k * (2 * TB) * 8));
                asm volatile("" ::: "memory");
            }
            if (BIG && !NORMS) {  // (the rounds' resident pass without the norms: its ring wraps too)
#pragma unroll
                for (int k = 0; k < RING; ++k) zr[k] = tl_ldz<NT>(P.zg, zb + (unsigned)(RoundsSplit<NORMS>::slot(k) * (2 * TB) * 8));
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int s = 0; s < (BIG ? WRG : U); ++s) {
                wreg[s] = *reinterpret_cast<const double2 *>(P.w + tb0 + s * (2 * TB));
                asm volatile("" ::: "memory");
            }
        }
#pragma unroll
        for (int f = 0; f < FRMAX; ++f)
            if (f < fcnt) {
                fw[f] = *reinterpret_cast<const double2 *>(P.w + ftb + (int64_t)(fa0 + f) * (2 * TB));
                fgz[f] = *reinterpret_cast<const double2 *>(P.zg + ftb + (int64_t)(fa0 + f) * (2 * TB));
            }
    };
    if (U < SLOTS) s_part[wave][0][lane] = 0.0;  // rows of the tile outside my chunk contribute nothing to my row vector
    // rounds (BIG): my tiles are (I + r, J), r < my_nr; round 0 is the resident one when `has`
    const int my_nr = BIG ? wdv.nr : 0, my_rs = wdv.rs;
    const int r_str = has ? 1 : 0;  // my first streamed round
    int lane_s = lane;  // the lane number as the streamed rounds see it (see below: not loop-invariant for the compiler)
    auto tile_off = [&](int r) -> unsigned {  // byte offset of this lane's first pair of tile (I + r, J)
        return (unsigned)((gl_tile_id(wdv.I + r, wdv.J, P.nb) * GL_TILE_ELEMS + (lane_s >> 5) * TB + 2 * (lane_s & 31)) * 8);
    };
    const int Jl = in_graph(wdv.J);
    const bool jv0 = (int64_t)Jl * TB + c0 < P.n, jv1 = (int64_t)Jl * TB + c0 + 1 < P.n;
    // where this wave starts its streamed tiles
    const int g0 = GL_TILE_STAGGER ? (int)(((unsigned)(wg * TW + wave) * 5u) & (unsigned)(SLOTS / RS - 1)) : 0;
    auto ring_load = [&](double2 (&wq)[RS], double2 (&zq)[RS], unsigned base, int grp) {
#pragma unroll
        for (int k = 0; k < RS; ++k) {
            wq[k] = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(P.w) + base + (unsigned)((grp * RS + k) * (2 * TB) * 8));
            zq[k] = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(P.zg) + base + (unsigned)((grp * RS + k) * (2 * TB) * 8));
        }
    };
    const int nx = BIG ? gd_int(GD_NX) : 0;  // split extra tiles of this workgroup
    auto xtile_off = [&](int k) -> unsigned {  // byte offset of this lane's first pair of extra tile k
        return (unsigned)(((int64_t)gd_int(GD_XT + k) * GL_TILE_ELEMS + (lane_s >> 5) * TB + 2 * (lane_s & 31)) * 8);
    };
    // ragged last block column: 0 / 1 factors of my two columns (only the variants with the norms keep them over the pass)
    auto col_valid = [&](int e) { return (RAGGED && has) ? ((int64_t)Jl * TB + c0 + e < P.n ? 1.0 : 0.0) : 1.0; };
    const double m0 = MASKW ? col_valid(0) : 1.0, m1 = MASKW ? col_valid(1) : 1.0;

    // ---- start-up rendezvous, wait (wave 0; the other waves go on and issue their bulk loads meanwhile)
    // Below the rounds nothing of the session's state is written before the end of the call, and the first thing a
    // workgroup needs from another one is a record of iteration 0: a launch that is not co-resident fails there (bounded
    // poll) exactly as it would fail here, with everything untouched.  So the wait itself is only taken by the rounds
    // (which update their streamed tiles in place from the first pass on) and under the test hook -- a poll issued now
    // would sit in this CU's memory queue behind the bulk loads of the other seven waves (measured: 12 us).
    const bool wait_here = BIG || P.check_rendezvous != 0;
    if (tid == 0) {
        const GlGridBar gb{P.bar, 0};
        s_flag[0] = (!wait_here || gl_spin_until(gb, P.bar, P.rendezvous)) ? 1 : 0;
        s_flag[1] = 0;
        s_flag[2] = 0;
        s_flag[3] = 0;
        s_scr[0] = 0.0, s_scr[1] = 1.0, s_scr[2] = 1.0, s_scr[3] = INFINITY;  // (a barrier stands between here and their first use)
    }
    // (Loads and stores of a wave complete out of order with respect to each other on gfx9, so with both kinds pending the
    //  compiler can only wait for vmcnt(0).  The prologue's small stores and loads are therefore drained here, and every
    //  iteration ends drained (bottom of the loop): the bulk loads below are then the only pending operations when the first
    //  pass starts, and it waits for w row pair by row pair -- vmcnt(31), vmcnt(30), ... -- instead of for all of it.)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    bulk_loads();
    // the node values of iteration 0 (requested before the bulk loads: this does not wait for those)
#pragma unroll
    for (int j = 0; j < NACQ; ++j) {
        const int k = wave + TW * j;
        if (k < nblk) s_node[k][lane] = double2{P.gamma * nv_d[j], P.gamma * (nv_d[j] - nv_p[j])};
    }
    tl_barrier();
    const bool met = s_flag[0] != 0;  // (no stamp here: a store between the bulk loads and the first pass would cost the pass its counted waits)
    if (met && !active) return;  // nothing to do here: no edges, no node block (nobody waits for this workgroup)
    if (NORMS) {  // sum of squares of the iterate this call starts from (waits for all of it: converge-mode calls are long)
        double acc = 0;
        if (BIG) {
#pragma unroll 1
            for (int r = r_str; r < my_nr; ++r) {
                const char *b = reinterpret_cast<const char *>(P.w) + tile_off(r);
#pragma unroll 4
                for (int s2 = 0; s2 < SLOTS; ++s2) {
                    const double2 v = *reinterpret_cast<const double2 *>(b + s2 * (2 * TB) * 8);
                    acc = __builtin_fma(v.y, v.y, __builtin_fma(v.x, v.x, acc));
                }
            }
        }
        if (BIG) {
#pragma unroll 1
            for (int k = 0; k < nx; ++k) {  // my four row pairs of every split extra tile
                const char *b = reinterpret_cast<const char *>(P.w) + xtile_off(k) + (unsigned)(wave * RS * (2 * TB) * 8);
#pragma unroll
                for (int s2 = 0; s2 < RS; ++s2) {
                    const double2 v = *reinterpret_cast<const double2 *>(b + s2 * (2 * TB) * 8);
                    acc = __builtin_fma(v.y, v.y, __builtin_fma(v.x, v.x, acc));
                }
            }
        }
#pragma unroll
        for (int s = 0; s < U; ++s) {
            const double2 wv = (BIG && s >= WRG) ? (has ? s_z[wave][ZLN + (s >= WRG ? s - WRG : 0)][lane] : double2{0.0, 0.0}) : wreg[s];
            acc = __builtin_fma(wv.y, wv.y, __builtin_fma(wv.x, wv.x, acc));
        }
#pragma unroll
        for (int f = 0; f < FRMAX; ++f) acc = __builtin_fma(fw[f].y, fw[f].y, __builtin_fma(fw[f].x, fw[f].x, acc));
        acc = wave_sum(acc);
        if (lane == 0) s_wcarry[wave] = acc;
    }
    bool failed = !met;
    int done_iters = 0, stopped = 0, fail_it = 0;
    double rel_p = INFINITY, rel_d = INFINITY;
    const bool rule_on = P.tol > 0;
    const bool gate_on = !NORMS && !BATCH && P.gate > 0 && P.tol > 0;  // (see TileDev::gate)
    constexpr int QN0 = NORMS ? 0 : 2;  // the norm records there are: all four, or the dual pair only (the gate)
    bool evaluated = false, exact = false;

    // (wave TW - 1) one norm record per node block, lane = block
    auto norms_of = [&](int j, double (&b)[4]) -> bool {  // sums of the records of local iteration j
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        const unsigned tag = P.tag0 + (unsigned)j + 1u;
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = 0.0;
#pragma unroll 1
        for (int b0 = 0; b0 < ((BIG || BATCH) ? (int)P.nb : 1); b0 += 64) {  // (nb <= 64 unless BIG or a batch)
            const bool mine = b0 + lane < P.nb;
            tl_u4 r[4];
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int q = QN0; q < 4; ++q) {
                    r[q] = tl_u4{0u, tag, 0u, tag};
                    if (mine)
                        r[q] = __builtin_amdgcn_raw_buffer_load_b128(r_bn, (((j & 3) * (int)P.nb + b0 + lane) * 4 + q) * 16, 0, 16);
                    ok = ok && tl_ok(r[q], tag);
                }
                asm volatile("" ::: "memory");
                if (__all(ok)) break;
                if (!tl_spin(spins, t0, err)) return false;
            }
#pragma unroll
            for (int q = QN0; q < 4; ++q) b[q] += wave_sum(mine ? tl_val(r[q]) : 0.0);
        }
        return true;
    };
    // BATCH: the screen of the coupled rule (smooth_learning.py:158-159,166-167 on (E x B) and (n x B) matrices: spectral
    // norms; lambda_max of a Gram matrix lies in [max diagonal, trace]):  s = {max_g ||dW_g||^2, sum_g ||W_g||^2,
    // max_g ||dD_g||^2, sum_g ||D_g||^2} of local iteration j from the node blocks' norm records -- lane = block, whole graphs
    // per sweep of the wave, every sum in a fixed order.
    // (pre / with_pre: the first sweep's records as requested earlier by screen_request)
    auto screen_request = [&](int j, tl_u4 (&r)[4]) __attribute__((always_inline)) {
        const unsigned tag = P.tag0 + (unsigned)j + 1u;
        int ls = lane;
        asm volatile("" : "+v"(ls));
        const bool mine = ls < (64 / nb1) * nb1 && ls < P.nb;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            r[q] = tl_u4{0u, tag, 0u, tag};
            if (mine) r[q] = __builtin_amdgcn_raw_buffer_load_b128(r_bn, (((j & 3) * (int)P.nb + ls) * 4 + q) * 16, 0, 16);
        }
    };
    auto screen_of = [&](int j, double (&s)[4], const tl_u4 (&pre)[4], const bool with_pre) __attribute__((always_inline)) -> bool {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        const unsigned tag = P.tag0 + (unsigned)j + 1u;
        int ls = lane;
        asm volatile("" : "+v"(ls));  // (what derives from the lane number is formed here and now: hoisted out of the iteration loop, once per copy of this, it lives across the pass -- in scratch)
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = 0.0;
        const int per = (64 / nb1) * nb1, first = ls - ls % nb1;  // lanes in use per sweep; my graph's first lane
#pragma unroll 1
        for (int b0 = 0; b0 < (int)P.nb; b0 += per) {
            const bool mine = ls < per && b0 + ls < P.nb;
            tl_u4 r[4];
            bool use_pre = with_pre && b0 == 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    r[q] = tl_u4{0u, tag, 0u, tag};
                    if (use_pre) r[q] = pre[q];
                    else if (mine) r[q] = __builtin_amdgcn_raw_buffer_load_b128(r_bn, (((j & 3) * (int)P.nb + b0 + ls) * 4 + q) * 16, 0, 16);
                    ok = ok && tl_ok(r[q], tag);
                }
                use_pre = false;

                asm volatile("" ::: "memory");
                if (__all(ok)) break;
                if (!tl_spin(spins, t0, err)) return false;
            }
            const double v0 = mine ? tl_val(r[0]) : 0.0, v2 = mine ? tl_val(r[2]) : 0.0;
            s[1] += wave_sum(mine ? tl_val(r[1]) : 0.0);
            s[3] += wave_sum(mine ? tl_val(r[3]) : 0.0);
            // per-graph sums over the graph's nb1 lanes, then the maximum over the graphs: on the VALU (DPP) where a graph's
            // lanes are an aligned group of 2, 4, 8 or 16 -- through ds_bpermute the two took ~1 us of dependent LDS round trips
            double a0 = 0.0, a2 = 0.0;
            if (nb1 == 1) {
                a0 = v0, a2 = v2;
            } else if (nb1 == 2 || nb1 == 4 || nb1 == 8 || nb1 == 16) {
                a0 = v0, a2 = v2;
                a0 += dpp_mov<DPP_QUAD_XOR1>(a0), a2 += dpp_mov<DPP_QUAD_XOR1>(a2);
                if (nb1 >= 4) a0 += dpp_mov<DPP_QUAD_XOR2>(a0), a2 += dpp_mov<DPP_QUAD_XOR2>(a2);
                if (nb1 >= 8) a0 += dpp_mov<DPP_HALF_MIRROR>(a0), a2 += dpp_mov<DPP_HALF_MIRROR>(a2);
                if (nb1 >= 16) a0 += dpp_mov<DPP_ROW_ROR8>(a0), a2 += dpp_mov<DPP_ROW_ROR8>(a2);
            } else {
#pragma unroll 1
                for (int k = 0; k < nb1; ++k) {
                    a0 += __shfl(v0, first + k);
                    a2 += __shfl(v2, first + k);
                }
            }
            s[0] = fmax(s[0], wave_max(mine ? a0 : 0.0));
            s[2] = fmax(s[2], wave_max(mine ? a2 : 0.0));
        }
        return true;
    };
    // Round 4: the look that keeps a batch's cruise going costs no workgroup anything at the top of an iteration.  It used to
    // read the records of iteration it - 1 there -- they arrive behind the node values: every workgroup sat 1.6 us per
    // iteration (records' round trip + ~1 us of cross-lane arithmetic) between its node values and its pass.  Now wave RW
    // judges iteration it - 1 at the BOTTOM of iteration it -- its records have been out for a whole pass, and the wave
    // would only wait: for node values (consumers), for wave 0's update (owners) --, extrapolates TWO iterations ahead, and the
    // next top only reads what it found.  The fresh records are read once, where the cruise ends; what is reported then
    // covers both iterations the look had not judged.
    constexpr int RW = BATCH ? TW - 2 : TW - 1;  // the wave that evaluates the rule (a batch: not the slice's wave)
    // (its findings wait in LDS -- written and read by wave RW alone, in program order -- not in registers across the pass)
    auto screen_bottom = [&](const int it, const tl_u4 (&pre)[4], const bool with_pre) __attribute__((always_inline)) {  // (wave RW of a batch with the rule on)
        double have = 0.0, room = 1.0, sr = 1.0, fig = s_scr[3];
        if (it >= 1 && P.it0 + it - 1 >= 1) {  // never at iteration 0 (:156-161)
            double b[4];
            if (screen_of(it - 1, b, pre, with_pre)) {
                const double rk = tl_screen_ahead(b, P.t2lo);
                const double f = (fig > rk && fig < INFINITY) ? rk / fig : 1.0;
                room = (rk > 1.05 * 1.05 && rk * f * f > 1.05 * 1.05) ? 1.0 : 0.0;
                sr = tl_screen_sure(b, P.t2lo) ? 1.0 : 0.0;
                fig = rk;
                have = 1.0;
            } else if (lane == 0) {
                s_flag[1] = 1;  // (seen at the next top)
            }
        }
        if (lane == 0) s_scr[0] = have, s_scr[1] = room, s_scr[2] = sr, s_scr[3] = fig;
    };


    // The iterate's way back to memory.  At the end of the call -- or, in a call whose last iteration is known in advance
    // (tol = 0, below the rounds, no norms: what bench.py times), right after the last pass: the stores then drain underneath
    // the last hand-off instead of after it (3 of the ~33 us a call costs beyond its iterations).
    bool w_stored = false;
    auto store_w = [&]() __attribute__((always_inline)) {
        int lw = lane;
        asm volatile("" : "+v"(lw));
        if (has) {
            const int64_t tb0 = chunk_off(lw);
#pragma unroll
            for (int s = 0; s < U; ++s) {
                double2 v = (BIG && s >= WRG) ? s_z[wave][ZLN + (s >= WRG ? s - WRG : 0)][lane] : wreg[s];
                if (RAGGED && !MASKW) v = double2{((int64_t)Jl * TB + 2 * (lw & 31) < P.n ? 1.0 : 0.0) * v.x, ((int64_t)Jl * TB + 2 * (lw & 31) + 1 < P.n ? 1.0 : 0.0) * v.y};
                *reinterpret_cast<double2 *>(P.wout + tb0 + s * (2 * TB)) = v;
            }
        }
#pragma unroll
        for (int f = 0; f < FRMAX; ++f)
            if (f < fcnt) {
                const int64_t i = (int64_t)fBl * TB + 2 * (fa0 + f) + (lw >> 5), j = (int64_t)fBl * TB + 2 * (lw & 31);
                *reinterpret_cast<double2 *>(P.wout + fringe_off(lw) + (int64_t)(fa0 + f) * (2 * TB)) =
                    double2{(i < j && j < P.n) ? fw[f].x : 0.0, (i < j + 1 && j + 1 < P.n) ? fw[f].y : 0.0};
            }
        w_stored = true;
    };
    // One iteration.  FIRST (iteration 0 of the call) is an instantiation of its own, in front of the loop: it takes its node
    // values from the prologue instead of polling for them, and -- the point of the copy -- nothing but the bulk loads of the
    // prologue is pending when its pass starts, so the pass waits for w row pair by row pair (vmcnt(31), vmcnt(30), ...) and
    // runs underneath the loads.  Inside the loop the same pass sees, on some path, a store or a poll in front of it, and
    // with loads and stores both pending a wave can only wait for all of them (gfx9 returns them out of order with respect
    // to each other).  false: the call ends here (stopped, failed, or all iterations done).
    auto iteration = [&](auto first_c, const int it) __attribute__((always_inline)) -> bool {
        constexpr bool FIRST = decltype(first_c)::value;
        const unsigned tag_in = P.tag0 + (unsigned)it;        // node values that are the input of iteration `it`
        const unsigned tag_out = P.tag0 + (unsigned)it + 1u;  // what iteration `it` produces
        const bool last_pass = it == P.maxit;
        // does this workgroup evaluate the rule of iteration it - 1 now?
        const bool want_rule = !FIRST && NORMS && it >= 1 && (last_pass ? wg == 0 : rule_on);
        if (last_pass && !want_rule) return false;
        const bool stamp = P.dbg != nullptr && it >= 8 && it < 24 && !last_pass;
        unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
        unsigned long long tx[4] = {0, 0, 0, 0};  // per wave: node values seen, records stored, (owners) records seen, node values stored
        if (stamp) ts[0] = __builtin_amdgcn_s_memrealtime();

        // ================= acquire: {gamma d, gamma u} of my node blocks, and the norm records of iteration it - 1 =================
        {
            bool good = true;
            if (!last_pass) {
                if constexpr (FIRST) {  // (the state the previous call, or init, left in memory: in s_node since the prologue)
                } else if (wave < nblk) {
                    // (two probes in flight, half a round trip apart, measured slower: their registers push the pass into
                    //  scratch, and most first probes succeed anyway)
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    unsigned spins = 0;
                    tl_u4 g[NACQ][2];
                    auto probe = [&](tl_u4 (&q)[NACQ][2]) {
#pragma unroll
                        for (int j = 0; j < NACQ; ++j) {
                            const int k = wave + TW * j;
                            q[j][0] = q[j][1] = tl_u4{0u, tag_in, 0u, tag_in};
                            if (k < nblk) {
                                const int off = (gd_int(GD_BLK + k) * 2 * TB + lane) * 16;
                                q[j][0] = __builtin_amdgcn_raw_buffer_load_b128(r_nval, off, 0, 16);
                                q[j][1] = __builtin_amdgcn_raw_buffer_load_b128(r_nval, off + TB * 16, 0, 16);
                            }
                        }
                        asm volatile("" ::: "memory");
                    };
                    auto valid = [&](const tl_u4 (&q)[NACQ][2]) {
                        bool ok = true;
#pragma unroll
                        for (int j = 0; j < NACQ; ++j) ok = ok && tl_ok(q[j][0], tag_in) && tl_ok(q[j][1], tag_in);
                        return __all(ok);
                    };
#if GL_TILE_PROBES == 2
                    // two probes in flight, half a round trip apart: what a probe finds is decided when it reaches memory, so a
                    // record that lands just behind one is picked up half a round trip later instead of a whole one
                    tl_u4 g2[NACQ][2];
                    probe(g);
                    __builtin_amdgcn_s_sleep(GL_TILE_PROBE_GAP);
                    probe(g2);
                    for (;;) {
                        if (valid(g)) break;  // (waits for the older probe only: the younger one stays in flight)
                        probe(g);
                        if (valid(g2)) {
#pragma unroll
                            for (int j = 0; j < NACQ; ++j) g[j][0] = g2[j][0], g[j][1] = g2[j][1];
                            break;
                        }
                        probe(g2);
                        if (!tl_spin(spins, t0, err)) {
                            good = false;
                            break;
                        }
                    }
#else
                    for (;;) {
                        probe(g);
                        if (valid(g)) break;
                        if (!tl_spin(spins, t0, err)) {
                            good = false;
                            break;
                        }
                    }
#endif
                    if (stamp) tx[0] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
                    for (int j = 0; j < NACQ; ++j) {
                        const int k = wave + TW * j;
                        if (k < nblk) s_node[k][lane] = double2{tl_val(g[j][0]), tl_val(g[j][1])};
                    }
                }
            }
            if (want_rule && wave == RW) {
                // The rule of iteration it - 1 needs the norm partials of EVERY owner, i.e. the slowest workgroup of the
                // chip, every iteration (measured: 1.8 us of 17).  So while the ratios are clearly above tol only the
                // records of iteration it - 2 are looked at -- published a whole iteration ago, no waiting -- and the exact
                // evaluation (records of it - 1, as the reference does it) starts once those ratios come within 25 % of tol:
                // from then on every iteration is judged exactly, this one included.  Ratios move by ~1/k per iteration,
                // so nothing is skipped; a run whose ratios fall by more than 20 % in ONE iteration straight below tol
                // stops one iteration late (inside the +-1 contract of SURVEY.md section 7).
                // The sums of the older records were formed at the END of the previous iteration (below: while this
                // workgroup would only wait for node values) and wait in LDS: the look costs the acquire nothing.
                double b[4] = {0, 0, 0, 0};
                int stop = 0;
                double rp = INFINITY, rd = INFINITY;
                if constexpr (BATCH) {
                    // The batch runs while the screen RULES OUT the coupled rule with 5 % to spare (the lower bound of one of
                    // the two ratios above 1.05 tol, now and extrapolated one iteration ahead) and ends at the first iteration where it does not: the exact spectral norms
                    // are the host's business from there on.  Should the lower bounds of that very iteration already lie below
                    // tol (a drop of 5 % in one iteration, or a start from a converged iterate) the rule may have held for
                    // an iterate these registers no longer hold: reported (rp = 1), and the caller redoes the run its own way.
                    // (round 4: the look itself sits at the bottom of the previous iteration -- screen_bottom)
                    const bool scr_have = s_scr[0] != 0.0;
                    const bool lag_sure = !scr_have || s_scr[2] != 0.0, lag_room = !scr_have || s_scr[1] != 0.0;
                    const bool end_now = last_pass || !lag_room;
                    if (end_now && good && P.it0 + it - 1 >= 1) {
                        tl_u4 none[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) none[q] = tl_u4{0u, 0u, 0u, 0u};
                        good = good && screen_of(it - 1, b, none, false);
                        if (good) {
                            const double rk = tl_screen_ahead(b, P.t2lo);
                            // (the budget ran out with room to spare in both iterations: not a stop)
                            stop = (last_pass && lag_room && rk > 1.05 * 1.05) ? 0 : 1;
                            rp = (lag_sure && tl_screen_sure(b, P.t2lo)) ? 0.0 : 1.0;
                            rd = rk * P.t2lo;
                        }
                    }
                    if (lane == 0) {
                        s_rel[0] = rp;
                        s_rel[1] = rd;
                        s_flag[2] = stop;
                        s_flag[3] = 3;
                    }
                } else {
                // The first iterations of every call are judged exactly too: a call that starts from a converged or nearly
                // converged iterate (edge_init, a resumed session) drops below tol at once, with no older record to warn of it
                bool sticky = exact;
                bool now_exact = exact || last_pass || it <= GL_TILE_EXACT_HEAD;
                if (!now_exact && it >= 2) {
                    // what the early look of iteration it - 1 found (one workgroup takes it for all, below; its record has been
                    // on its way for a whole hand-off): -1 failed, 1 below tol, 2 within 25 % of it
                    const unsigned tag = P.tag0 + (unsigned)it;
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    unsigned spins = 0;
                    tl_u4 r;
                    for (;;) {
                        r = tl_u4{0u, tag, 0u, tag};
                        if (lane < 3) r = __builtin_amdgcn_raw_buffer_load_b128(r_lk, ((((it - 1) & 3) * 4) + lane) * 16, 0, 16);
                        asm volatile("" ::: "memory");
                        if (__all(tl_ok(r, tag))) break;
                        if (!tl_spin(spins, t0, err)) {
                            good = false;
                            break;
                        }
                    }
                    const double mine = tl_val(r);
                    const int look = good ? (int)lane_bcast(mine, 0) : -1;
                    good = good && look >= 0;
                    if (look == 1) stop = 1, rp = lane_bcast(mine, 1), rd = lane_bcast(mine, 2);  // (a jump straight below tol: seen one iteration late)
                    if (look == 2) now_exact = sticky = true;
                }
                if (now_exact && !stop) {
                    good = good && norms_of(it - 1, b);
                    const long long itp = P.it0 + it - 1;  // the iteration these norms belong to
                    if (itp >= 1) {  // never at iteration 0 (:156-161)
                        rp = sqrt(b[0]) / sqrt(b[1]);
                        rd = sqrt(b[2]) / sqrt(b[3]);
                    }
                    stop = (itp >= 1 && rp < P.tol && rd < P.tol) ? 1 : 0;
                    if (itp >= 1 && rp < 1.25 * P.tol && rd < 1.25 * P.tol) sticky = true;  // (the screen's threshold)
                }
                if (lane == 0) {
                    s_rel[0] = rp;
                    s_rel[1] = rd;
                    s_flag[2] = stop;
                    s_flag[3] = (sticky ? 1 : 0) | (now_exact ? 2 : 0);
                }
                }
            }
            if (!NORMS && gate_on && !FIRST && !last_pass && it >= 2 && wave == TW - 1) {
                // the gate: what the look at the dual norms of iteration it - 2 found (one workgroup takes it for all, below; its
                // record has been on its way for a whole hand-off): 2 = the dual ratio has come within gate x tol
                const unsigned tag = P.tag0 + (unsigned)it;
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                unsigned spins = 0;
                tl_u4 r;
                for (;;) {
                    r = tl_u4{0u, tag, 0u, tag};
                    if (lane < 3) r = __builtin_amdgcn_raw_buffer_load_b128(r_lk, ((((it - 1) & 3) * 4) + lane) * 16, 0, 16);
                    asm volatile("" ::: "memory");
                    if (__all(tl_ok(r, tag))) break;
                    if (!tl_spin(spins, t0, err)) {
                        good = false;
                        break;
                    }
                }
                const int look = good ? (int)lane_bcast(tl_val(r), 0) : -1;
                good = good && look >= 0;
                if (lane == 0) s_flag[2] = look == 2 ? 2 : 0;
            }
            if (!good) s_flag[1] = 1;
        }
        tl_barrier();
        if (s_flag[1]) {
            failed = true;
            fail_it = it;
            return false;
        }
        if (!NORMS && gate_on && s_flag[2] == 2) {  // the same record, the same iteration, in every workgroup
            stopped = 2;
            return false;
        }
        if (want_rule) {
            exact = (s_flag[3] & 1) != 0;
            if ((s_flag[3] & 2) || s_flag[2]) {  // (a look at older records only reports when it stops the run)
                rel_p = s_rel[0];
                rel_d = s_rel[1];
                evaluated = true;
            }
            if (s_flag[2]) stopped = 1;
        }
        if (stopped || last_pass) return false;  // the rule of iteration it - 1 holds: its iterate is the result

        // ================= edge phase =================
        if (stamp) ts[1] = __builtin_amdgcn_s_memrealtime();
        double nqy = 0, nw = 0;
        double accp[NACQ];  // BIG: what the rounds before the last one add to the records I publish
#pragma unroll
        for (int j = 0; j < NACQ; ++j) accp[j] = 0.0;
        if constexpr (!BIG) {
            if (has) {
                const double2 cv0 = s_node[wdv.sj][c0], cv1 = s_node[wdv.sj][c0 + 1];
                double col[2] = {0.0, 0.0};
                tl_chunk_pass<U, MASKW, NORMS>(P, wreg, zr, &s_z[wave][0][lane], zb, &s_node[wdv.si][2 * a0], cv0.x, cv1.x, cv0.y,
                                               cv1.y, m0, m1, &s_part[wave][0][2 * a0], col, lane, wave >= TW / 2, nqy, nw);
                tl_put_cols(col, &s_part[wave][1][0], lane);
            }
        } else {
            // Round 0 stands apart from the loop over the later rounds so that the rings of the streamed pass are not live
            // while the resident pass runs (its 256 registers are spoken for).
            const int nrw = gd_int(GD_NRND);
            double col[2] = {0.0, 0.0};
            double2 cv0 = double2{0.0, 0.0}, cv1 = cv0;
            if (my_nr > 0) cv0 = s_node[wdv.sj][c0], cv1 = s_node[wdv.sj][c0 + 1];
            auto fold = [&](int r) {  // row vectors of round r -> the records I publish
#pragma unroll
                for (int j = 0; j < NACQ; ++j) {
                    const int k = wave + TW * j;
                    if (k < nblk) {
                        const int rm = ((r < 4 ? gd_int(GD_ROW + k) : gd_int(GD_ROW2 + k)) >> (8 * (r & 3))) & 255;
#pragma unroll
                        for (int cs2 = 0; cs2 < TW; ++cs2)
                            if ((rm >> cs2) & 1) accp[j] += s_part[cs2][0][lane];
                    }
                }
            };
            if (my_nr > 0) {
                const double2 *rows = &s_node[my_rs & 15][0];
                if (has) {
                    if (NORMS) {  // (without the norms the ring wraps: it was filled by the previous pass, or by the prologue)
                        unsigned zb0 = zb;  // (not loop-invariant for the compiler: it would keep the ring's 64-bit addresses over the iteration, in scratch)
                        asm volatile("" : "+v"(zb0));
#pragma unroll
                        for (int k = 0; k < RING; ++k) zr[k] = tl_ldz<NT>(P.zg, zb0 + (unsigned)(RoundsSplit<NORMS>::slot(k) * (2 * TB) * 8));
                    }
                    tl_resident_pass_rounds<MASKW, NORMS, NT, !NORMS>(P, wreg, zr, &s_z[wave][0][lane], zb, rows, cv0.x, cv1.x, cv0.y, cv1.y, m0, m1,
                                                              &s_part[wave][0][0], col, lane, wave >= TW / 2, nqy, nw);
                } else {  // (a column whose first tile here is its diagonal tile)
                    asm volatile("" : "+v"(lane_s));  // (as below)
                    double2 wq[RS], zq[RS];
                    ring_load(wq, zq, tile_off(0), g0);
                    tl_stream_pass<NORMS, NT>(P, wq, zq, tile_off(0), tile_off(0), rows, cv0.x, cv1.x, cv0.y, cv1.y, wdv.I == wdv.J, jv0, jv1,
                                              &s_part[wave][0][0], col, lane_s, nqy, nw, g0);
                }
            }
            if (nrw > 1) {
                tl_barrier();
                fold(0);
                tl_barrier();
                // what the streamed rounds derive from the lane number (offsets, masks) is recomputed here every iteration:
                // hoisted out of the iteration loop it would live through the resident pass, which has no register to spare
                asm volatile("" : "+v"(lane_s));
                double2 wq[RS], zq[RS];
#pragma unroll
                for (int k = 0; k < RS; ++k) wq[k] = zq[k] = double2{0.0, 0.0};
                if (my_nr > 1) ring_load(wq, zq, tile_off(1), g0);
#pragma unroll 1
                for (int r = 1; r < nrw; ++r) {
                    if (r < my_nr) {
                        const unsigned tb = tile_off(r), nxt = r + 1 < my_nr ? tile_off(r + 1) : tb;
                        const double2 *rows = &s_node[(my_rs >> (4 * r)) & 15][0];
                        tl_stream_pass<NORMS, NT>(P, wq, zq, tb, nxt, rows, cv0.x, cv1.x, cv0.y, cv1.y, wdv.I + r == wdv.J, jv0, jv1,
                                                  &s_part[wave][0][0], col, lane_s, nqy, nw, g0);
                    }
                    if (r == nrw - 1) break;  // (the barrier and the last fold are those of the publish phase below)
                    tl_barrier();
                    fold(r);
                    tl_barrier();
                }
            }
            if (my_nr > 0) tl_put_cols(col, &s_part[wave][1][0], lane);
            // Split extra tiles: every wave streams four row pairs (group `wave`) of the same tile -- one ring fill per wave
            // instead of eight ring turns of one wave, and a CU's share of the stream stays at one tile.  Their row and
            // column vectors go to s_part[.][3] and [2] and are folded into the records' accumulators at once.
            if (nx > 0) {
            asm volatile("" : "+v"(lane_s));
            double2 xwq[RS], xzq[RS];
            ring_load(xwq, xzq, xtile_off(0), wave);
#pragma unroll 1
            for (int k = 0; k < nx; ++k) {
                const int xs = gd_int(GD_XS + k), xsi = xs & 255, xsj = (xs >> 8) & 255;
                const int xJ = gd_int(GD_BLK + xsj);
                const unsigned tb = xtile_off(k), nxt = k + 1 < nx ? xtile_off(k + 1) : tb;  // (the ring's last requests: the next extra's rows)
                const int cl = 2 * (lane_s & 31);
                const double2 xv0 = s_node[xsj][cl], xv1 = s_node[xsj][cl + 1];
                double colx[2] = {0.0, 0.0};
                s_part[wave][3][lane] = 0.0;  // (my group fills its own 8 rows)
                tl_stream_pass<NORMS, NT>(P, xwq, xzq, tb, nxt, &s_node[xsi][0], xv0.x, xv1.x, xv0.y, xv1.y, ((xs >> 16) & 1) != 0,
                                          (int64_t)xJ * TB + cl < P.n, (int64_t)xJ * TB + cl + 1 < P.n, &s_part[wave][3][0], colx, lane_s, nqy, nw,
                                          wave, 1);
                tl_put_cols(colx, &s_part[wave][2][0], lane_s);
                tl_barrier();
#pragma unroll
                for (int j = 0; j < NACQ; ++j) {
                    const int kb = wave + TW * j;
                    if (kb < nblk && (kb == xsi || kb == xsj)) {
#pragma unroll
                        for (int cs2 = 0; cs2 < TW; ++cs2) {
                            if (kb == xsj) accp[j] += s_part[cs2][2][lane];
                            if (kb == xsi) accp[j] += s_part[cs2][3][lane];
                        }
                    }
                }
                tl_barrier();
            }
            }
        }
        if (fcnt > 0) {  // my fringe row pairs (diagonal tile: entries i < j < n only; zg is ZG_OUTSIDE elsewhere: p = 0 there)
            const double c = P.c, omc = P.omc;
            const double2 cv0 = s_node[fs][c0], cv1 = s_node[fs][c0 + 1];
            double colp0 = 0, colp1 = 0;
            double rsp[FRMAX];
#pragma unroll
            for (int f = 0; f < FRMAX; ++f) {
                rsp[f] = 0.0;
                if (f < fcnt) {
                    const int a = fa0 + f;
                    const double2 rv = s_node[fs][2 * a + half];
                    double pp[2], t[2];
                    tl_edge(c, fw[f].x, fgz[f].x, rv.x, rv.y, cv0.x, cv0.y, pp[0], t[0]);
                    tl_edge(c, fw[f].y, fgz[f].y, rv.x, rv.y, cv1.x, cv1.y, pp[1], t[1]);
                    if (NORMS) {  // (without the norms the entries outside the lattice carry a bounded value nobody reads)
                        const int64_t i = (int64_t)fBl * TB + 2 * a + half, j = (int64_t)fBl * TB + c0;
                        t[0] = (i < j && j < P.n) ? t[0] : 0.0;  // (w is 0 there and stays 0)
                        t[1] = (i < j + 1 && j + 1 < P.n) ? t[1] : 0.0;
                    }
                    tl_new_w<NORMS>(c, omc, fw[f].x, t[0], nqy, nw);
                    tl_new_w<NORMS>(c, omc, fw[f].y, t[1], nqy, nw);
                    colp0 += pp[0];
                    colp1 += pp[1];
                    rsp[f] = half_sum(pp[0] + pp[1]);
                }
            }
            colp0 = add_xor32(colp0);
            colp1 = add_xor32(colp1);
            // Rows and columns of a diagonal tile are the same node block, and that block is my chunk's row or
            // column block (or I have no chunk): the group's partials are folded into that vector of mine.
            // LDS operations of one wave execute in order.
            double *vp = &s_part[wave][ftgt][0];
            if (half == 0) {
                // (the row vector of a chunk shorter than a tile is zero outside its rows: rewritten every pass)
                const bool keep0 = has && (ftgt == 1 || (c0 >= 2 * a0 && c0 < 2 * (a0 + U)));
                const double b0 = keep0 ? vp[c0] : 0.0, b1 = keep0 ? vp[c0 + 1] : 0.0;
                vp[c0] = b0 + colp0;
                vp[c0 + 1] = b1 + colp1;
            }
            if ((lane & 31) == 0) {
#pragma unroll
                for (int f = 0; f < FRMAX; ++f)
                    if (f < fcnt) vp[2 * (fa0 + f) + half] += rsp[f];
            }
        }
#if GL_TILE_EARLY_STORE
        if constexpr (!NORMS && !BIG && !BATCH && !FIRST) {
            if (!rule_on && it + 1 == P.maxit) store_w();  // the call's last pass is done: w is final
        }
#endif
        if constexpr (BATCH) {
            if (P.ring) {  // (fire and forget: drained with everything else at the bottom of the iteration, a hand-off later)
                int ld = lane;
                asm volatile("" : "+v"(ld));  // (the offsets are formed here and now: kept over the pass they cost it registers)
                double *slot = P.ring + (int64_t)(it + 1) * P.ring_stride;
                if (has) {
                    const int64_t tb0 = chunk_off(ld);
#pragma unroll
                    for (int s = 0; s < U; ++s) *reinterpret_cast<double2 *>(slot + tb0 + s * (2 * TB)) = wreg[s];
                }
#pragma unroll
                for (int f = 0; f < FRMAX; ++f)
                    if (f < fcnt) {
                        const int64_t i = (int64_t)fBl * TB + 2 * (fa0 + f) + (ld >> 5), j = (int64_t)fBl * TB + 2 * (ld & 31);
                        *reinterpret_cast<double2 *>(slot + fringe_off(ld) + (int64_t)(fa0 + f) * (2 * TB)) =
                            double2{(i < j && j < P.n) ? fw[f].x : 0.0, (i < j + 1 && j + 1 < P.n) ? fw[f].y : 0.0};
                    }
            }
        }
        if (NORMS) {
            nqy = wave_sum(nqy);
            nw = wave_sum(nw);
        }
        if (NORMS && lane == 0) {
            // ||w_old||^2 of this iteration was accumulated when w_old was produced (squaring the old values inside
            // the pass keeps them alive and costs 75 registers): the sum over the new values is for the next one
            s_wn[wave][0] = nqy;
            s_wn[wave][1] = s_wcarry[wave];
            s_wcarry[wave] = nw;
        }
        // (the rule's early look, after the publish phase: its records are requested now)
        const bool look_on = !BATCH && (NORMS ? (rule_on && !exact) : gate_on) && wg == P.rule_wg && wave == TW - 1 && it >= 1 && it + 1 < P.maxit;
        tl_u4 lk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) lk[q] = tl_u4{0u, P.tag0 + (unsigned)it, 0u, P.tag0 + (unsigned)it};
        if (!BIG && look_on && lane < P.nb) {
#pragma unroll
            for (int q = QN0; q < 4; ++q)
                lk[q] = __builtin_amdgcn_raw_buffer_load_b128(r_bn, ((((it - 1) & 3) * (int)P.nb + lane) * 4 + q) * 16, 0, 16);
        }
        if (stamp) ts[2] = __builtin_amdgcn_s_memrealtime();
        if (stamp && lane == 0) {  // per-wave duration of the pass (after the workgroup's stamps)
            unsigned long long *o = P.dbg + (size_t)P.NWG * 16 * 6 + (((size_t)wg * 16 + (it - 8)) * TW + wave) * 2;
            o[0] = ts[1];
            o[1] = ts[2];
        }
        tl_barrier();

        // ================= publish: one record per node block I touch =================
#pragma unroll
        for (int j = 0; j < NACQ; ++j) {
            const int k = wave + TW * j;
            if (k >= nblk) break;
            double sp = accp[j];
            int rm = gd_int(GD_ROW + k);
            const int cm = gd_int(GD_COL + k);
            if (BIG) {  // the row vectors of the last round
                const int r = gd_int(GD_NRND) - 1;
                rm = ((r < 4 ? rm : gd_int(GD_ROW2 + k)) >> (8 * (r & 3))) & 255;
            }
            // All sixteen vectors are requested at once and the ones the masks leave out are zeroed by an AND (x + 0.0 = x: the
            // sum and its order are what they were): written as `if (bit) sp += s_part[..]` the masks -- wave-uniform -- become
            // sixteen scalar branches with one LDS round trip each, three to five of them taken one after the other, between
            // the end of the workgroup's slowest pass and its records (0.5 us measured; the owners wait for exactly that).
            // (two halves: sixteen values at once cost the variants with the norms, which have no register to spare, a few
            //  dwords of scratch right here)
            int lp = lane;
            asm volatile("" : "+v"(lp));  // (the LDS addresses are formed here and now: hoisted out of the iteration loop they live in scratch)
#pragma unroll
            for (int h4 = 0; h4 < TW; h4 += TW / 2) {
                double rvec[TW / 2], cvec[TW / 2];
#pragma unroll
                for (int q = 0; q < TW / 2; ++q) {
                    rvec[q] = s_part[h4 + q][0][lp];
                    cvec[q] = s_part[h4 + q][1][lp];
                }
#pragma unroll
                for (int q = 0; q < TW / 2; ++q) {
                    sp += tl_keep(rvec[q], rm >> (h4 + q));
                    sp += tl_keep(cvec[q], cm >> (h4 + q));
                }
            }
            __builtin_amdgcn_raw_buffer_store_b128(tl_gran(sp, tag_out), r_part, (gd_int(GD_ENT + k) * TB + lp) * 16, 0, 16);
        }
        if (NORMS && tid == 0) {  // (without the norms nobody reads these records)
            double a = s_wn[0][0], b = s_wn[0][1];
#pragma unroll
            for (int v = 1; v < TW; ++v) {
                a += s_wn[v][0];
                b += s_wn[v][1];
            }
            const int off = (((it & 3) * P.NWG + wg) * 2) * 16;
            __builtin_amdgcn_raw_buffer_store_b128(tl_gran(a, tag_out), r_wn, off, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(tl_gran(b, tag_out), r_wn, off + 16, 0, 16);
        }
        done_iters = it + 1;
        if (stamp) ts[3] = tx[1] = __builtin_amdgcn_s_memrealtime();
        // The rule's early look for the top of iteration it + 1: the records of iteration it - 1 (a whole iteration old).
        // ONE workgroup takes it for all (P.rule_wg: no node block of its own, so it would only wait for node values now)
        // and publishes what it found as a tagged record that every workgroup reads at its next top -- in the owners the
        // look sat in front of the poll of the degree records (0.9 us per iteration).  Its loads were issued before the
        // publish phase (lk, above) and are only looked at now; the screen compares squares (no square root, no
        // division: 0.4 us for a lone wave); the ratios themselves are formed only when the look stops the run.
        auto early_look = [&]() {
            if (!look_on) return;
            double b[4];
            bool have = false;
            if (!BIG) {
                const unsigned tag = P.tag0 + (unsigned)it;  // records of local iteration it - 1
                bool ok = true;
#pragma unroll
                for (int q = QN0; q < 4; ++q) ok = ok && tl_ok(lk[q], tag);
                if (__all(ok)) {
                    have = true;
#pragma unroll
                    for (int q = QN0; q < 4; ++q) b[q] = wave_sum(lane < P.nb ? tl_val(lk[q]) : 0.0);
                }
            }
            int look = 0;
            if (!have) look = norms_of(it - 1, b) ? 0 : -1;
            double lp = INFINITY, ld = INFINITY;
            if (!NORMS) {  // the gate: the dual ratio alone, against gate x tol (squares: no root, no division)
                if (look == 0 && P.it0 + it - 1 >= 1) {
                    double tl = P.gate;
                    asm volatile("" : "+s"(tl));
                    if (b[2] < tl * tl * b[3]) look = 2;
                }
            } else if (look == 0 && P.it0 + it - 1 >= 1) {  // never for iteration 0 (:156-161)
                double tl = P.tol;
                asm volatile("" : "+s"(tl));  // (formed here: hoisted out of the iteration loop the squares live in scratch)
                const double t2 = tl * tl, u2 = (1.25 * tl) * (1.25 * tl);
                if (b[0] < t2 * b[1] && b[2] < t2 * b[3]) {
                    lp = sqrt(b[0]) / sqrt(b[1]), ld = sqrt(b[2]) / sqrt(b[3]);
                    look = (lp < P.tol && ld < P.tol) ? 1 : 2;
                } else if (b[0] < u2 * b[1] && b[2] < u2 * b[3]) {
                    look = 2;
                }
            }
            if (lane < 3) {  // for every workgroup's next top
                const double v = lane == 0 ? (double)look : (lane == 1 ? lp : ld);
                __builtin_amdgcn_raw_buffer_store_b128(tl_gran(v, tag_out), r_lk, (((it & 3) * 4) + lane) * 16, 0, 16);
            }
        };
        if (own < 0) early_look();
        if constexpr (BATCH) {
            // A batch's look at iteration it - 1 for the next top (screen_bottom).  A consumer would only wait for node values now.
            if (own < 0 && rule_on && wave == RW) {
                tl_u4 none[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) none[q] = tl_u4{0u, 0u, 0u, 0u};
                screen_bottom(it, none, false);
            }
        }

        // ================= node phase: the owner of a block sums its records and updates the dual state =================
        if (own >= 0) {
            tl_barrier();  // everybody has read s_part (its first 4 KiB become s_own now)
            double (*s_own)[TB] = reinterpret_cast<double (*)[TB]>(&s_part[0][0][0]);
            bool good = true;
            double sp = 0;
            // (wave 0: what its update needs of the dual state, requested while the records are still on their way)
            int lnp = lane;
            asm volatile("" : "+v"(lnp));  // (the address is formed here and now, see below)
            // (not in the variants with the norms: eight more registers across the poll put theirs into scratch)
            double d_cur = 0, yb_cur = 0, pb_cur = 0, gsw_cur = 0;
            if (!NORMS && wave == 0) d_cur = s_dual[0][lnp], yb_cur = s_dual[1][lnp], pb_cur = s_dual[2][lnp], gsw_cur = s_dual[4][lnp];
            tl_u4 spre[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) spre[q] = tl_u4{0u, 0u, 0u, 0u};
            bool have_pre = false;
            {
                early_look();  // (wave TW - 1: before its share of the poll, whose records are still on their way)
                if constexpr (BATCH && U < 16) {  // (wave RW: the records its look will read, requested now -- their round trip passes during the poll; U = 16 and 32 have no 16 registers to hold them across it: 68 and 300 bytes of scratch)
                    if (rule_on && wave == RW && it >= 1 && P.it0 + it - 1 >= 1) {
                        screen_request(it - 1, spre);
                        have_pre = true;
                    }
                }
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                unsigned spins = 0;
                // wave 1 also fetches the block sums of gamma u that this iteration's input state was published with (a whole
                // pass ago): sum_j gamma u_j over all nodes, in a fixed order
                constexpr int NUS = BIG ? 3 : 1;  // node blocks per lane (nb <= 64, rounds <= 192)
                const int gb0 = BATCH ? own - in_graph(own) : 0, gnb = BATCH ? nb1 : (int)P.nb;  // my graph's node blocks
                tl_u4 gu[NUS];
                auto load_gu = [&]() {
#pragma unroll
                    for (int q = 0; q < NUS; ++q) {
                        gu[q] = tl_u4{0u, tag_in, 0u, tag_in};
                        if (wave == 1 && q * 64 + lane < gnb)
                            gu[q] = __builtin_amdgcn_raw_buffer_load_b128(r_us, (int)(((tag_in & 1u) * (unsigned)P.nb + (unsigned)(gb0 + q * 64 + lane)) * 16u), 0, 16);
                    }
                };
                load_gu();
                for (int eb = own_e0 + wave; eb < own_e1 && good; eb += TW * RND) {  // rounds of up to RND records per wave
                    tl_u4 g[RND];
                    auto probe = [&](tl_u4 (&q)[RND]) {
#pragma unroll
                        for (int u = 0; u < RND; ++u) {
                            const int e = eb + u * TW;
                            q[u] = tl_u4{0u, tag_out, 0u, tag_out};
                            if (e < own_e1) q[u] = __builtin_amdgcn_raw_buffer_load_b128(r_part, (e * TB + lane) * 16, 0, 16);
                        }
                        asm volatile("" ::: "memory");
                    };
                    auto valid = [&](const tl_u4 (&q)[RND]) {
                        bool ok = true;
#pragma unroll
                        for (int u = 0; u < RND; ++u) ok = ok && tl_ok(q[u], tag_out);
                        return __all(ok);
                    };
#if GL_TILE_PROBES == 2
                    tl_u4 g2[RND];  // (two probes in flight, as in the acquire)
                    probe(g);
                    __builtin_amdgcn_s_sleep(GL_TILE_PROBE_GAP);
                    probe(g2);
                    for (;;) {
                        if (valid(g)) break;
                        probe(g);
                        if (valid(g2)) {
#pragma unroll
                            for (int u = 0; u < RND; ++u) g[u] = g2[u];
                            break;
                        }
                        probe(g2);
                        if (!tl_spin(spins, t0, err)) {
                            good = false;
                            break;
                        }
                    }
#else
                    for (;;) {
                        probe(g);
                        if (valid(g)) break;
                        if (!tl_spin(spins, t0, err)) {
                            good = false;
                            break;
                        }
                    }
#endif
#pragma unroll
                    for (int u = 0; u < RND; ++u)
                        if (eb + u * TW < own_e1) sp += tl_val(g[u]);
                }
                if (stamp) tx[2] = __builtin_amdgcn_s_memrealtime();
                s_own[wave][lane] = sp;
                if (wave == 1) {
                    for (;;) {
                        bool ok = true;
#pragma unroll
                        for (int q = 0; q < NUS; ++q) ok = ok && tl_ok(gu[q], tag_in);
                        asm volatile("" ::: "memory");
                        if (__all(ok)) break;
                        if (!tl_spin(spins, t0, err)) {
                            good = false;
                            break;
                        }
                        load_gu();
                    }
                    double t = 0.0;
#pragma unroll
                    for (int q = 0; q < NUS; ++q) t += (q * 64 + lane < gnb) ? tl_val(gu[q]) : 0.0;
                    t = wave_sum(t);
                    if (lane == 0) s_gu = t;
                }
                if (!good) s_flag[1] = 1;
            }
            if (stamp) ts[4] = __builtin_amdgcn_s_memrealtime();
            tl_barrier();
            if (s_flag[1]) {
                failed = true;
                fail_it = it;
                return false;
            }
            if (wave == 0) {
                // (addresses formed from the lane number here and now: kept over the pass they end up in scratch in the
                //  variants with norms, and their reloads sit between the owner's poll and the node values it publishes)
                int ln = lane;
                asm volatile("" : "+v"(ln));
                sp = s_own[0][ln];
#pragma unroll
                for (int v = 1; v < TW; ++v) sp += s_own[v][ln];
                const bool real = (int64_t)in_graph(own) * TB + ln < P.n;
                if (NORMS) d_cur = s_dual[0][ln], yb_cur = s_dual[1][ln], pb_cur = s_dual[2][ln], gsw_cur = s_dual[4][ln];
                const double gsp = P.gamma * sp;
                const double qb = pb_cur + gsp;           // :154
                const double dq = qb - yb_cur;            // == -ybar + qbar (:159)
                double dq2 = real ? dq * dq : 0.0;
                double d2 = real ? d_cur * d_cur : 0.0;
                const double dnew = (d_cur - yb_cur) + qb;                  // :164
                // gamma S w' = (1 - c) gamma S w + c gamma S p + gamma ((n - 2) gamma u_i + sum_j gamma u_j)  (head of the file)
                const double gu_i = P.gamma * (d_cur - pb_cur);
                const double gsw = __builtin_fma(P.omc, gsw_cur, __builtin_fma(P.c, gsp, P.gamma * __builtin_fma(P.nm2, gu_i, s_gu)));
                const double ybn = dnew + gsw;                              // :148 of the next iteration
                const double pbn = (ybn - sqrt(ybn * ybn + P.a4g)) / 2;     // :151
                const double d_new = real ? dnew : 0.0, pb_new = real ? pbn : 0.0;
                s_dual[3][ln] = d_cur;
                s_dual[0][ln] = d_new;
                if constexpr (BATCH) {
                    if (P.dring) {  // the dual state after this iteration, as the end of a call would leave it: {d, ybar, pbar, dprev}
                        double *ds = P.dring + (int64_t)(it + 1) * P.dring_stride + (int64_t)own * TB + ln;
                        const int64_t np = (int64_t)P.nb * TB;
                        ds[0] = d_new;
                        ds[np] = real ? ybn : 0.0;
                        ds[2 * np] = pb_new;
                        ds[3 * np] = d_cur;
                    }
                }
                s_dual[1][ln] = real ? ybn : 0.0;
                s_dual[2][ln] = pb_new;
                s_dual[4][ln] = real ? gsw : 0.0;
                const int off = (own * 2 * TB + ln) * 16;
                const double gu_new = P.gamma * (d_new - pb_new);
                __builtin_amdgcn_raw_buffer_store_b128(tl_gran(P.gamma * d_new, tag_out), r_nval, off, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b128(tl_gran(gu_new, tag_out), r_nval, off + TB * 16, 0, 16);
                const double us = wave_sum(gu_new);
                if (ln == 0) __builtin_amdgcn_raw_buffer_store_b128(tl_gran(us, tag_out), r_us, (int)(((tag_out & 1u) * (unsigned)P.nb + (unsigned)own) * 16u), 0, 16);
                if (stamp) tx[3] = __builtin_amdgcn_s_memrealtime();
                if (NORMS || gate_on) {  // my block's norm partials (the node values are out: nobody waits for these)
                    dq2 = wave_sum(dq2);
                    d2 = wave_sum(d2);
                    if (ln == 2 || ln == 3) {
                        const double v = ln == 2 ? dq2 : d2;
                        s_bn[ln] = v;
                        __builtin_amdgcn_raw_buffer_store_b128(tl_gran(v, tag_out), r_bn, ((((it & 3) * (int)P.nb + own) * 4) + ln) * 16, 0, 16);
                        if constexpr (BATCH) {  // (... and its norm partials, behind the four vectors)
                            if (P.dring) P.dring[(int64_t)(it + 1) * P.dring_stride + 4 * (int64_t)P.nb * TB + 4 * own + ln] = v;
                        }
                    }
                }
            } else if (BATCH && rule_on && wave == RW) {
                // An owner's look: beside wave 0's update and wave TW - 1's slice.  The node values leave on time; the workgroup itself
                // reaches its next top a few tenths of a microsecond later, where it waits a hand-off for the other owners' values
                // anyway.  (Measured elsewhere: in front of the owner's poll for its records the look held the barrier behind that
                // poll back -- 6.4 us per iteration at n = 256, B = 10; at the very end of the iteration it held the next top's
                // barrier back -- 6.7; at the top itself it was 7.1; 5.3 without the rule.  The lambdas of the look are always_inline:
                // left to the inliner's judgement the kernel's argument block -- captured by reference two lambdas deep -- stayed
                // in scratch, 450-720 bytes.)
                screen_bottom(it, spre, have_pre);
            } else if (NORMS && wave == TW - 1) {
                // my slice of the workgroups' norm records (workgroups own, own + nb, ...), beside wave 0's update and off the
                // owner's critical path (in front of the barrier above it cost 0.5-1 us per iteration: the node values of
                // the whole chip waited for it)
                // (a batch: the workgroups of MY graph, so that a block's partials belong to one graph)
                const int nsrc = BATCH ? (own / nb1) * P.NWG1 + in_graph(own) + lane * nb1 : own + lane * (int)P.nb;
                const bool nmine = (BATCH ? nsrc < (own / nb1 + 1) * P.NWG1 : nsrc < P.NWG) && ((slice_mask >> lane) & 1ull);
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                unsigned spins = 0;
                tl_u4 na, nb2;
                bool got = true;
                for (;;) {
                    na = nb2 = tl_u4{0u, tag_out, 0u, tag_out};
                    if (nmine) {
                        const int off = (((it & 3) * P.NWG + nsrc) * 2) * 16;
                        na = __builtin_amdgcn_raw_buffer_load_b128(r_wn, off, 0, 16);
                        nb2 = __builtin_amdgcn_raw_buffer_load_b128(r_wn, off + 16, 0, 16);
                    }
                    asm volatile("" ::: "memory");
                    if (__all(tl_ok(na, tag_out) && tl_ok(nb2, tag_out))) break;
                    if (!tl_spin(spins, t0, err)) {
                        got = false;
                        break;
                    }
                }
                const double n0 = wave_sum(nmine ? tl_val(na) : 0.0), n1 = wave_sum(nmine ? tl_val(nb2) : 0.0);
                int l2 = lane;
                asm volatile("" : "+v"(l2));  // (addresses formed from the lane number here and now, as in wave 0's update)
                if (!got) {
                    if (l2 == 0) s_flag[1] = 1;  // (seen at the next top)
                } else if (l2 < 2) {
                    const double v = l2 == 0 ? n0 : n1;
                    s_bn[l2] = v;
                    __builtin_amdgcn_raw_buffer_store_b128(tl_gran(v, tag_out), r_bn, ((((it & 3) * (int)P.nb + own) * 4) + l2) * 16, 0, 16);
                    if constexpr (BATCH) {
                        if (P.dring) P.dring[(int64_t)(it + 1) * P.dring_stride + 4 * (int64_t)P.nb * TB + 4 * own + l2] = v;
                    }
                }
            }
            tl_barrier();  // s_own has been read: s_part is free for the next pass (its zeros first, below)
            if (U < SLOTS) s_part[wave][0][lane] = 0.0;  // s_own overwrote row vectors: back to zero outside the chunks' rows
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see the prologue (by now everything has long completed)
        if (it == 0) call_stamp(3);  // end of the first iteration
        if (stamp && lane == 0) {
            unsigned long long *o = P.dbg + (size_t)P.NWG * (16 * (6 + 2 * TW) + 8 * TW) + (((size_t)wg * 16 + (it - 8)) * TW + wave) * 4;
            for (int q = 0; q < 4; ++q) o[q] = tx[q];
        }
        if (stamp && tid == 0) {
            ts[5] = __builtin_amdgcn_s_memrealtime();
            unsigned long long *o = P.dbg + ((size_t)wg * 16 + (it - 8)) * 6;
            for (int q = 0; q < 6; ++q) o[q] = ts[q];
        }
        return true;
    };
    if (met && iteration(std::true_type{}, 0))
        for (int it = 1; it <= P.maxit; ++it)
            if (!iteration(std::false_type{}, it)) break;

    if (failed) {  // nothing of the session's state has been written
        if (wg == 0 && tid == 0) {
            // where the session stood: ctr[1] for gl_resident_finish (synchronous calls), ctr[2] for resident_check
            // (asynchronous ones).  ctr[1] may hold anything here -- the streaming engine re-arms it to 0 after its calls
            P.ctr[1] = P.it0;
            P.ctr[2] = P.it0;
            // 1: the workgroups never met (also: below the rounds, the first hand-offs of the call never arrived -- the same
            // thing seen one step later, and nothing has been written); 2: a hand-off was lost in mid-run
            P.ctr[4] = (!met || (!BIG && fail_it <= 1)) ? 1 : 2;
        }
        return;
    }
    // ---- registers back to the iterate buffer, dual state back to memory, bookkeeping for the host / the other engines.
    //      Entries outside i < j < n return to the exact zeros every engine expects there.
    call_stamp(4);
    if (!w_stored) store_w();
    if (own >= 0 && wave == 0 && done_iters > 0) {
        P.dvar[own_i] = s_dual[0][lane];
        P.ybar[own_i] = s_dual[1][lane];
        P.pbar[own_i] = s_dual[2][lane];
        P.dprev[own_i] = s_dual[3][lane];
        if (lane < 4) P.bnorm[((P.it0 + done_iters - 1) & 1) * 4 * P.nb + 4 * own + lane] = s_bn[lane];
    }
    if (P.dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        call_stamp(5);
    }
    if (wg == 0 && tid == 0) {
        const long long itf = P.it0 + done_iters;
        if (evaluated) {  // the ratios of the last iteration whose rule was evaluated (inf for iteration 0, :156-161)
            P.dstat[0] = rel_p;
            P.dstat[1] = rel_d;
        }
        P.ctr[0] = stopped;
        P.ctr[1] = itf;
        P.ctr[2] = itf;
        P.ctr[3] = itf > 0 ? itf - 1 : 0;
        P.ctr[4] = 0;
    }
}

// zg = 2 gamma z in the tiled layout (the product the reference forms, :150, rounded the same way), ZG_OUTSIDE at every
// entry outside i < j < n.  One workgroup per tile; formed once per gl_logdeg_init.
__global__ __launch_bounds__(256) void k_make_zg(const double *__restrict__ z, double *__restrict__ zg, int64_t n, int64_t nb, double g2) {
    int I, J;
    gl_tile_decode((int64_t)blockIdx.x, nb, I, J);
    const int64_t base = (int64_t)blockIdx.x * GL_TILE_ELEMS;
#pragma unroll
    for (int k = 0; k < GL_TILE_ELEMS / (2 * 256); ++k) {
        const int e = 2 * ((int)threadIdx.x + 256 * k);  // two neighbouring columns of one row
        const int r = e / TB, c = e % TB;
        const int64_t i = (int64_t)I * TB + r, j = (int64_t)J * TB + c;
        const double2 v = *reinterpret_cast<const double2 *>(z + base + e);
        double2 o;
        o.x = (i < j && j < n) ? g2 * v.x : ZG_OUTSIDE;
        o.y = (i < j + 1 && j + 1 < n) ? g2 * v.y : ZG_OUTSIDE;
        *reinterpret_cast<double2 *>(zg + base + e) = o;
    }
}

// the same for a batch: blockIdx.y = graph, distances from the sessions' own buffers, tiles of graph g at g ntiles1
__global__ __launch_bounds__(256) void k_make_zg_batch(const LogdegDev *__restrict__ devs, double *__restrict__ zg, int64_t n, int64_t nb1,
                                                       int64_t ntiles1, double g2) {
    int I, J;
    gl_tile_decode((int64_t)blockIdx.x, nb1, I, J);
    const double *z = devs[blockIdx.y].z + (int64_t)blockIdx.x * GL_TILE_ELEMS;
    double *o_ = zg + ((int64_t)blockIdx.y * ntiles1 + blockIdx.x) * GL_TILE_ELEMS;
#pragma unroll
    for (int k = 0; k < GL_TILE_ELEMS / (2 * 256); ++k) {
        const int e = 2 * ((int)threadIdx.x + 256 * k);
        const int r = e / TB, c = e % TB;
        const int64_t i = (int64_t)I * TB + r, j = (int64_t)J * TB + c;
        const double2 v = *reinterpret_cast<const double2 *>(z + e);
        double2 o;
        o.x = (i < j && j < n) ? g2 * v.x : ZG_OUTSIDE;
        o.y = (i < j + 1 && j + 1 < n) ? g2 * v.y : ZG_OUTSIDE;
        *reinterpret_cast<double2 *>(o_ + e) = o;
    }
}

}  // namespace

// =========================================================================== host side
struct gl_tile {
    TileDev dev{};
    int U = SLOTS;              // row pairs per chunk: 32, 16 or 8
    bool big = false;           // rounds (n > 4096)
    bool nt = false;            // ... whose streamed tiles no longer fit the Infinity Cache: non-temporal loads and stores
    void *arena = nullptr;      // descriptors, record buffers, rendezvous words: one allocation
    size_t gran_off = 0, gran_bytes = 0;  // the record buffers inside it
    unsigned long long *dbg = nullptr;
    double *zg = nullptr;       // 2 gamma z in the tiled layout, ZG_OUTSIDE outside i < j < n (inside the arena) ...
    double zg_g2 = 0.0;         // ... the factor it was formed with (0: not yet) ...
    int64_t zg_version = -1;    // ... and the version of the session's distances it was formed from
    unsigned next_tag = 1u;     // first tag of the next launch (a launch uses maxit + 2 consecutive ones, never 0)
    unsigned arrivals = 0;      // workgroups that have arrived at the start-up rendezvous over all launches so far
    bool batch = false;         // a batch of graphs as one block-diagonal lattice (gl_tile_prepare_batch)
    double *ring = nullptr, *dring = nullptr;  // ... whose next launch writes every iterate out (gl_tile_batch_set_ring)
    int64_t ring_stride = 0, dring_stride = 0;
    int nb1 = 0, NWG1 = 0;      // ... node blocks and workgroups of ONE graph
    int64_t ntiles1 = 0;        // ... and its tiles
};

void gl_tile_free(gl_logdeg *s) {
    if (!s || !s->tile) return;
    if (s->tile->arena) (void)hipFree(s->tile->arena);
    if (s->tile->dbg) (void)hipFree(s->tile->dbg);
    delete s->tile;
    s->tile = nullptr;
}

// slices, record entries, device arena: common to both decompositions
// (batch_nb1 > 0: a batch -- the lattice of batch_nb1 node blocks and batch_nwg1 workgroups replicated; the owner of a node
//  block sums the norm records of the workgroups of ITS graph)
static int tile_finish(gl_logdeg *s, std::vector<WaveDesc> &wd, std::vector<WgDesc> &gd, const int NWG, const int U, const bool big,
                       const int batch_nb1 = 0, const int batch_nwg1 = 0) {
    gl_ctx *ctx = s->ctx;
    const LogdegDev &D = s->dev;
    const int nb = (int)D.nb;
    const int F = nb * (nb - 1) / 2;
    const int nb1 = batch_nb1 > 0 ? batch_nb1 : nb, nwg1 = batch_nb1 > 0 ? batch_nwg1 : NWG;
    std::vector<std::vector<int>> touch((size_t)nb);  // block -> workgroups (ascending)
    std::vector<int> owner((size_t)nb, -1);           // block -> the workgroup that owns it
    for (int g = 0; g < NWG; ++g) {
        for (int k = 0; k < gd[(size_t)g].nblk; ++k) touch[(size_t)gd[(size_t)g].blk[k]].push_back(g);
        if (gd[(size_t)g].own >= 0) owner[(size_t)gd[(size_t)g].own] = g;
    }
    for (int B = 0; B < nb; ++B) {
        if (owner[(size_t)B] < 0) return gl_set_error(ctx, GL_E_STATE, "internal: node block %d has no owner", B);
        const int first = (B / nb1) * nwg1, bl = B % nb1;
        for (int r = 0; bl + r * nb1 < nwg1; ++r) {
            if (r >= 64) return gl_set_error(ctx, GL_E_STATE, "internal: more than 64 norm records per owner");
            if (gd[(size_t)(first + bl + r * nb1)].active)
                (r < 32 ? gd[(size_t)owner[(size_t)B]].slice_lo : gd[(size_t)owner[(size_t)B]].slice_hi) |= 1u << (r & 31);
        }
    }
    int n_entries = 0;
    for (int B = 0; B < nb; ++B) {
        gd[(size_t)owner[(size_t)B]].own_e0 = n_entries;
        for (int g : touch[(size_t)B]) {
            WgDesc &G = gd[(size_t)g];
            for (int k = 0; k < G.nblk; ++k)
                if (G.blk[k] == B) G.ent[k] = n_entries;
            ++n_entries;
        }
        gd[(size_t)owner[(size_t)B]].own_e1 = n_entries;
        if (touch[(size_t)B].empty()) return gl_set_error(ctx, GL_E_STATE, "internal: node block %d has no producer", B);
    }

    gl_tile *t = new (std::nothrow) gl_tile();
    if (!t) return GL_E_NOMEM;
    auto up256 = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t b_wd = up256(wd.size() * sizeof(WaveDesc)), b_gd = up256(gd.size() * sizeof(WgDesc));
    const size_t b_part = up256((size_t)n_entries * TB * 16), b_nval = up256((size_t)nb * 2 * TB * 16), b_us = up256((size_t)2 * nb * 16);
    const size_t b_wn = up256((size_t)4 * NWG * 2 * 16), b_bn = up256((size_t)4 * nb * 4 * 16), b_lk = up256((size_t)4 * 4 * 16);
    const size_t b_bar = up256((size_t)GL_BAR_WORDS * 4);
    const size_t b_zg = up256((size_t)D.ntiles * GL_TILE_ELEMS * sizeof(double));
    const size_t total = b_wd + b_gd + b_part + b_nval + b_us + b_wn + b_bn + b_lk + b_bar + b_zg;
    if (gl_ctx_malloc(ctx, &t->arena, total, "tile") != hipSuccess) {
        delete t;
        return gl_set_error(ctx, GL_E_NOMEM, "tile engine: hipMalloc(%zu) failed", total);
    }
    char *at = (char *)t->arena;
    TileDev &P = t->dev;
    P.wd = (const WaveDesc *)at, at += b_wd;
    P.gd = (const WgDesc *)at, at += b_gd;
    t->gran_off = (size_t)(at - (char *)t->arena);
    P.part = (tl_u4 *)at, at += b_part;
    P.nval = (tl_u4 *)at, at += b_nval;
    P.usum = (tl_u4 *)at, at += b_us;
    P.wnorm = (tl_u4 *)at, at += b_wn;
    P.bnrm = (tl_u4 *)at, at += b_bn;
    P.look = (tl_u4 *)at, at += b_lk;
    t->gran_bytes = (size_t)(at - (char *)t->arena) - t->gran_off;
    P.bar = (unsigned *)at, at += b_bar;
    t->zg = (double *)at;
    P.zg = t->zg;
    hipError_t e = hipMemsetAsync(t->arena, 0, total - b_zg, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync((void *)P.wd, wd.data(), wd.size() * sizeof(WaveDesc), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync((void *)P.gd, gd.data(), gd.size() * sizeof(WgDesc), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // wd / gd are locals
    if (e != hipSuccess) {
        (void)hipFree(t->arena);
        delete t;
        return gl_set_error(ctx, GL_E_HIP, "tile engine set-up: %s", hipGetErrorString(e));
    }
    P.dvar = D.dvar;
    P.ybar = D.ybar;
    P.pbar = D.pbar;
    P.dprev = D.dprev;
    P.bnorm = D.bnorm;
    P.dstat = D.dstat;
    P.ctr = D.ctr;
    P.n = D.n;
    P.nb = D.nb;
    P.NWG = NWG;
    P.n_entries = n_entries;
    P.rule_wg = NWG - 1;  // the rule's early look: a workgroup that has edges but no node block of its own, if there is one
    for (int g = NWG - 1; g >= 0; --g)
        if (gd[(size_t)g].active && gd[(size_t)g].own < 0) {
            P.rule_wg = g;
            break;
        }
    if (!gd[(size_t)P.rule_wg].active) P.rule_wg = 0;  // (workgroup 0 owns block 0: always active)
    t->U = U;
    t->big = big;
    t->batch = batch_nb1 > 0;
    t->nb1 = nb1;
    t->NWG1 = nwg1;
    t->ntiles1 = batch_nb1 > 0 ? D.ntiles / (nb / nb1) : D.ntiles;
    if (big) {  // what streams per iteration: w and z of every tile no wave keeps in registers
        int64_t streamed = 0;
        for (const WaveDesc &w : wd) streamed += w.nr - (w.tile >= 0 ? 1 : 0);
        for (const WgDesc &G : gd) streamed += G.nx;
        // (the hint on z pays from n ~ 6400 on: w + z of the streamed tiles ~ 200 MB, see tl_ld_stream)
        t->nt = streamed * GL_TILE_ELEMS * 16 > ((int64_t)200 << 20);
        if (const char *e = getenv("GRAPHLEARN_AMD_TILE_NT")) t->nt = atoi(e) != 0;
    }
    P.dbg = nullptr;
    if (getenv("GRAPHLEARN_AMD_RES_STAMPS") && hipMalloc((void **)&t->dbg, (size_t)NWG * (16 * (6 + 2 * TW) + 8 * TW + 16 * 4 * TW) * 8) == hipSuccess) {
        (void)hipMemset(t->dbg, 0, (size_t)NWG * (16 * (6 + 2 * TW) + 8 * TW + 16 * 4 * TW) * 8);
        P.dbg = t->dbg;
        int maxblk = 0;
        for (const WgDesc &G : gd) maxblk = std::max(maxblk, G.nblk);
        fprintf(stderr, "[tile] n %lld nb %d F %d U %d NWG %d records %d (%.1f per block) max blocks per workgroup %d\n", (long long)D.n, nb,
                F, U, NWG, n_entries, (double)n_entries / nb, maxblk);
    }
    s->tile = t;
    return GL_OK;
}

// n > 4096: rectangles of 8 block columns x R block rows, one per workgroup, worked through in rounds (k_tile, BIG).
// Column strip t holds block columns 8 t .. 8 t + 7 and the block rows 0 .. its last column; it is cut into groups of
// R rows, R the smallest number of rounds with which the chip's workgroups cover the lattice (R + 8 <= NBW node
// blocks per workgroup).  Wave v of a workgroup takes column J = 8 t + v: tiles (I, J), I = I0 .. min(I0 + R - 1, J).
static int tile_prepare_rounds(gl_logdeg *s, bool allow_split = true) {
    auto why = [&](int code) {  // (GRAPHLEARN_AMD_TILE_DEBUG: which condition sent the session to the streaming engine)
        if (getenv("GRAPHLEARN_AMD_TILE_DEBUG")) fprintf(stderr, "[tile] rounds do not apply to n = %lld: condition %d\n", (long long)s->dev.n, code);
        return GL_OK;
    };
    gl_ctx *ctx = s->ctx;
    const LogdegDev &D = s->dev;
    const int ncu = ctx->prop.multiProcessorCount, nb = (int)D.nb;
    if ((int64_t)gl_num_tiles(D.n) * GL_TILE_ELEMS * 8 >= (int64_t)1 << 32) return why(1);  // (32-bit byte offsets in the streamed pass)
    if (nb > 192) return why(7);  // (an owner's wave reads the block sums of gamma u three per lane)
    // strips: a narrow first one (nb mod 8 columns, at the short end of the lattice), then 8 columns each
    std::vector<int> sc0, sc1;
    for (int c = 0; c < nb;) {
        const int wdt = (c == 0 && nb % TW != 0) ? nb % TW : TW;
        sc0.push_back(c), sc1.push_back(c + wdt);
        c += wdt;
    }
    const int nstrips = (int)sc0.size();
    struct Rect {
        int t, I0, I1;
    };
    std::vector<Rect> rects;
    std::vector<std::pair<int, int>> extras;  // split extra tiles (I, J)
    // "Split" decomposition, for lattices just past the register files: ONE block row per rectangle (every wave's tile is
    // resident), and the rectangles the chip has no workgroup for -- the ones with the fewest tiles, along the diagonal --
    // are dissolved: their tiles become split extras, streamed by all eight waves of a workgroup together (k_tile).  With
    // whole streamed rounds instead, the few workgroups that get a second block row stream eight tiles each and the chip
    // waits for them: 21 us per iteration at n = 4160 against 7 at n = 4096.
    if (allow_split && !getenv("GRAPHLEARN_AMD_TILE_NO_SPLIT")) {
        std::vector<std::pair<int, Rect>> all;  // (tiles, rectangle)
        for (int t = 0; t < nstrips; ++t)
            for (int I = 0; I < sc1[(size_t)t]; ++I) all.push_back({sc1[(size_t)t] - std::max(sc0[(size_t)t], I), Rect{t, I, I + 1}});
        const int over = (int)all.size() - ncu;
        if (over > 0 && ncu >= nb) {
            std::stable_sort(all.begin(), all.end(), [](const std::pair<int, Rect> &a, const std::pair<int, Rect> &b) { return a.first < b.first; });
            int64_t nex = 0;
            for (int q = 0; q < over; ++q) nex += all[(size_t)q].first;
            for (size_t q = (size_t)over; q < all.size(); ++q)  // (the diagonal tiles of the rectangles that stay are split too)
                nex += all[q].second.I0 >= sc0[(size_t)all[q].second.t] ? 1 : 0;
            if (nex <= (int64_t)XMAX * ncu * 3 / 4) {  // (a quarter to spare: the node-block lists do not always allow the densest packing)
                for (int q = 0; q < over; ++q) {
                    const Rect &rc = all[(size_t)q].second;
                    for (int J = std::max(sc0[(size_t)rc.t], rc.I0); J < sc1[(size_t)rc.t]; ++J) extras.push_back({rc.I0, J});
                }
                for (size_t q = (size_t)over; q < all.size(); ++q) rects.push_back(all[q].second);
                std::sort(rects.begin(), rects.end(), [](const Rect &a, const Rect &b) { return a.t != b.t ? a.t < b.t : a.I0 < b.I0; });
            }
        }
    }
    const bool split = !rects.empty();
    int R = 0, NWG = 0;
    if (split) R = 1, NWG = (int)rects.size();
    for (int r = 1; rects.empty() && r <= NBW - TW && r <= 8; ++r) {
        int cnt = 0;
        for (int t = 0; t < nstrips; ++t) cnt += (sc1[(size_t)t] + r - 1) / r;
        if (cnt <= ncu) {
            R = r, NWG = cnt;
            break;
        }
    }
    if (R == 0) return why(2);  // too large for one rectangle per workgroup: the streaming engine serves it
    // groups per strip; the workgroups left over go, one at a time, to the strip with the most rows per group (more
    // tiles in registers, fewer streamed)
    std::vector<int> grp((size_t)nstrips);
    for (int t = 0; t < nstrips; ++t) grp[(size_t)t] = (sc1[(size_t)t] + R - 1) / R;
    if (rects.empty() && !getenv("GRAPHLEARN_AMD_TILE_NO_SPREAD"))
        for (; NWG < ncu; ++NWG) {
            int best = -1;
            double most = 1.0;
            for (int t = 0; t < nstrips; ++t) {
                const double per = (double)sc1[(size_t)t] / grp[(size_t)t] * (sc1[(size_t)t] - sc0[(size_t)t]) / TW;
                if (grp[(size_t)t] < sc1[(size_t)t] && per > most) most = per, best = t;
            }
            if (best < 0) break;
            ++grp[(size_t)best];
        }
    NWG = std::max(NWG, nb);  // every node block needs an owner
    if (NWG > ncu || NWG > 64 * nb) return why(3);
    std::vector<WaveDesc> wd((size_t)NWG * TW);
    std::vector<WgDesc> gd((size_t)NWG);
    for (WaveDesc &w : wd) {
        memset(&w, 0, sizeof w);
        w.tile = -1;
        w.fB = -1;
    }
    for (WgDesc &G : gd) memset(&G, 0, sizeof G);
    if (rects.empty())  // whole rounds: strip t cut into grp[t] groups of block rows
        for (int t = 0; t < nstrips; ++t) {
            const int h = sc1[(size_t)t], ng = grp[(size_t)t];
            for (int q = 0; q < ng; ++q) rects.push_back(Rect{t, (int)((int64_t)q * h / ng), (int)((int64_t)(q + 1) * h / ng)});
        }
    if ((int)rects.size() > NWG) return why(8);
    {
        for (int g = 0; g < (int)rects.size(); ++g) {
            const int t = rects[(size_t)g].t, I0 = rects[(size_t)g].I0, I1 = rects[(size_t)g].I1;  // block rows [I0, I1)
            {
            WgDesc &G = gd[(size_t)g];
            auto slot_of = [&](int B) -> int {
                for (int k = 0; k < G.nblk; ++k)
                    if (G.blk[k] == B) return k;
                if (G.nblk >= NBW) return -1;
                G.blk[G.nblk] = B;
                return G.nblk++;
            };
            for (int v = 0; v < TW; ++v) {
                const int J = sc0[(size_t)t] + v;
                if (J >= sc1[(size_t)t] || J < I0) continue;  // (below the diagonal: nothing)
                if (split && J == I0) {  // a diagonal tile is never one wave's streamed tile here (eight ring turns: 15 us)
                    extras.push_back({I0, J});
                    continue;
                }
                WaveDesc &w = wd[(size_t)(g * TW + v)];
                w.I = I0;
                w.J = J;
                w.nr = std::min(I1 - I0, J - I0 + 1);
                if (w.nr > 8) return why(4);
                if (I0 < J) w.tile = (int)gl_tile_id(I0, J, nb);  // round 0 from the registers
                w.sj = slot_of(J);
                if (w.sj < 0) return why(5);
                G.colmask[w.sj] |= 1 << v;
                for (int r = 0; r < w.nr; ++r) {
                    const int sl = slot_of(I0 + r);
                    if (sl < 0) return why(6);
                    if (r == 0) w.si = sl;
                    w.rs |= sl << (4 * r);
                    (r < 4 ? G.rowmask[sl] : G.rowmask2[sl]) |= (int)(1u << (8 * (r & 3) + v));
                    G.nrnd = std::max(G.nrnd, r + 1);
                }
            }
            }
        }
    }
    // split extras: each goes to the workgroup with the fewest so far that has room for its node blocks (the ones that know
    // both blocks already first)
    for (const std::pair<int, int> &ex : extras) {
        int best = -1, best_cost = 1 << 30;
        for (int q = 0; q < (int)rects.size(); ++q) {
            const WgDesc &G = gd[(size_t)q];
            if (G.nx >= XMAX) continue;
            int add = 2 - (ex.first == ex.second ? 1 : 0);
            for (int k = 0; k < G.nblk; ++k) add -= (G.blk[k] == ex.first) + (G.blk[k] == ex.second && ex.first != ex.second);
            if (G.nblk + add > NBW) continue;
            const int cost = G.nx * 4 + add;
            if (cost < best_cost) best_cost = cost, best = q;
        }
        if (best < 0) return tile_prepare_rounds(s, false);  // (no room in the block lists: whole rounds instead)
        WgDesc &G = gd[(size_t)best];
        auto slot_of = [&](int B) -> int {
            for (int k = 0; k < G.nblk; ++k)
                if (G.blk[k] == B) return k;
            G.blk[G.nblk] = B;
            return G.nblk++;
        };
        const int si = slot_of(ex.first), sj = slot_of(ex.second);
        G.xt[G.nx] = (int)gl_tile_id(ex.first, ex.second, nb);
        G.xs[G.nx] = si | (sj << 8) | (ex.first == ex.second ? 1 << 16 : 0);
        ++G.nx;
    }
    for (int q = 0; q < NWG; ++q) {
        gd[(size_t)q].own = q < nb ? q : -1;
        gd[(size_t)q].active = (gd[(size_t)q].nblk > 0 || gd[(size_t)q].own >= 0) ? 1 : 0;
    }
    return tile_finish(s, wd, gd, NWG, SLOTS, true);
}

// Builds the static work decomposition.  GL_OK with s->tile == nullptr when the engine does not apply.
int gl_tile_prepare(gl_logdeg *s) {
    gl_ctx *ctx = s->ctx;
    s->tile = nullptr;
    bool forced = false;
    if (const char *e = getenv("GRAPHLEARN_AMD_ENGINE")) {
        if (!strcmp(e, "stream") || !strcmp(e, "resident1")) return GL_OK;
        forced = !strcmp(e, "resident") || !strcmp(e, "tile");
    }
    const LogdegDev &D = s->dev;
    const int ncu = ctx->prop.multiProcessorCount;
    const int nb = (int)D.nb;
    if (nb > 64 || nb * (nb - 1) / 2 > TW * ncu) return tile_prepare_rounds(s);  // does not fit the register files
    std::vector<WaveDesc> wd;
    std::vector<WgDesc> gd;
    int NWG = 0, U = SLOTS;
    (void)tile_plan_fit(nb, ncu, forced, wd, gd, NWG, U, getenv("GRAPHLEARN_AMD_TILE_U") ? atoi(getenv("GRAPHLEARN_AMD_TILE_U")) : 0,
                        getenv("GRAPHLEARN_AMD_TILE_NO_EXTRA_WG") == nullptr);
    if (NWG == 0) return GL_OK;
    return tile_finish(s, wd, gd, NWG, U, false);
}

// A batch of B graphs of one size (s: a session that stands for all of them -- dev.n of ONE graph, dev.nb / dev.ntiles and
// every buffer B-fold, graph after graph): the decomposition of one graph on a B-th of the chip, replicated.  GL_OK with
// s->tile == nullptr when it does not apply.
int gl_tile_prepare_batch(gl_logdeg *s, int B) {
    gl_ctx *ctx = s->ctx;
    s->tile = nullptr;
    if (const char *e = getenv("GRAPHLEARN_AMD_ENGINE"))
        if (!strcmp(e, "stream") || !strcmp(e, "resident1")) return GL_OK;
    const int ncu = ctx->prop.multiProcessorCount;
    const int nb1 = (int)(s->dev.nb / B);
    const int64_t ntiles1 = s->dev.ntiles / B;
    if (B < 1 || nb1 * B != s->dev.nb || ncu / B < nb1) return GL_OK;
    std::vector<WaveDesc> wd1;
    std::vector<WgDesc> gd1;
    int NWG1 = 0, U = SLOTS;
    (void)tile_plan_fit(nb1, ncu / B, false, wd1, gd1, NWG1, U, getenv("GRAPHLEARN_AMD_TILE_U") ? atoi(getenv("GRAPHLEARN_AMD_TILE_U")) : 0,
                        getenv("GRAPHLEARN_AMD_TILE_NO_EXTRA_WG") == nullptr);
    if (NWG1 == 0 || NWG1 > 64 * nb1) return GL_OK;
    std::vector<WaveDesc> wd((size_t)B * NWG1 * TW);
    std::vector<WgDesc> gd((size_t)B * NWG1);
    for (int g = 0; g < B; ++g)
        for (int q = 0; q < NWG1; ++q) {
            WgDesc G = gd1[(size_t)q];
            for (int k = 0; k < G.nblk; ++k) G.blk[k] += g * nb1;
            if (G.own >= 0) G.own += g * nb1;
            gd[(size_t)g * NWG1 + q] = G;
            for (int v = 0; v < TW; ++v) {
                WaveDesc w = wd1[(size_t)q * TW + v];
                if (w.tile >= 0) w.tile += (int)(g * ntiles1), w.I += g * nb1, w.J += g * nb1;
                if (w.fB >= 0) w.fB += g * nb1;
                wd[((size_t)g * NWG1 + q) * TW + v] = w;
            }
        }
    return tile_finish(s, wd, gd, B * NWG1, U, false, nb1, NWG1);
}

// the launches from now on write every iterate to `ring` (slots of ring_stride doubles) and its dual variable to `dring`
// (null: stop doing so)
void gl_tile_batch_set_ring(gl_logdeg *s, double *ring, int64_t ring_stride, double *dring, int64_t dring_stride) {
    gl_tile *t = s->tile;
    t->ring = ring, t->dring = dring;
    t->ring_stride = ring_stride, t->dring_stride = dring_stride;
}

// 2 gamma z of every graph of the batch in the batch's own tiled buffer (devs: the sessions' device views, on the device)
int gl_tile_batch_make_zg(gl_logdeg *s, const LogdegDev *devs_dev, int B) {
    gl_tile *t = s->tile;
    k_make_zg_batch<<<dim3((unsigned)t->ntiles1, (unsigned)B), dim3(256), 0, s->ctx->stream>>>(devs_dev, t->zg, s->dev.n, t->nb1, t->ntiles1,
                                                                                          s->dev.g2);
    GL_HIP(s->ctx, hipGetLastError());
    t->zg_version = s->init_version;
    return GL_OK;
}

bool gl_tile_writes_final_parity(const gl_logdeg *s) { return s->tile != nullptr && !s->tile->big; }

// Runs up to `iters` iterations in ONE launch (asynchronous).  final_parity < 0: the iterate stays in w[it0 & 1] and the
// caller fixes the ping-pong parity afterwards; 0 / 1 (fixed-iteration calls below the rounds): it is written to that
// buffer directly.  norms = false (only with tol = 0): the ratios of the stopping rule are not formed at all (dstat
// keeps its previous values).
// gate > 0 (with norms = false and tol > 0): the variant without the norms runs while the dual ratio alone rules the stopping
// rule out and ends (ctr[0] = 2) when it comes within gate x tol (TileDev::gate).
int gl_tile_launch(gl_logdeg *s, int64_t iters, double tol, bool norms, int final_parity, double gate) {
    gl_ctx *ctx = s->ctx;
    gl_tile *t = s->tile;
    TileDev P = t->dev;
    const LogdegDev &D = s->dev;
    if (t->batch && t->zg_version != s->init_version) return gl_set_error(ctx, GL_E_STATE, "internal: the batch's distances were not set");
    if (t->zg_version != s->init_version) {  // first launch since gl_logdeg_init: the distances or gamma may be new
        k_make_zg<<<dim3((unsigned)D.ntiles), dim3(256), 0, ctx->stream>>>(D.z, t->zg, D.n, D.nb, D.g2);
        GL_HIP(ctx, hipGetLastError());
        t->zg_version = s->init_version;
    }
    P.w = D.w[s->it & 1];
    // where the registers go at the end: a fixed-iteration call knows the parity of its last iterate now and writes it where
    // everybody expects it; a call that may stop early (and the rounds, which update their streamed tiles in place) keeps
    // the buffer, and gl_resident_finish copies if the parity came out odd
    P.wout = (!t->big && tol <= 0 && final_parity >= 0) ? D.w[final_parity] : P.w;
    P.gamma = D.gamma;
    P.c = D.c;
    P.omc = 1.0 - D.c;
    P.nm2 = (double)(D.n - 2);
    P.a4g = D.a4g;
    P.tol = tol;
    P.nb1 = t->nb1;
    P.NWG1 = t->NWG1;
    P.ntiles1 = t->ntiles1;
    P.t2lo = tol * tol * (1 + 1e-9);
    P.ring = t->batch ? t->ring : nullptr;
    P.dring = t->batch ? t->dring : nullptr;
    P.ring_stride = t->ring_stride;
    P.dring_stride = t->dring_stride;
    if ((tol > 0 && !(gate > 0 && !norms)) || t->batch) norms = true;
    P.gate = (!norms && tol > 0) ? gate * tol : 0.0;
    P.it0 = s->it;
    P.maxit = (int)std::min<int64_t>(iters, 1 << 30);
    if (t->next_tag == 0u || t->next_tag > 0xE0000000u - (unsigned)P.maxit) {  // the tags would wrap into old ones
        GL_HIP(ctx, hipMemsetAsync((char *)t->arena + t->gran_off, 0, t->gran_bytes, ctx->stream));
        t->next_tag = 1u;
    }
    P.tag0 = t->next_tag;
    t->next_tag += (unsigned)P.maxit + 2u;
    if (t->arrivals > 0xF0000000u - (unsigned)P.NWG) {  // (after ~16 million launches)
        GL_HIP(ctx, hipMemsetAsync(P.bar, 0, GL_BAR_WORDS * 4, ctx->stream));
        t->arrivals = 0;
    }
    t->arrivals += (unsigned)P.NWG;
    P.check_rendezvous = getenv("GRAPHLEARN_AMD_RES_TEST_NOT_RESIDENT") ? 1 : 0;
    P.rendezvous = t->arrivals + (P.check_rendezvous ? 1u : 0u);
    const bool ragged = D.n % TB != 0;
    const dim3 grid((unsigned)P.NWG), block(TT);
    auto go = [&](auto kern) { kern<<<grid, block, 0, ctx->stream>>>(P); };
    // Without the norms the pass itself does not depend on raggedness (the mask is applied at the write-back only), so ONE
    // instantiation serves every lattice: the RAGGED one (whose factors are 1 everywhere on a lattice that is not).
#define GL_TILE_GO(UU, BB)                                                                       \
    do {                                                                                         \
        if (!norms) go(k_tile<UU, true, false, BB>);                                             \
        else if (ragged) go(k_tile<UU, true, true, BB>);                                         \
        else go(k_tile<UU, false, true, BB>);                                                    \
    } while (0)
#define GL_TILE_GO_BATCH(UU)                                                                     \
    do {                                                                                         \
        if (ragged) go(k_tile<UU, true, true, 0, true>);                                         \
        else go(k_tile<UU, false, true, 0, true>);                                               \
    } while (0)
    if (t->batch) {
        if (t->U == 32) GL_TILE_GO_BATCH(32);
        else if (t->U == 16) GL_TILE_GO_BATCH(16);
        else GL_TILE_GO_BATCH(8);
    } else if (t->big && t->nt)
        GL_TILE_GO(32, 2);
    else if (t->big)
        GL_TILE_GO(32, 1);
    else if (t->U == 32)
        GL_TILE_GO(32, 0);
    else if (t->U == 16)
        GL_TILE_GO(16, 0);
    else
        GL_TILE_GO(8, 0);
#undef GL_TILE_GO
#undef GL_TILE_GO_BATCH
    GL_HIP(ctx, hipGetLastError());
    return GL_OK;
}

// development: phase stamps of iterations 8..23 of the last launch (GRAPHLEARN_AMD_RES_STAMPS=1), after a sync
void gl_tile_dump_stamps(gl_logdeg *s) {
    if (!s || !s->tile || !s->tile->dbg) return;
    const int NWG = s->tile->dev.NWG, nb = (int)s->tile->dev.nb;
    std::vector<unsigned long long> h((size_t)NWG * (16 * (6 + 2 * TW) + 8 * TW + 16 * 4 * TW));
    (void)hipMemcpy(h.data(), s->tile->dbg, h.size() * 8, hipMemcpyDeviceToHost);
    if (const char *raw = getenv("GRAPHLEARN_AMD_RES_STAMPS_RAW")) {  // the stamps themselves, for tools/stamps_raw.py
        if (FILE *f = fopen(raw, "ab")) {
            const unsigned long long hdr[4] = {0x544c5354414d5053ull, (unsigned long long)NWG, (unsigned long long)nb, (unsigned long long)TW};
            fwrite(hdr, 8, 4, f);
            fwrite(h.data(), 8, h.size(), f);
            fclose(f);
        }
    }
    {  // the call as a whole (100 MHz ticks -> us), relative to the earliest entry of any wave
        const unsigned long long *cw = h.data() + (size_t)NWG * 16 * (6 + 2 * TW);
        unsigned long long t0 = ~0ull;
        for (int v = 0; v < NWG * TW; ++v)
            if (cw[(size_t)v * 8]) t0 = std::min(t0, cw[(size_t)v * 8]);
        double mean[6] = {0}, last[6] = {0};
        long cnt = 0;
        for (int v = 0; v < NWG * TW; ++v) {
            const unsigned long long *o = cw + (size_t)v * 8;
            if (!o[0] || !o[5]) continue;
            ++cnt;
            for (int q = 0; q < 6; ++q) {
                if (!o[q]) continue;  // (stamps 1, 2 are not taken: a store between the bulk loads and the first pass would cost it its counted waits)
                const double d = (double)(o[q] - t0) / 100.0;
                mean[q] += d, last[q] = std::max(last[q], d);
            }
        }
        if (cnt)
            fprintf(stderr, "[tile call stamps] us after the first wave's entry, mean / latest wave: entry %.1f / %.1f  first iteration done %.1f / %.1f  "
                            "loop done %.1f / %.1f  stored %.1f / %.1f\n",
                    mean[0] / cnt, last[0], mean[3] / cnt, last[3], mean[4] / cnt, last[4], mean[5] / cnt, last[5]);
    }
    {  // per-wave pass durations: mean, and mean over workgroups of (slowest - fastest wave of the workgroup)
        const unsigned long long *pw = h.data() + (size_t)NWG * 16 * 6;
        double sum = 0, skew = 0, mx = 0, first_last = 0;
        long cnt = 0, wcnt = 0;
        double per_wave[TW] = {0};
        long per_cnt[TW] = {0};
        for (int w = 0; w < NWG; ++w)
            for (int k = 1; k < 16; ++k) {
                double lo = 1e30, hi = 0;
                unsigned long long e_first = ~0ull, e_last = 0;
                for (int v = 0; v < TW; ++v) {
                    const unsigned long long *o = pw + (((size_t)w * 16 + k) * TW + v) * 2;
                    if (!o[1]) continue;
                    const double d = (double)(o[1] - o[0]);
                    sum += d, ++cnt;
                    per_wave[v] += d, ++per_cnt[v];
                    lo = std::min(lo, d), hi = std::max(hi, d);
                    e_first = std::min(e_first, o[1]), e_last = std::max(e_last, o[1]);
                    mx = std::max(mx, d);
                }
                if (hi > 0) skew += hi - lo, first_last += (double)(e_last - e_first), ++wcnt;
            }
        if (cnt) {
            fprintf(stderr, "[tile stamps] pass per wave: mean %.2f us, slowest %.2f; inside a workgroup slowest - fastest %.2f, last end - first end %.2f; by wave:",
                    sum / cnt / 100, mx / 100, skew / wcnt / 100, first_last / wcnt / 100);
            for (int v = 0; v < TW; ++v) fprintf(stderr, " %.2f", per_cnt[v] ? per_wave[v] / per_cnt[v] / 100 : 0.0);
            fprintf(stderr, "\n");
        }
    }
    // per iteration: the latest end of an edge phase, the latest publish, the latest owner's node values out,
    // relative to the earliest start of that iteration's acquire
    double acq = 0, edge = 0, pub = 0, own_wait = 0, own_upd = 0, span = 0, edge_max = 0, acq_min = 0;
    int kc = 0;
    for (int k = 1; k < 16; ++k) {
        unsigned long long t_first = ~0ull, t_next = ~0ull;
        double e_max = 0, a_min = 1e30, a_sum = 0, e_sum = 0, p_sum = 0, w_sum = 0, u_sum = 0;
        int cnt = 0, ocnt = 0;
        for (int w = 0; w < NWG; ++w) {
            const unsigned long long *o = &h[((size_t)w * 16 + k) * 6], *pv = &h[((size_t)w * 16 + k - 1) * 6];
            if (!o[5] || !pv[5]) continue;
            t_first = std::min(t_first, pv[0]);
            t_next = std::min(t_next, o[0]);
            a_sum += (double)(o[1] - o[0]);
            a_min = std::min(a_min, (double)(o[1] - o[0]));
            e_sum += (double)(o[2] - o[1]);
            e_max = std::max(e_max, (double)(o[2] - o[1]));
            p_sum += (double)(o[3] - o[2]);
            ++cnt;
            if (w < nb && o[4]) {
                w_sum += (double)(o[4] - o[3]);
                u_sum += (double)(o[5] - o[4]);
                ++ocnt;
            }
        }
        if (!cnt) continue;
        acq += a_sum / cnt, edge += e_sum / cnt, pub += p_sum / cnt, edge_max += e_max, acq_min += a_min;
        if (ocnt) own_wait += w_sum / ocnt, own_upd += u_sum / ocnt;
        span += (double)(t_next - t_first);
        ++kc;
    }
    if (kc)
        fprintf(stderr, "[tile stamps] us per iteration %.2f | mean: acquire (wait for node values) %.2f (fastest %.2f)  edge %.2f (slowest %.2f)  "
                        "publish %.2f  owner: wait for records %.2f  update + publish %.2f\n",
                span / kc / 100, acq / kc / 100, acq_min / kc / 100, edge / kc / 100, edge_max / kc / 100, pub / kc / 100, own_wait / kc / 100,
                own_upd / kc / 100);
}

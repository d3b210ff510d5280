// Resident engine of the log-degree loop: ALL iterations in one persistent launch, with the
// edge iterate w held in REGISTERS across iterations.
//
// Reference path: graph_learn/smooth_learning.py:146-167 (same arithmetic as gl_logdeg.hip, which
// stays the general engine).  Why: per iteration the streaming engine moves 24 B per edge (read w,
// read z, write w).  At n <= 4096 the whole iterate fits the register files of the chip
// and LDS of the chip (64 MiB of 128 MiB + 40 MiB), so only z (8 B per edge) has to stream; the price is that the
// two grid-wide dependencies of an iteration (degree sums -> dual update -> next edge pass) become
// in-kernel grid barriers instead of kernel boundaries.
//
// Geometry: one workgroup of 8 waves per CU (it owns 128 KiB of the CU's LDS, so exactly one
// workgroup is resident per CU); wave slot s = 8*workgroup + wave owns
//   - one "bulk" tile (64x64, off-diagonal; columns beyond n masked): half of its w lives in 64 VGPRs per
//     lane, the other half in a wave-private 16 KiB slice of LDS (one ds_read/ds_write_b128 per
//     row pair and iteration -- never in HBM),
//   - a share of the "fringe" (the diagonal tiles, with masked entries), split at row-pair
//     granularity over all slots and streamed from memory every iteration.
// Per iteration: edge phase (all slots) -> grid barrier -> node phase (workgroup b updates node
// block b) -> grid barrier -> stopping rule (every workgroup, same data, same order).
//
// Inter-workgroup hand-offs follow the write-through recipe of the CDNA4 guide (Guideline 16,
// R1): every handed-off word is stored with an agent-scope relaxed atomic store (global_store sc1),
// every storing wave drains (s_waitcnt vmcnt(0)) before the workgroup barrier, ONE lane per
// workgroup adds to a monotonic counter, the consumer polls that counter with sc1 loads and then
// reads the handed-off words with agent-scope relaxed atomic loads (sc1, never through L1 or the
// scalar cache).  One workgroup per CU; spins are bounded (wall-clock) and a timeout raises an
// error word instead of hanging the device.
#include <algorithm>
#include <vector>

#include "gl_gridbar.h"
#include "gl_lanes.h"
#include "gl_logdeg.h"

#ifndef GL_RES_MIN_BULK_TILES
#define GL_RES_MIN_BULK_TILES 600
#endif

namespace {

constexpr int TB = GL_TB;
constexpr int RW = 8;             // waves per workgroup (two per SIMD, 256 registers each)
constexpr int RS = 16;            // row pairs of the wave's tile held in registers; the other 16 live in LDS
constexpr int RT = RW * 64;       // threads per workgroup
constexpr int SLOTS = TB / 2;     // 32 row pairs per tile

typedef unsigned long long u64;

__device__ __forceinline__ double ld_sc1(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load((const u64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_sc1(double *p, double v) {
    __hip_atomic_store((u64 *)p, (u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct ResDev {
    double *w;              // the buffer that holds the iterate on entry and on exit (tiled layout)
    const double *z;
    double *dvar, *ybar, *pbar, *dprev;
    double *rowslab_p, *rowslab_w;  // [J][node]  (same meaning as in the streaming engine)
    double *colslab;        // entries of 2 x 64: bulk tile f -> entry f; fringe (slot, seg) -> F + slot*KF + seg
    const int *bulk_tile;   // bulk slot -> tile id
    const int *fr_tile;     // fringe tile list
    const int *col_ptr;     // column block J -> its colslab rows [col_ptr[J], col_ptr[J+1]) (contiguous)
    const int *col_ent;     // producer entry id (bulk slot, or F + slot*KF + segment) -> colslab row
    double *tnorm;          // per slot {sum (q-y)^2, sum w^2}
    double *bnorm;          // the streaming engine's [2][nb][4] buffer: iteration `it` uses half it & 1
    double *dstat;          // {rel_primal, rel_dual, tol}
    long long *ctr;         // {done, final_iter, it_edge, it_node}
    unsigned *bar;          // barrier state (layout: see grid_barrier)
    unsigned long long *dbg;  // development only (GRAPHLEARN_AMD_RES_STAMPS=1): per-workgroup phase stamps
    int64_t n, nb, n_pad;
    int F, NF, S, NWG, KF;
    int64_t R;              // fringe row pairs
    double gamma, c, g2, a4g, tol;
    long long it0;          // iterations completed before this launch
    int rendezvous;         // workgroups expected at the start-up rendezvous (= NWG; NWG + 1 in the fallback test)
    int maxit;
};

// Edge phase of one resident bulk tile (no masked entries): 32 row pairs -- the first RS in
// registers, the rest in the wave's LDS slice -- the z ring 8 slots ahead, row sums by half-wave
// butterflies per group of 8, column sums in registers.
// The empty asm after every slot pins the running sums: without it LLVM sinks the six reduction
// chains to the end of the unrolled body and spills every addend on the way.
struct NodeVals {
    double dj0, dj1, pj0, pj1;  // dvar / pbar of this lane's two columns
    double dvec, pvec;          // dvar / pbar of row `lane` of the tile
};

__device__ __forceinline__ NodeVals load_node_vals(const ResDev &P, int I, int J, int lane) {
    int oj = J * TB + 2 * (lane & 31), oi = I * TB + lane;
    asm volatile("" : "+v"(oj), "+v"(oi));  // recompute the six addresses each iteration rather than spill them
    NodeVals v;
    v.dj0 = ld_sc1(P.dvar + oj);
    v.dj1 = ld_sc1(P.dvar + oj + 1);
    v.pj0 = ld_sc1(P.pbar + oj);
    v.pj1 = ld_sc1(P.pbar + oj + 1);
    v.dvec = ld_sc1(P.dvar + oi);
    v.pvec = ld_sc1(P.pbar + oi);
    return v;
}

__device__ __forceinline__ double2 ld_z(const double *base, unsigned byte_off) {
    return *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(base) + byte_off);
}

// MASKED (lattices whose last block column is ragged): columns beyond n are kept at zero and out of every
// sum by a 0 / 1 factor per lane column (x 1.0 is exact, so full tiles compute bit-identical values; one
// code path, no select chains).  Lattices with n a multiple of 64 run the unmasked instantiation.
template <bool MASKED>
__device__ __forceinline__ void bulk_tile_pass(const ResDev &P, double2 (&wreg)[RS], double2 *wl, double2 (&zr)[8],
                                               const NodeVals &nv, const int I, const int J, const unsigned zb_in,
                                               const int entry, const int lane, double &nqy, double &nw) {
    const int half = lane >> 5, c0 = 2 * (lane & 31);
    const double c = P.c, g = P.gamma, g2 = P.g2;
    const double dj0 = nv.dj0, dj1 = nv.dj1, pj0 = nv.pj0, pj1 = nv.pj1, dvec = nv.dvec, pvec = nv.pvec;
    unsigned zb = zb_in;
    int rbase = J * (int)P.n_pad + I * TB + 2 * ((lane >> 2) & 7) + half;  // this lane's row-slab entry of group 0
    asm volatile("" : "+v"(zb), "+v"(rbase));  // not loop-invariant for the compiler: addresses are recomputed, not spilled
    const double m0 = (!MASKED || J * TB + c0 < P.n) ? 1.0 : 0.0, m1 = (!MASKED || J * TB + c0 + 1 < P.n) ? 1.0 : 0.0;
    double colp0 = 0, colp1 = 0, colw0 = 0, colw1 = 0;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        double rowp[8], roww[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int s = 8 * gq + k;
            // (two v_readlane + select per entry instead of the bpermute were measured slower: the phase is
            // VALU-bound, the bpermute runs on the LDS pipe)
            const double di = __shfl(dvec, 2 * s + half, 64);
            const double pi = __shfl(pvec, 2 * s + half, 64);
            const double2 wv = s < RS ? wreg[s < RS ? s : 0] : wl[(s - RS) * 64];
            double wn[2], pp[2], nq2[2], nw2[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const double w = e ? wv.y : wv.x;
                const double z = e ? zr[k].y : zr[k].x;
                const double y = c * w - g * (di + (e ? dj1 : dj0));    // :147
                const double tt = y - g2 * z;                           // :150
                const double pr = tt > 0 ? tt : 0.0;
                const double q = c * pr - g * (pi + (e ? pj1 : pj0));   // :153
                const double dq = q - y;                                // == -y + q (:158)
                if (MASKED) {
                    const double m = e ? m1 : m0;
                    wn[e] = m * ((w - y) + q);
                    pp[e] = m * pr;
                    nq2[e] = m * (dq * dq);
                } else {
                    wn[e] = (w - y) + q;                                // :163
                    pp[e] = pr;
                    nq2[e] = dq * dq;
                }
                nw2[e] = w * w;
            }
            nqy += nq2[0] + nq2[1];
            nw += nw2[0] + nw2[1];
            colp0 += pp[0];
            colp1 += pp[1];
            colw0 += wn[0];
            colw1 += wn[1];
            rowp[k] = pp[0] + pp[1];
            roww[k] = wn[0] + wn[1];
            if (s < RS)
                wreg[s < RS ? s : 0] = double2{wn[0], wn[1]};
            else
                wl[(s - RS) * 64] = double2{wn[0], wn[1]};
            // refill the ring: slot s + 8 of this tile (wraps into the next iteration)
            // uniform base + 32-bit lane offset: one address register per wave.  With 64-bit per-slot
            // pointers the compiler hoists 28 of them out of the iteration loop, spills them, and every
            // reload drains the prefetch ring (scratch loads share vmcnt with the z loads).
            zr[k] = ld_z(P.z, zb + (unsigned)(((s + 8) & (SLOTS - 1)) * (2 * TB) * 8));
            asm volatile("" : "+v"(nqy), "+v"(nw), "+v"(colp0), "+v"(colp1), "+v"(colw0), "+v"(colw1));
        }
        const double rw = halfwave_rows8(roww, lane);
        const double rp = halfwave_rows8(rowp, lane);
        if ((lane & 3) == 0) {
            const int64_t idx = (int64_t)rbase + 16 * gq;
            st_sc1(P.rowslab_p + idx, rp);
            st_sc1(P.rowslab_w + idx, rw);
        }
    }
    colp0 = add_xor32(colp0);
    colp1 = add_xor32(colp1);
    colw0 = add_xor32(colw0);
    colw1 = add_xor32(colw1);
    if (half == 0) {
        double *dst = P.colslab + (int64_t)entry * (2 * TB) + c0;  // entry = this tile's colslab row
        st_sc1(dst, colp0);
        st_sc1(dst + 1, colp1);
        st_sc1(dst + TB, colw0);
        st_sc1(dst + TB + 1, colw1);
    }
}

// The grid barrier (XCD-hierarchical, bounded spins) is in gl_gridbar.h.
template <bool MASKED>
__global__ __launch_bounds__(RT, 2) void k_resident(ResDev P) {
    __shared__ double2 s_w[RW][SLOTS - RS][64];  // 128 KiB: the LDS half of every wave's tile
    __shared__ double s_sum[2][RW][TB];
    __shared__ double s_red[2][RW];
    __shared__ double s_rel[2];
    __shared__ int s_ok;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wg = blockIdx.x;
    const int slot = wg * RW + wave;
    const int bulk_row = slot < P.F ? P.col_ent[slot] : 0;  // this slot's row of colslab
    const int half = lane >> 5, c0 = 2 * (lane & 31), lofs = half * TB + c0;
    const double c = P.c, g = P.gamma, g2 = P.g2;

    // ---- my bulk tile: w into registers (first RS row pairs) and LDS (the rest); z ring 8 slots
    //      ahead (z never changes, so the prefetch also runs across iterations and grid barriers)
    const bool has_bulk = slot < P.F;
    int I = 0, J = 0;
    int64_t tb = 0;
    unsigned zb = 0;  // byte offset of this lane's first z entry (the z array of a resident problem is < 4 GB)
    double2 wreg[RS], zr[8];
    double2 *wl = &s_w[wave][0][lane];
#pragma unroll
    for (int s = 0; s < RS; ++s) wreg[s] = double2{0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 8; ++k) zr[k] = double2{0.0, 0.0};
    if (has_bulk) {
        const int64_t t = P.bulk_tile[slot];
        gl_tile_decode(t, P.nb, I, J);
        I = __builtin_amdgcn_readfirstlane(I);
        J = __builtin_amdgcn_readfirstlane(J);
        tb = t * GL_TILE_ELEMS + lofs;
        zb = (unsigned)(tb * 8);
#pragma unroll
        for (int s = 0; s < RS; ++s) wreg[s] = *reinterpret_cast<const double2 *>(P.w + tb + s * (2 * TB));
#pragma unroll
        for (int s = RS; s < SLOTS; ++s) wl[(s - RS) * 64] = *reinterpret_cast<const double2 *>(P.w + tb + s * (2 * TB));
#pragma unroll
        for (int k = 0; k < 8; ++k) zr[k] = *reinterpret_cast<const double2 *>(P.z + tb + k * (2 * TB));
    }
    // ---- my share of the fringe (row pairs of diagonal / ragged tiles)
    const int64_t fr0 = (int64_t)slot * P.R / P.S, fr1 = (int64_t)(slot + 1) * P.R / P.S;

    const int my_e0 = wg < P.nb ? P.col_ptr[wg] : 0, my_e1 = wg < P.nb ? P.col_ptr[wg + 1] : 0;
    // tile ids of my first two fringe pairs and the colslab rows of my first two fringe segments: known
    // before the loop, so that the pair's data loads do not wait for index loads every iteration
    const int64_t ft0 = fr0 < fr1 ? P.fr_tile[fr0 >> 5] : -1, ft1 = fr0 + 1 < fr1 ? P.fr_tile[(fr0 + 1) >> 5] : -1;
    const int frow0 = fr0 < fr1 ? P.col_ent[P.F + slot * P.KF] : 0;
    const int frow1 = fr0 + 1 < fr1 && P.KF > 1 ? P.col_ent[P.F + slot * P.KF + 1] : 0;
    const GlGridBar gb{P.bar, P.rendezvous};
    GlBarState bs{};
    // an earlier launch of this session failed (workgroups not co-resident) and has not been looked at yet:
    // stay out of the way, the host redoes everything from where that launch stood
    if (P.ctr[4] != 0) return;
    if (tid == 0) s_ok = gl_barrier_setup(gb, bs) ? 1 : 0;
    __syncthreads();
    bool ok = s_ok != 0;
    const bool met = ok;
    int done_iters = 0;
    int stopped = 0;

    for (int iter = 0; ok && iter < P.maxit; ++iter) {
        double nqy = 0, nw = 0;
        unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
        const bool stamp = P.dbg != nullptr && iter >= 8 && iter < 24;
        if (stamp) ts[0] = __builtin_amdgcn_s_memrealtime();
        // ================= edge phase: my resident bulk tile =================
        if (has_bulk) bulk_tile_pass<MASKED>(P, wreg, wl, zr, load_node_vals(P, I, J, lane), I, J, zb, bulk_row, lane, nqy, nw);
        // ================= edge phase: my fringe row pairs (streamed, masked, in place) =================
        if (fr0 < fr1) {
            int seg = 0;
            int64_t tcur = -1;
            int fI = 0, fJ = 0;
            double dj0 = 0, dj1 = 0, pj0 = 0, pj1 = 0;
            double colp0 = 0, colp1 = 0, colw0 = 0, colw1 = 0;
            for (int64_t fp = fr0; fp < fr1; ++fp) {
                const int64_t t = fp == fr0 ? ft0 : (fp == fr0 + 1 ? ft1 : P.fr_tile[fp >> 5]);
                const int a = (int)(fp & (SLOTS - 1));
                if (t != tcur) {
                    if (tcur >= 0) {  // flush the previous tile's column partials
                        colp0 = add_xor32(colp0);
                        colp1 = add_xor32(colp1);
                        colw0 = add_xor32(colw0);
                        colw1 = add_xor32(colw1);
                        if (half == 0) {
                            const int row = seg == 0 ? frow0 : (seg == 1 ? frow1 : P.col_ent[P.F + slot * P.KF + seg]);
                            double *dst = P.colslab + (int64_t)row * (2 * TB) + c0;
                            st_sc1(dst, colp0);
                            st_sc1(dst + 1, colp1);
                            st_sc1(dst + TB, colw0);
                            st_sc1(dst + TB + 1, colw1);
                        }
                        colp0 = colp1 = colw0 = colw1 = 0.0;
                        ++seg;
                    }
                    tcur = t;
                    gl_tile_decode(t, P.nb, fI, fJ);
                    dj0 = ld_sc1(P.dvar + (int64_t)fJ * TB + c0);
                    dj1 = ld_sc1(P.dvar + (int64_t)fJ * TB + c0 + 1);
                    pj0 = ld_sc1(P.pbar + (int64_t)fJ * TB + c0);
                    pj1 = ld_sc1(P.pbar + (int64_t)fJ * TB + c0 + 1);
                }
                const int64_t off = t * GL_TILE_ELEMS + (int64_t)a * (2 * TB) + lofs;
                const double2 wl = *reinterpret_cast<const double2 *>(P.w + off);
                const double2 zl = *reinterpret_cast<const double2 *>(P.z + off);
                const int i = fI * TB + 2 * a + half;
                const double di = ld_sc1(P.dvar + i), pi = ld_sc1(P.pbar + i);
                double wn[2], pp[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int j = fJ * TB + c0 + e;
                    const bool valid = i < j && j < P.n;
                    const double w = e ? wl.y : wl.x, z = e ? zl.y : zl.x;
                    const double y = c * w - g * (di + (e ? dj1 : dj0));
                    const double tt = y - g2 * z;
                    const double pr = tt > 0 ? tt : 0.0;
                    const double q = c * pr - g * (pi + (e ? pj1 : pj0));
                    const double wnew = (w - y) + q;
                    const double dq = q - y;
                    pp[e] = valid ? pr : 0.0;
                    wn[e] = valid ? wnew : 0.0;
                    nqy += valid ? dq * dq : 0.0;
                    nw += valid ? w * w : 0.0;
                }
                colp0 += pp[0];
                colp1 += pp[1];
                colw0 += wn[0];
                colw1 += wn[1];
                *reinterpret_cast<double2 *>(P.w + off) = double2{wn[0], wn[1]};  // in place: this pair is mine
                const double rp = half_sum(pp[0] + pp[1]);
                const double rw = half_sum(wn[0] + wn[1]);
                if ((lane & 31) == 0) {
                    const int64_t idx = (int64_t)fJ * P.n_pad + i;
                    st_sc1(P.rowslab_p + idx, rp);
                    st_sc1(P.rowslab_w + idx, rw);
                }
            }
            colp0 = add_xor32(colp0);
            colp1 = add_xor32(colp1);
            colw0 = add_xor32(colw0);
            colw1 = add_xor32(colw1);
            if (half == 0) {
                const int row = seg == 0 ? frow0 : (seg == 1 ? frow1 : P.col_ent[P.F + slot * P.KF + seg]);
                double *dst = P.colslab + (int64_t)row * (2 * TB) + c0;
                st_sc1(dst, colp0);
                st_sc1(dst + 1, colp1);
                st_sc1(dst + TB, colw0);
                st_sc1(dst + TB + 1, colw1);
            }
        }
        nqy = wave_sum(nqy);
        nw = wave_sum(nw);
        if (lane == 0) {
            st_sc1(P.tnorm + 2 * slot, nqy);
            st_sc1(P.tnorm + 2 * slot + 1, nw);
        }

        // ================= all degree partials are out =================
        if (stamp) ts[1] = __builtin_amdgcn_s_memrealtime();
        if (!(ok = gl_grid_barrier(gb, bs, &s_ok))) break;
        if (stamp) ts[2] = __builtin_amdgcn_s_memrealtime();

        // ================= node phase: workgroup wg updates node blocks wg, wg + NWG, ... =================
        for (int64_t B = wg; B < P.nb; B += P.NWG) {
            const int64_t i = B * TB + lane;
            // every load of the block's update is issued before the first use: one round trip
            double rp[8], rw[8], cp[8], cw[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {  // row partials of tiles (B, Jb), Jb = B + wave + 8u
                const int64_t Jb = B + wave + (int64_t)u * RW;
                const bool on = Jb < P.nb;
                rp[u] = on ? ld_sc1(P.rowslab_p + Jb * P.n_pad + i) : 0.0;
                rw[u] = on ? ld_sc1(P.rowslab_w + Jb * P.n_pad + i) : 0.0;
            }
            // colslab range of the block: known before the loop for the workgroup's first block, so the
            // slab loads below do not wait for an index load
            const int e0 = B == wg ? my_e0 : P.col_ptr[B], e1 = B == wg ? my_e1 : P.col_ptr[B + 1];
#pragma unroll
            for (int u = 0; u < 8; ++u) {  // column partials of the tiles (Ib, B)
                const int e = e0 + wave + u * RW;
                const bool on = e < e1;
                cp[u] = on ? ld_sc1(P.colslab + (int64_t)e * (2 * TB) + lane) : 0.0;
                cw[u] = on ? ld_sc1(P.colslab + (int64_t)e * (2 * TB) + TB + lane) : 0.0;
            }
            double t0 = 0, t1 = 0;
            {
                const int64_t q = B + (int64_t)tid * P.nb;
                if (q < P.S) {
                    t0 = ld_sc1(P.tnorm + 2 * q);
                    t1 = ld_sc1(P.tnorm + 2 * q + 1);
                }
            }
            double d_old = 0, pb_old = 0, yb_old = 0;
            if (wave == 0) {
                d_old = ld_sc1(P.dvar + i);
                pb_old = ld_sc1(P.pbar + i);
                yb_old = P.ybar[i];
            }
            double sp = 0, sw = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                sp += rp[u];
                sw += rw[u];
            }
            for (int64_t Jb = B + wave + 8 * RW; Jb < P.nb; Jb += RW) {  // only beyond 64 column blocks
                sp += ld_sc1(P.rowslab_p + Jb * P.n_pad + i);
                sw += ld_sc1(P.rowslab_w + Jb * P.n_pad + i);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                sp += cp[u];
                sw += cw[u];
            }
            for (int e = e0 + wave + 8 * RW; e < e1; e += RW) {
                sp += ld_sc1(P.colslab + (int64_t)e * (2 * TB) + lane);
                sw += ld_sc1(P.colslab + (int64_t)e * (2 * TB) + TB + lane);
            }
            s_sum[0][wave][lane] = sp;
            s_sum[1][wave][lane] = sw;
            for (int64_t q = B + ((int64_t)tid + RT) * P.nb; q < P.S; q += (int64_t)RT * P.nb) {
                t0 += ld_sc1(P.tnorm + 2 * q);
                t1 += ld_sc1(P.tnorm + 2 * q + 1);
            }
            t0 = wave_sum(t0);
            t1 = wave_sum(t1);
            if (lane == 0) {
                s_red[0][wave] = t0;
                s_red[1][wave] = t1;
            }
            __syncthreads();
            if (wave == 0) {
                sp = 0;
                sw = 0;
#pragma unroll
                for (int v = 0; v < RW; ++v) {
                    sp += s_sum[0][v][lane];
                    sw += s_sum[1][v][lane];
                }
                const bool real = i < P.n;
                const double qb = pb_old + P.gamma * sp;   // :154
                const double dq = qb - yb_old;             // == -ybar + qbar (:159)
                double dq2 = real ? dq * dq : 0.0;
                double d2 = real ? d_old * d_old : 0.0;
                const double dnew = (d_old - yb_old) + qb;  // :164
                const double ybn = dnew + P.gamma * sw;     // :148 of the next iteration
                const double pbn = (ybn - sqrt(ybn * ybn + P.a4g)) / 2;  // :151
                P.dprev[i] = d_old;
                st_sc1(P.dvar + i, real ? dnew : 0.0);
                P.ybar[i] = real ? ybn : 0.0;
                st_sc1(P.pbar + i, real ? pbn : 0.0);
                dq2 = wave_sum(dq2);
                d2 = wave_sum(d2);
                if (lane == 0) {
                    double r0 = 0, r1 = 0;
#pragma unroll
                    for (int v = 0; v < RW; ++v) {
                        r0 += s_red[0][v];
                        r1 += s_red[1][v];
                    }
                    double *bn = P.bnorm + ((P.it0 + iter) & 1) * 4 * P.nb + 4 * B;
                    st_sc1(bn + 0, r0);
                    st_sc1(bn + 1, r1);
                    st_sc1(bn + 2, dq2);
                    st_sc1(bn + 3, d2);
                }
            }
            __syncthreads();
        }

        // ================= the new dual state is out =================
        if (stamp) ts[3] = __builtin_amdgcn_s_memrealtime();
        if (!(ok = gl_grid_barrier(gb, bs, &s_ok))) break;
        if (stamp) ts[4] = __builtin_amdgcn_s_memrealtime();

        done_iters = iter + 1;
        {  // stopping rule of this iteration, after its update (:156-167): same data, same order everywhere
            if (wave == 0) {
                const double *bn = P.bnorm + ((P.it0 + iter) & 1) * 4 * P.nb;
                double b0 = 0, b1 = 0, b2 = 0, b3 = 0;
                for (int64_t bb = lane; bb < P.nb; bb += 64) {
                    b0 += ld_sc1(bn + 4 * bb + 0);
                    b1 += ld_sc1(bn + 4 * bb + 1);
                    b2 += ld_sc1(bn + 4 * bb + 2);
                    b3 += ld_sc1(bn + 4 * bb + 3);
                }
                {  // the four sums in one halving butterfly (7 exchanges instead of 24): lane 8 q holds sum q
                    const double four[4] = {b0, b1, b2, b3};
                    const double c = add_xor32(halfwave_rows4(four, lane));
                    const long long cb = __double_as_longlong(c);
                    auto pick = [&](int l) {
                        const int lo = __builtin_amdgcn_readlane((int)cb, l), hi = __builtin_amdgcn_readlane((int)(cb >> 32), l);
                        return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
                    };
                    b0 = pick(0);
                    b1 = pick(8);
                    b2 = pick(16);
                    b3 = pick(24);
                }
                if (lane == 0) {
                    // four square roots and two divisions are ~0.35 us for this lone wave, on the critical path of every
                    // workgroup: they are taken only when they can matter -- the rule is `rp < tol && rd < tol`, and the
                    // host reads the two ratios after the last iteration of the call
                    const double t2 = P.tol * P.tol * (1 + 1e-12);
                    if ((b0 > t2 * b1 || b2 > t2 * b3) && iter + 1 < P.maxit) {
                        s_rel[0] = INFINITY;
                        s_rel[1] = INFINITY;
                    } else {
                        s_rel[0] = sqrt(b0) / sqrt(b1);
                        s_rel[1] = sqrt(b2) / sqrt(b3);
                    }
                }
            }
            __syncthreads();
            const double rp = s_rel[0], rd = s_rel[1];
            const long long itp = P.it0 + iter;
            if (wg == 0 && tid == 0 && itp >= 1) {
                P.dstat[0] = rp;
                P.dstat[1] = rd;
            }
            if (itp >= 1 && rp < P.tol && rd < P.tol) stopped = 1;  // never at iteration 0
            __syncthreads();  // s_rel is rewritten next iteration
        }
        if (stamp && tid == 0) {
            ts[5] = __builtin_amdgcn_s_memrealtime();
            unsigned long long *o = P.dbg + ((size_t)wg * 16 + (iter - 8)) * 6;
            for (int q = 0; q < 6; ++q) o[q] = ts[q];
        }
        if (stopped) break;
    }

    // ---- registers back to the iterate buffer, bookkeeping for the host / the streaming engine
    if (has_bulk) {
#pragma unroll
        for (int s = 0; s < RS; ++s) *reinterpret_cast<double2 *>(P.w + tb + s * (2 * TB)) = wreg[s];
#pragma unroll
        for (int s = RS; s < SLOTS; ++s) *reinterpret_cast<double2 *>(P.w + tb + s * (2 * TB)) = wl[(s - RS) * 64];
    }
    if (wg == 0 && tid == 0) {
        const long long itf = P.it0 + done_iters;
        P.ctr[0] = stopped;
        P.ctr[1] = itf;
        P.ctr[2] = itf;
        P.ctr[3] = itf > 0 ? itf - 1 : 0;
        P.ctr[4] = ok ? 0 : (met ? 2 : 1);  // 1: the workgroups never met (nothing modified); 2: a barrier failed in mid-run
    }
}

}  // namespace

struct gl_resident {
    ResDev dev{};
    int *bulk_tile = nullptr, *fr_tile = nullptr, *col_ptr = nullptr, *col_ent = nullptr;
    double *colslab = nullptr, *tnorm = nullptr;
    unsigned *bar = nullptr;
    unsigned long long *dbg = nullptr;
};

void gl_resident_free(gl_logdeg *s) {
    if (!s || !s->res) return;
    gl_resident *r = s->res;
    void *ptrs[] = {r->bulk_tile, r->fr_tile, r->col_ptr, r->col_ent, r->colslab, r->tnorm, r->bar, r->dbg};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete r;
    s->res = nullptr;
}

// Builds the static work decomposition.  Returns GL_OK with s->res == nullptr when the engine does
// not apply (problem too large for the register files, too small to pay off, or GRAPHLEARN_AMD_ENGINE=stream).
int gl_resident_prepare(gl_logdeg *s) {
    gl_ctx *ctx = s->ctx;
    s->res = nullptr;
    bool forced = false;
    if (const char *e = getenv("GRAPHLEARN_AMD_ENGINE")) {
        if (!strcmp(e, "stream") || !strcmp(e, "tile")) return GL_OK;
        forced = !strcmp(e, "resident") || !strcmp(e, "resident1");
    }
    const LogdegDev &D = s->dev;
    const int ncu = ctx->prop.multiProcessorCount;
    // bulk = off-diagonal tiles (the last block column may be ragged: masked in the pass); fringe = diagonal tiles
    std::vector<int> bulk, fringe;
    for (int64_t I = 0; I < D.nb; ++I)
        for (int64_t J = I; J < D.nb; ++J) {
            (I < J ? bulk : fringe).push_back((int)gl_tile_id(I, J, D.nb));
        }
    const int F = (int)bulk.size(), NF = (int)fringe.size();
    if (F > ncu * RW) return GL_OK;  // does not fit: the streaming engine handles it
    // One iteration of this engine costs ~20-24 us almost independently of n (two grid barriers and a
    // single wave per tile), the streaming engine ~1 us per 100 tiles on top of ~10: measured break-even
    // at n ~ 2150; it is chosen from 600 bulk tiles (n > 2240) on.  GRAPHLEARN_AMD_ENGINE=resident forces it wherever it fits.
    if (!forced && F < GL_RES_MIN_BULK_TILES) return GL_OK;
    int NWG = std::max(1, (F + RW - 1) / RW);
    NWG = std::max<int>(NWG, (int)std::min<int64_t>(D.nb, ncu));  // enough workgroups for the node phase / fringe
    NWG = std::min(NWG, ncu);
    const int S = NWG * RW;
    const int64_t R = (int64_t)NF * SLOTS;
    const int64_t maxfr = (R + S - 1) / S;
    const int KF = (int)((maxfr + SLOTS - 1) / SLOTS) + 1;

    // column index: bulk tile entries + fringe (slot, segment) entries per column block
    std::vector<std::vector<int>> lists((size_t)D.nb);
    for (int f = 0; f < F; ++f) {
        int I, J;
        gl_tile_decode(bulk[(size_t)f], D.nb, I, J);
        lists[(size_t)J].push_back(f);
    }
    for (int sl = 0; sl < S; ++sl) {
        const int64_t a = (int64_t)sl * R / S, b = (int64_t)(sl + 1) * R / S;
        int seg = 0;
        int64_t last = -1;
        for (int64_t fp = a; fp < b; ++fp) {
            const int64_t k = fp >> 5;
            if (k != last) {
                int I, J;
                gl_tile_decode(fringe[(size_t)k], D.nb, I, J);
                if (seg >= KF) return gl_set_error(ctx, GL_E_STATE, "internal: fringe segment overflow");
                lists[(size_t)J].push_back(F + sl * KF + seg);
                ++seg;
                last = k;
            }
        }
    }
    std::vector<int> ptr((size_t)D.nb + 1, 0), ent;
    for (int64_t J = 0; J < D.nb; ++J) {
        ptr[(size_t)J + 1] = ptr[(size_t)J] + (int)lists[(size_t)J].size();
        ent.insert(ent.end(), lists[(size_t)J].begin(), lists[(size_t)J].end());
    }
    std::vector<int> pos((size_t)F + (size_t)S * KF, 0);  // producer entry id -> colslab row (grouped by column block)
    for (size_t k = 0; k < ent.size(); ++k) pos[(size_t)ent[k]] = (int)k;

    gl_resident *r = new (std::nothrow) gl_resident();
    if (!r) return GL_E_NOMEM;
    s->res = r;
    auto up = [&](int **dst, const std::vector<int> &v) -> int {
        GL_MALLOC(ctx, (void **)dst, std::max<size_t>(v.size(), 1) * sizeof(int), "resident1");
        if (!v.empty()) GL_HIP(ctx, hipMemcpy(*dst, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
        return GL_OK;
    };
    int rc;
    if ((rc = up(&r->bulk_tile, bulk)) || (rc = up(&r->fr_tile, fringe)) || (rc = up(&r->col_ptr, ptr)) ||
        (rc = up(&r->col_ent, pos))) {
        gl_resident_free(s);
        return rc;
    }
    const size_t ncol = (size_t)F + (size_t)S * KF;
    if (gl_ctx_malloc(ctx, (void **)&r->colslab, ncol * 2 * GL_TB * 8, "resident1") != hipSuccess ||
        gl_ctx_malloc(ctx, (void **)&r->tnorm, (size_t)S * 2 * 8, "resident1") != hipSuccess ||
        gl_ctx_malloc(ctx, (void **)&r->bar, GL_BAR_WORDS * 4, "resident1") != hipSuccess) {
        gl_resident_free(s);
        return gl_set_error(ctx, GL_E_NOMEM, "resident engine: hipMalloc failed");
    }
    (void)hipMemset(r->colslab, 0, ncol * 2 * GL_TB * 8);
    (void)hipMemset(r->tnorm, 0, (size_t)S * 2 * 8);
    ResDev &P = r->dev;
    P.z = D.z;
    P.dvar = D.dvar;
    P.ybar = D.ybar;
    P.pbar = D.pbar;
    P.dprev = D.dprev;
    P.rowslab_p = D.rowslab_p;
    P.rowslab_w = D.rowslab_w;
    P.colslab = r->colslab;
    P.bulk_tile = r->bulk_tile;
    P.fr_tile = r->fr_tile;
    P.col_ptr = r->col_ptr;
    P.col_ent = r->col_ent;
    P.tnorm = r->tnorm;
    P.bnorm = D.bnorm;
    P.dstat = D.dstat;
    P.ctr = D.ctr;
    P.bar = r->bar;
    P.dbg = nullptr;
    if (getenv("GRAPHLEARN_AMD_RES_STAMPS")) {
        if (hipMalloc((void **)&r->dbg, (size_t)NWG * 16 * 6 * 8) == hipSuccess) {
            (void)hipMemset(r->dbg, 0, (size_t)NWG * 16 * 6 * 8);
            P.dbg = r->dbg;
        }
    }
    P.n = D.n;
    P.nb = D.nb;
    P.n_pad = D.n_pad;
    P.F = F;
    P.NF = NF;
    P.S = S;
    P.NWG = NWG;
    P.KF = KF;
    P.R = R;
    return GL_OK;
}

// Runs up to `iters` iterations in ONE launch (asynchronous).  The iterate stays in w[it0 & 1];
// the caller fixes the ping-pong parity afterwards (gl_resident_finish).
int gl_resident_launch(gl_logdeg *s, int64_t iters, double tol) {
    gl_ctx *ctx = s->ctx;
    gl_resident *r = s->res;
    ResDev P = r->dev;
    const LogdegDev &D = s->dev;
    P.w = D.w[s->it & 1];
    P.gamma = D.gamma;
    P.c = D.c;
    P.g2 = D.g2;
    P.a4g = D.a4g;
    P.tol = tol;
    P.it0 = s->it;
    P.maxit = (int)std::min<int64_t>(iters, 1 << 30);
    P.rendezvous = P.NWG + (getenv("GRAPHLEARN_AMD_RES_TEST_NOT_RESIDENT") ? 1 : 0);
    GL_HIP(ctx, hipMemsetAsync(r->bar, 0, GL_BAR_WORDS * 4, ctx->stream));
    if (D.n % GL_TB == 0)
        k_resident<false><<<dim3((unsigned)P.NWG), dim3(RT), 0, ctx->stream>>>(P);
    else  // ragged last block column: its tiles are masked bulk tiles
        k_resident<true><<<dim3((unsigned)P.NWG), dim3(RT), 0, ctx->stream>>>(P);
    GL_HIP(ctx, hipGetLastError());
    return GL_OK;
}

// After the launch has completed: number of iterations it ran, stop flag, parity fix-up.
// *not_resident: the workgroups never met at the start-up rendezvous (the GPU is shared, or fewer
// CUs are available than the grid needs); nothing was modified and the caller runs the streaming
// engine instead.  A barrier that fails after the rendezvous is an error.
int gl_resident_finish(gl_logdeg *s, int64_t *iters_done, bool *stopped, bool *not_resident, bool *gated_out) {
    gl_ctx *ctx = s->ctx;
    long long *ct = (long long *)((double *)ctx->pinned + 8);
    GL_HIP(ctx, hipMemcpyAsync(ct, s->dev.ctr, 5 * 8, hipMemcpyDeviceToHost, ctx->stream));
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *not_resident = false;
    if (ct[4] != 0) {
        if (ct[4] != 1 || (int64_t)ct[1] != s->it)
            return gl_set_error(ctx, GL_E_HIP, "resident engine: grid barrier timed out in mid-run");
        *not_resident = true;
        GL_HIP(ctx, hipMemsetAsync(s->dev.ctr, 0, 2 * 8, ctx->stream));
        GL_HIP(ctx, hipMemsetAsync(s->dev.ctr + 4, 0, 8, ctx->stream));
        GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return GL_OK;
    }
    const int64_t itf = (int64_t)ct[1];
    const LogdegDev &D = s->dev;
    if ((itf & 1) != (s->it & 1)) {  // the iterate must live in w[it & 1] for everybody else
        const size_t bytes = (size_t)D.ntiles * GL_TILE_ELEMS * sizeof(double);
        GL_HIP(ctx, hipMemcpyAsync(D.w[itf & 1], D.w[s->it & 1], bytes, hipMemcpyDeviceToDevice, ctx->stream));
    }
    gl_tile_dump_stamps(s);
    if (s->res && s->res->dbg) {
        const int NWG = s->res->dev.NWG;
        std::vector<unsigned long long> h((size_t)NWG * 16 * 6);
        (void)hipMemcpy(h.data(), s->res->dbg, h.size() * 8, hipMemcpyDeviceToHost);
        double ph[5] = {0, 0, 0, 0, 0};
        long cnt = 0;
        for (int w = 0; w < NWG; ++w)
            for (int k = 0; k < 16; ++k) {
                const unsigned long long *o = &h[((size_t)w * 16 + k) * 6];
                if (!o[5]) continue;
                for (int q = 0; q < 5; ++q) ph[q] += (double)(o[q + 1] - o[q]);
                ++cnt;
            }
        // slowest arriver / pure barrier latency per iteration (averaged over the stamped iterations)
        double mx_edge = 0, mx_node = 0, mn_b1 = 0, mn_b2 = 0;
        int kc = 0;
        for (int k = 0; k < 16; ++k) {
            double me = 0, mn = 0, b1 = 1e30, b2 = 1e30;
            bool any = false;
            for (int w = 0; w < NWG; ++w) {
                const unsigned long long *o = &h[((size_t)w * 16 + k) * 6];
                if (!o[5]) continue;
                any = true;
                me = std::max(me, (double)(o[1] - o[0]));
                b1 = std::min(b1, (double)(o[2] - o[1]));
                mn = std::max(mn, (double)(o[3] - o[2]));
                b2 = std::min(b2, (double)(o[4] - o[3]));
            }
            if (any) {
                mx_edge += me, mx_node += mn, mn_b1 += b1, mn_b2 += b2;
                ++kc;
            }
        }
        if (kc)
            fprintf(stderr, "[resident stamps] slowest edge %.2f  fastest barrier1 %.2f  slowest node %.2f  fastest barrier2 %.2f\n",
                    mx_edge / kc / 100, mn_b1 / kc / 100, mx_node / kc / 100, mn_b2 / kc / 100);
        if (cnt)
            fprintf(stderr, "[resident stamps] per iteration (us): edge %.2f  barrier1 %.2f  node %.2f  barrier2 %.2f  stop-rule %.2f  (NWG %d)\n",
                    ph[0] / cnt / 100, ph[1] / cnt / 100, ph[2] / cnt / 100, ph[3] / cnt / 100, ph[4] / cnt / 100, NWG);
    }
    if (iters_done) *iters_done = itf - s->it;
    if (stopped) *stopped = ct[0] == 1;  // (2: the tile engine's dual gate ended the launch -- not the stopping rule; see gated_out)
    if (gated_out) *gated_out = ct[0] == 2;
    s->it = itf;
    // the done flag belongs to the call that raised it
    GL_HIP(ctx, hipMemsetAsync(s->dev.ctr, 0, 2 * 8, ctx->stream));
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GL_OK;
}

// Static work decomposition of the tile engine (gl_tile.hip) below the rounds: which wave works on which chunk of which tile,
// who takes the diagonal tiles' row pairs, which node blocks a workgroup touches.  Host code only, no HIP in here: the header
// is also compiled by plain g++ into the harness of tests/test_tile_plan_cpu.py, which checks the plan's invariants for
// every lattice size without a GPU.
#pragma once

#include <algorithm>
#include <cstring>
#include <functional>
#include <utility>
#include <vector>

#ifndef GL_TILE_MIN_BULK_TILES
#define GL_TILE_MIN_BULK_TILES 3
#endif

namespace gl_tile_plan {

constexpr int TB = 64;        // tile edge (= GL_TB, gl_common.h)
constexpr int SLOTS = TB / 2;  // 32 row pairs per tile
constexpr int TW = 8;          // waves per workgroup (two per SIMD)
constexpr int TT = TW * 64;
constexpr int NBW = 16;        // node blocks one workgroup may touch in the rounds (8 columns + up to 8 rows); descriptor size
constexpr int NBW_FIT = 13;    // ... while the lattice fits the register files (8 chunks: 9 seen); what s_node holds there
constexpr int FRMAX = 2;       // fringe row pairs per wave
constexpr int RING = 4;        // register ring of streamed z row pairs (U = 32)
constexpr int NACQ = (NBW + TW - 1) / TW;  // node blocks one wave fetches
constexpr int RND = 4;         // records per wave and round of an owner's poll
constexpr int XMAX = 3;        // split extra tiles per workgroup (rounds just past the register files)

struct WaveDesc {  // 16 ints
    int tile;      // bulk tile of my chunk (-1: none)
    int I, J;      // its node blocks
    int a0;        // first row pair of the chunk inside the tile
    int si, sj;    // slots of I and J in the workgroup's block list
    int fcnt;      // fringe group: row pairs fa0 .. fa0 + fcnt - 1 of diagonal tile (fB, fB); 0: none
    int fa0, fB, fs;  // ... its first row pair, node block, slot of that block in the workgroup's list
    int ftgt;      // where its degree partials are folded: 0 = my row vector (fB == I, or no chunk), 1 = my column vector (fB == J)
    // n > 4096 ("rounds", see k_tile): the wave works on tiles (I + r, J), r = 0 .. nr - 1, one per round; the first one
    // is resident (tile >= 0) when it is not a diagonal tile, the others stream
    int nr;        // rounds this wave takes part in
    int rs;        // slot of node block I + r in the workgroup's block list: 4 bits per round
    int pad[3];
};
struct WgDesc {  // 128 ints
    int nblk;
    int own;             // node block this workgroup owns (-1: none)
    int own_e0, own_e1;  // its records in `part`
    int blk[NBW], ent[NBW], rowmask[NBW], colmask[NBW];
    unsigned slice_lo, slice_hi;  // owners: bit r set = workgroup own + r nb takes part (publishes norm records)
    int active;                   // this workgroup has edges or owns a block (the others leave after the rendezvous)
    int nrnd;                     // rounds of this workgroup (n > 4096; rowmask holds 8 bits per round, rowmask2 rounds 4 .. 7)
    int rowmask2[NBW];
    // rounds, "split extras" (just past the register files): tiles (beyond the rectangle) that ALL eight waves of the
    // workgroup stream together, 4 row pairs each
    int nx;            // how many (<= XMAX)
    int xt[4];         // their tile ids
    int xs[4];         // slots of their row block (bits 0..7) and column block (8..15) in the workgroup's block list; bit 16: diagonal tile
    int pad[128 - 8 - 5 * NBW - 9];
};
static_assert(sizeof(WaveDesc) == 64 && sizeof(WgDesc) == 512, "descriptor sizes");


inline long long plan_tile_id(long long I, long long J, long long nb) { return I * nb - I * (I - 1) / 2 + (J - I); }  // = gl_tile_id

// The static work decomposition of ONE lattice of nb node blocks on at most ncu workgroups (below the rounds).
// NWG_out = 0: it does not apply.
inline int tile_plan_fit(const int nb, const int ncu, const bool forced, std::vector<WaveDesc> &wd_out, std::vector<WgDesc> &gd_out,
                         int &NWG_out, int &U_out, const int force_u = 0, const bool extra_wg = true) {
    NWG_out = 0;
    U_out = SLOTS;
    const int F = nb * (nb - 1) / 2;  // bulk (off-diagonal) tiles
    if (nb > 64 || F > TW * ncu) return 0;  // does not fit the register files
    if (!forced && F < GL_TILE_MIN_BULK_TILES) return 0;  // (n <= 128 belongs to the small engine)
    // chunk length: the finest split that still fits one chunk per wave of the chip (shorter edge phase)
    int U = F * 4 <= TW * ncu ? 8 : (F * 2 <= TW * ncu ? 16 : 32);
    if ((force_u == 8 || force_u == 16 || force_u == 32) && F * (SLOTS / force_u) <= TW * ncu) U = force_u;
    // Order of the bulk tiles.  A workgroup's 8 chunks are U / 4 tiles: one "super-tile" of SR block rows x SC block
    // columns (2 x 4, 2 x 2, 1 x 2), which touches SR + SC node blocks only.  Complete super-tiles come first, one per
    // workgroup; the tiles of incomplete ones (along the diagonal and the lattice's edge) follow row by row.
    const int SR = U == 8 ? 1 : 2, SC = U == 32 ? 4 : 2;
    std::vector<std::pair<int, int>> bulk, rest;
    for (int IB = 0; SR * IB < nb; ++IB)
        for (int JB = 0; SC * JB < nb; ++JB) {
            // (Every other super-tile lists its tiles the other way round -- second block row first; with one block row:
            //  second half of the columns first.  The first four chunks of a workgroup go to the OLDER waves of its SIMDs,
            //  the ones that take the fringe: were they always the super-tile's first row, no older wave would ever work on
            //  an odd block row, and the diagonal tiles of the first odd blocks would find no taker among them -- 120 groups
            //  short at 64 blocks.)
            const bool flip = ((IB + JB) & 1) != 0;
            std::vector<std::pair<int, int>> st;
            for (int a0_ = 0; a0_ < SR; ++a0_)
                for (int b0_ = 0; b0_ < SC; ++b0_) {
                    const int a = (flip && SR > 1) ? SR - 1 - a0_ : a0_;
                    const int b = (flip && SR == 1) ? (b0_ + SC / 2) % SC : b0_;
                    const int I = SR * IB + a, J = SC * JB + b;
                    if (I < J && J < nb) st.push_back({I, J});
                }
            auto &dst = (int)st.size() == SR * SC ? bulk : rest;
            dst.insert(dst.end(), st.begin(), st.end());
        }
    std::sort(rest.begin(), rest.end());
    bulk.insert(bulk.end(), rest.begin(), rest.end());
    const int per_tile = SLOTS / U, nchunks = F * per_tile;
    int NWG = (nchunks + TW - 1) / TW;
    NWG = std::max(NWG, (SLOTS * nb * 4 + TW * FRMAX * 3 - 1) / (TW * FRMAX * 3));  // room for the fringe, a third to spare
    NWG = std::max(NWG, nb);                                                         // every node block needs an owner
    if (NWG > ncu || NWG > 64 * nb) return 0;
    std::vector<WaveDesc> wd((size_t)NWG * TW);
    auto blank = [](WaveDesc &w) {
        memset(&w, 0, sizeof w);
        w.tile = -1;
        w.fB = -1;
    };
    for (WaveDesc &w : wd) blank(w);
    for (int q = 0; q < nchunks; ++q) {
        WaveDesc &w = wd[(size_t)q];
        const int f = q / per_tile;
        w.tile = (int)plan_tile_id(bulk[(size_t)f].first, bulk[(size_t)f].second, nb);
        w.I = bulk[(size_t)f].first;
        w.J = bulk[(size_t)f].second;
        w.a0 = (q % per_tile) * U;
    }
    // fringe: the 32 row pairs of diagonal tile (B, B) go, in groups of <= FRMAX, to waves whose chunk already works on
    // block B as its row or column block (then the workgroup's block list does not grow and the partials fold into
    // the chunk's vectors), else to waves without a chunk.  A chunk of tile (I, J) can serve I or J.
    //
    // Who takes them, in this order (measured at n = 4096, phase stamps of every wave of the chip: the OLDER wave of a SIMD
    // -- waves 0..3 win its issue arbitration -- ends its pass 0.25 us before the younger one, which is two row pairs of
    // slack; a younger wave that carries a fringe group as well ends 0.5 us after everybody else, and every owner waits
    // for the slowest producer of its block, every iteration):
    //   1. the older half of the chunk waves, `cap` row pairs each (the even share), then up to FRMAX;
    //   2. waves without a chunk; workgroups are ADDED for the purpose while the chip has CUs left;
    //   3. only then the younger half.
    // Within a stage the deal is a maximum assignment (augmenting paths: a wave serves ONE of its two blocks, and at
    // n = 4096 the older half's capacity, 2016 row pairs, is barely below the 2048 there are -- a greedy deal by parity
    // left 250 of them over, i.e. to a hundred younger waves).
    {
        std::vector<int> need((size_t)nb, SLOTS);
        const int cap = nchunks > 0 ? std::max(1, std::min(FRMAX, (SLOTS * nb + nchunks - 1) / nchunks)) : FRMAX;
        auto give = [&](WaveDesc &w, int B, int most) {
            const int take = std::min(most, need[(size_t)B]);
            if (take <= 0 || w.fcnt != 0) return;
            w.fB = B;
            w.fa0 = SLOTS - need[(size_t)B];
            w.fcnt = take;
            w.ftgt = (w.tile >= 0 && w.I != B) ? 1 : 0;  // my column vector when B is my column block
            need[(size_t)B] -= take;
        };
        // One stage over the chunk waves of one half: groups of `most` row pairs by a maximum assignment, then whatever
        // smaller remainders fit.
        auto chunk_stage = [&](bool older, int most) {
            std::vector<int> cand;  // waves of this half that carry a chunk and no fringe group yet
            for (size_t v = 0; v < wd.size(); ++v)
                if (wd[v].tile >= 0 && wd[v].fcnt == 0 && (((int)v % TW) < TW / 2) == older) cand.push_back((int)v);
            std::vector<int> room((size_t)nb);  // groups of `most` row pairs block B still has to hand out
            for (int B = 0; B < nb; ++B) room[(size_t)B] = need[(size_t)B] / most;
            std::vector<int> at(wd.size(), -1);              // wave -> the block it serves
            std::vector<std::vector<int>> taken((size_t)nb); // block -> waves that serve it
            std::vector<char> seen((size_t)nb);
            auto pref = [&](const WaveDesc &w, int k) {  // the wave's two blocks, the one its parity prefers first
                const bool odd = ((w.I + w.J + w.a0 / U) & 1) != 0;
                return (k == 0) == odd ? w.J : w.I;
            };
            std::function<bool(int)> place = [&](int v) -> bool {
                for (int k = 0; k < 2; ++k) {
                    const int B = pref(wd[(size_t)v], k);
                    if (seen[(size_t)B]) continue;
                    seen[(size_t)B] = 1;
                    if ((int)taken[(size_t)B].size() < room[(size_t)B]) {
                        taken[(size_t)B].push_back(v);
                        at[(size_t)v] = B;
                        return true;
                    }
                    for (size_t q = 0; q < taken[(size_t)B].size(); ++q) {  // move one of its waves to that wave's other block
                        const int v2 = taken[(size_t)B][q];
                        at[(size_t)v2] = -1;
                        if (place(v2)) {  // (v2 has found room elsewhere: `seen` keeps it from coming back here)
                            taken[(size_t)B][q] = v;
                            at[(size_t)v] = B;
                            return true;
                        }
                        at[(size_t)v2] = B;
                    }
                }
                return false;
            };
            for (int v : cand) {
                std::fill(seen.begin(), seen.end(), 0);
                (void)place(v);
            }
            for (int v : cand)
                if (at[(size_t)v] >= 0) give(wd[(size_t)v], at[(size_t)v], most);
            for (int v : cand) {  // remainders below a whole group
                WaveDesc &w = wd[(size_t)v];
                if (w.fcnt != 0) continue;
                const int B0 = pref(w, 0), B1 = pref(w, 1);
                give(w, need[(size_t)B0] >= need[(size_t)B1] ? B0 : B1, most);
            }
        };
        auto idle_stage = [&]() {  // waves without a chunk: the block with the most row pairs left, a workgroup's waves side by side
            int prev = -1;
            for (size_t v = 0; v < wd.size(); ++v) {
                WaveDesc &w = wd[v];
                if (w.fcnt != 0 || w.tile >= 0) continue;
                if ((int)v % TW == 0) prev = -1;
                int B = 0;
                for (int b = 1; b < nb; ++b)
                    if (need[(size_t)b] > need[(size_t)B]) B = b;
                if (prev >= 0 && need[(size_t)prev] > 0) B = prev;  // (stay on the previous wave's block: fewer node blocks per workgroup)
                give(w, B, FRMAX);
                prev = B;
            }
        };
        chunk_stage(true, cap);
        if (cap < FRMAX) chunk_stage(true, FRMAX);
        idle_stage();
        int left = 0;
        for (int B = 0; B < nb; ++B) left += need[(size_t)B];
        if (left > 0 && extra_wg) {
            const int extra = std::min(ncu - NWG, (left + TW * FRMAX - 1) / (TW * FRMAX));
            if (extra > 0) {
                NWG += extra;
                wd.resize((size_t)NWG * TW);
                for (size_t v = wd.size() - (size_t)extra * TW; v < wd.size(); ++v) blank(wd[v]);
                idle_stage();
            }
        }
        chunk_stage(false, cap);
        if (cap < FRMAX) chunk_stage(false, FRMAX);
        for (int B = 0; B < nb; ++B)
            if (need[(size_t)B] > 0) return 0;  // (tiny lattices on few workgroups: the other engines serve them)
    }
    if (NWG > 64 * nb) return 0;
    // block lists, records, masks
    std::vector<WgDesc> gd((size_t)NWG);
    for (int g = 0; g < NWG; ++g) {
        WgDesc &G = gd[(size_t)g];
        memset(&G, 0, sizeof G);
        G.own = g < nb ? g : -1;
        auto slot_of = [&](int B) -> int {
            for (int k = 0; k < G.nblk; ++k)
                if (G.blk[k] == B) return k;
            if (G.nblk >= NBW_FIT) return -1;
            G.blk[G.nblk] = B;
            return G.nblk++;
        };
        for (int v = 0; v < TW; ++v) {
            WaveDesc &w = wd[(size_t)(g * TW + v)];
            if (w.tile >= 0) {
                w.si = slot_of(w.I);
                w.sj = slot_of(w.J);
                if (w.si < 0 || w.sj < 0) return 0;  // too many node blocks for one workgroup's LDS
                G.rowmask[w.si] |= 1 << v;
                G.colmask[w.sj] |= 1 << v;
            }
            if (w.fcnt > 0) {
                w.fs = slot_of(w.fB);
                if (w.fs < 0) return 0;
                if (w.tile < 0) G.rowmask[w.fs] |= 1 << v;  // a wave without a chunk: its row vectors carry the group alone
            }
        }
        G.active = (G.nblk > 0 || G.own >= 0) ? 1 : 0;
    }
    wd_out.swap(wd);
    gd_out.swap(gd);
    NWG_out = NWG;
    U_out = U;
    return 0;
}



}  // namespace gl_tile_plan

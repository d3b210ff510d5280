// Test harness (CPU, plain g++): prints one line per lattice size with the invariants of the tile engine's static work
// decomposition (graph_learning_amd/csrc/gl_tile_plan.h); tests/test_tile_plan_cpu.py reads them.
#include <cstdio>
#include <cstdlib>
#include <set>

#include "gl_tile_plan.h"
using namespace gl_tile_plan;

int main(int argc, char **argv) {
    const int ncu = argc > 1 ? atoi(argv[1]) : 256;
    for (int nb = 2; nb <= 64; ++nb) {
        std::vector<WaveDesc> wd;
        std::vector<WgDesc> gd;
        int NWG = 0, U = 0;
        tile_plan_fit(nb, ncu, true, wd, gd, NWG, U);
        if (!NWG) {
            printf("nb %d plan 0\n", nb);
            continue;
        }
        // every chunk of every bulk tile exactly once
        std::set<std::pair<int, int>> chunks;
        bool chunks_ok = true;
        // every row pair of every diagonal tile exactly once
        std::vector<std::vector<int>> rows(nb, std::vector<int>(SLOTS, 0));
        int young = 0, older = 0, idle = 0, fringe_adjacent = 1;
        for (size_t v = 0; v < wd.size(); ++v) {
            const WaveDesc &w = wd[v];
            if (w.tile >= 0) {
                chunks_ok = chunks_ok && w.I < w.J && w.J < nb && w.tile == (int)plan_tile_id(w.I, w.J, nb) && w.a0 % U == 0 && w.a0 + U <= SLOTS;
                chunks_ok = chunks_ok && chunks.insert({w.tile, w.a0}).second;
            }
            if (w.fcnt) {
                if (w.fB < 0 || w.fB >= nb || w.fcnt > FRMAX || w.fa0 < 0 || w.fa0 + w.fcnt > SLOTS) {
                    fringe_adjacent = 0;
                    continue;
                }
                for (int r = w.fa0; r < w.fa0 + w.fcnt; ++r) ++rows[w.fB][r];
                if (w.tile >= 0 && w.fB != w.I && w.fB != w.J) fringe_adjacent = 0;  // a chunk wave serves one of ITS blocks
                if (w.tile >= 0 && w.ftgt != (w.fB == w.I ? 0 : 1)) fringe_adjacent = 0;
                if (w.tile >= 0) ((int)(v % TW) < TW / 2 ? older : young) += 1;
                else ++idle;
            }
        }
        bool fringe_ok = true;
        for (int B = 0; B < nb; ++B)
            for (int r = 0; r < SLOTS; ++r) fringe_ok = fringe_ok && rows[B][r] == 1;
        const int F = nb * (nb - 1) / 2;
        chunks_ok = chunks_ok && (int)chunks.size() == F * (SLOTS / U);
        int maxblk = 0, owners = 0;
        bool lists_ok = (int)gd.size() == NWG && (int)wd.size() == NWG * TW;
        for (int g = 0; g < NWG && lists_ok; ++g) {
            const WgDesc &G = gd[g];
            maxblk = std::max(maxblk, G.nblk);
            owners += G.own >= 0;
            for (int v = 0; v < TW; ++v) {
                const WaveDesc &w = wd[(size_t)g * TW + v];
                if (w.tile >= 0) lists_ok = lists_ok && G.blk[w.si] == w.I && G.blk[w.sj] == w.J && ((G.rowmask[w.si] >> v) & 1) && ((G.colmask[w.sj] >> v) & 1);
                if (w.fcnt) lists_ok = lists_ok && G.blk[w.fs] == w.fB;
            }
        }
        printf("nb %d plan 1 U %d NWG %d chunks_ok %d fringe_ok %d adjacent_ok %d lists_ok %d maxblk %d owners %d older %d younger %d idle %d\n", nb, U, NWG,
               (int)chunks_ok, (int)fringe_ok, fringe_adjacent, (int)lists_ok, maxblk, owners, older, young, idle);
    }
    return 0;
}

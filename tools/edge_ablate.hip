// Development probe: which part of the edge pass costs what?  Includes the product kernel source
// and times ablated instantiations (results are WRONG by construction; timing only).
//   ./edge_ablate [n=4096] [reps=300]
// build (from graph_learning_amd/csrc): hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../include -I. \
//   ../../tools/edge_ablate.hip gl_ctx.hip gl_gram.hip gl_theta.hip gl_resident.hip gl_small.hip -o ../../tools/edge_ablate -ldl
#include "../graph_learning_amd/csrc/gl_logdeg.hip"
#include <chrono>

template <int ABL>
static double time_variant(gl_logdeg *s, int reps) {
    gl_ctx *ctx = s->ctx;
    const LogdegDev &P = s->dev;  // the session's own grid: seg_dest / col_ptr are only valid for it
    for (int i = 0; i < 10; ++i) k_edge<false, ABL><<<dim3(P.G), dim3(64), 0, ctx->stream>>>(P, P.dvar, P.pbar);
    hipStreamSynchronize(ctx->stream);
    hipEventRecord(ctx->ev0, ctx->stream);
    for (int i = 0; i < reps; ++i) k_edge<false, ABL><<<dim3(P.G), dim3(64), 0, ctx->stream>>>(P, P.dvar, P.pbar);
    hipEventRecord(ctx->ev1, ctx->stream);
    hipEventSynchronize(ctx->ev1);
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    return ms * 1e3 / reps;
}

int main(int argc, char **argv) {
    int64_t n = argc > 1 ? atol(argv[1]) : 4096;
    int reps = argc > 2 ? atoi(argv[2]) : 300;
    gl_ctx *ctx;
    if (gl_ctx_create(0, &ctx)) { printf("no ctx\n"); return 1; }
    gl_logdeg *s;
    if (gl_logdeg_create(ctx, n, &s)) { printf("create: %s\n", gl_last_error(ctx)); return 1; }
    std::vector<double> z((size_t)n * (n - 1) / 2);
    for (size_t i = 0; i < z.size(); ++i) z[i] = 0.5 + (double)((i * 2654435761u) % 1000) / 700.0;
    gl_logdeg_set_distances(s, z.data());
    gl_logdeg_init(s, 1, 1, 0.5, nullptr);
    gl_logdeg_iterate_async(s, 20);
    gl_ctx_sync(ctx);
    const double by = 12.0 * n * (n - 1);
#define RUN(ABL, what) { double us = time_variant<ABL>(s, reps); \
    printf("G %5d ABL %2d  %-46s %7.2f us  algo %5.2f TB/s\n", s->dev.G, ABL, what, us, by / us / 1e6); }
    RUN(0, "full kernel");
    RUN(1, "- node gathers");
    RUN(2, "- row sums");
    RUN(4, "- col sums");
    RUN(16, "- norms");
    RUN(8, "- w store");
    RUN(32, "- z load");
    RUN(23, "- gathers, row, col, norms");
    RUN(31, "- everything but loads+math (no store)");
    RUN(63, "- everything but w load + math");
    RUN(128 + 23, "loads + store only (no math, no sums)");
    RUN(128 + 16 + 1, "no math, no norms, no gathers (keeps row/col sums)");
    // ---- per-wave phase stamps (ABL 64): where does a wave spend its cycles?
    {
        LogdegDev P = s->dev;
        const size_t words = (size_t)P.G * 8 * 4;
        unsigned long long *dbg = nullptr;
        hipMalloc((void **)&dbg, words * 8);
        hipMemset(dbg, 0, words * 8);
        P.dbg = dbg;
        for (int i = 0; i < 3; ++i) k_edge<false, 64><<<dim3(P.G), dim3(64), 0, ctx->stream>>>(P, P.dvar, P.pbar);
        hipStreamSynchronize(ctx->stream);
        std::vector<unsigned long long> h(words);
        hipMemcpy(h.data(), dbg, words * 8, hipMemcpyDeviceToHost);
        unsigned long long tmin = ~0ull, tmax = 0;
        double wait = 0, comp = 0, red = 0, gap = 0; long cnt = 0;
        for (int b = 0; b < P.G; ++b) {
            unsigned long long prev_end = 0;
            for (int k = 0; k < 8; ++k) {
                const unsigned long long *o = &h[((size_t)b * 8 + k) * 4];
                if (o[3] == 0) break;
                tmin = std::min(tmin, o[0]); tmax = std::max(tmax, o[3]);
                wait += (double)(o[1] - o[0]); comp += (double)(o[2] - o[1]); red += (double)(o[3] - o[2]);
                if (prev_end) gap += (double)(o[0] - prev_end);
                prev_end = o[3]; ++cnt;
            }
        }
        printf("stamps: %ld steps over %d waves; per step (cycles): wait %.0f  compute+issue %.0f  reductions %.0f  between-steps %.0f ; "
               "kernel span %.0f cycles (first stamp to last)\n", cnt, P.G, wait / cnt, comp / cnt, red / cnt, gap / cnt, (double)(tmax - tmin));
        // a few individual waves
        for (int b : {0, 1, 777, 2047}) { if (b >= P.G) continue; printf(" wave %4d:", b);
            for (int k = 0; k < 8; ++k) { const unsigned long long *o = &h[((size_t)b * 8 + k) * 4]; if (!o[3]) break;
                printf(" [w%llu c%llu r%llu]", o[1]-o[0], o[2]-o[1], o[3]-o[2]); } printf("\n"); }
    }
    return 0;
}

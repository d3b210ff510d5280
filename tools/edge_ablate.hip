// Development probe: which part of the edge pass costs what?  Includes the product kernel source
// and times ablated instantiations (results are WRONG by construction; timing only).
//   ./edge_ablate [n=4096] [reps=300]
#include "../graph_learning_amd/csrc/gl_logdeg.hip"
#include <chrono>

template <int ABL>
static double time_variant(gl_logdeg *s, int reps) {
    gl_ctx *ctx = s->ctx;
    const LogdegDev &P = s->dev;  // the session's own grid: seg_dest / col_ptr are only valid for it
    for (int i = 0; i < 10; ++i) k_edge<false, ABL><<<dim3(P.G), dim3(64), 0, ctx->stream>>>(P);
    hipStreamSynchronize(ctx->stream);
    hipEventRecord(ctx->ev0, ctx->stream);
    for (int i = 0; i < reps; ++i) k_edge<false, ABL><<<dim3(P.G), dim3(64), 0, ctx->stream>>>(P);
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
    return 0;
}

// Internal declarations shared by the HIP translation units of libgraphlearn_amd.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "graphlearn_amd.h"

#define GL_API extern "C" __attribute__((visibility("default")))

// ---------------------------------------------------------------------------
// Context: one GPU, one stream, a last-error string and a pair of timer events.
struct gl_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    hipDeviceProp_t prop;
    // small pinned host staging area for status words / reductions
    void *pinned = nullptr;
    void *pinned_dev = nullptr;  // the same memory as the device sees it (mapped)
    size_t pinned_bytes = 0;  // status area only (the staging area follows it)
    bool stage_busy = false;  // an asynchronous staged upload may still be reading the staging area
    // the staging area is a ring of GL_STAGE_SLOTS slots, each fenced by an event of its own: consecutive uploads (a batch of
    // windows: one block of signals per graph) do not wait for the stream to drain between them
    hipEvent_t stage_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool stage_pending[4] = {false, false, false, false};
    int stage_next = 0;
    bool small_attrs_set = false;  // dynamic-LDS limit of the small-engine kernels raised on this device
    // device scratch for the batched entry points (grown on demand, kept: no hipMalloc / hipFree per call)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    // the batched solver's stand-in session for the tile engine (gl_logdeg.hip: BatchCruise), kept for the next call
    void *batch_cruise = nullptr;
    void (*batch_cruise_free)(void *) = nullptr;
    bool batch_cruise_busy = false;  // gl_logdeg_batch_iterate is working on it: gl_ctx_trim_retained leaves it alone
    // released device blocks of the sessions, kept for the next session of the same size (gl_ctx_alloc / gl_ctx_release):
    // a batch of windows creates and destroys a session per graph, and hipFree alone cost 0.3 ms each
    struct PoolBlock {
        void *p;
        size_t bytes;
    };
    std::vector<PoolBlock> pool;
    size_t pool_bytes = 0;
};

// At least `bytes` of device scratch, valid until the next gl_ctx_scratch call on this context (the stream is in
// order and the batched entry points return synchronised, so consecutive calls may reuse it).
int gl_ctx_scratch(gl_ctx *ctx, size_t bytes, void **out);
// Device blocks that outlive their session: a released block serves the next request of exactly its size.  Everything runs
// on the context's one stream, so reuse is ordered behind the previous owner's work without a synchronisation.
int gl_ctx_alloc(gl_ctx *ctx, size_t bytes, void **out);
void gl_ctx_release(gl_ctx *ctx, void *p, size_t bytes);
// hipMalloc for everything else the library allocates.  When the device has no room, what the context itself retains -- the
// released session blocks, the batched solver's stand-in session with its ring (unless a call is using it) -- is given back
// first and the allocation tried once more; the runtime's last-error word is cleared either way, so a caller that falls back
// to another path does not trip over it at its next hipGetLastError().  `tag` names the site for the test hook
// GRAPHLEARN_AMD_TEST_ALLOC_FAIL=<tag>: allocations of that site fail as if the device were full.
hipError_t gl_ctx_malloc(gl_ctx *ctx, void **out, size_t bytes, const char *tag);
size_t gl_ctx_trim_retained(gl_ctx *ctx);  // returns the bytes given back (0: there was nothing)
#define GL_MALLOC(ctx, pp, bytes, tag)                                                                          \
    do {                                                                                                        \
        const size_t b_ = (bytes);                                                                              \
        hipError_t e_ = gl_ctx_malloc((ctx), (void **)(pp), b_, (tag));                                         \
        if (e_ != hipSuccess)                                                                                   \
            return gl_set_error((ctx), e_ == hipErrorOutOfMemory ? GL_E_NOMEM : GL_E_HIP,                       \
                                "hipMalloc(%zu) [%s] failed: %s (%s:%d)", b_, (tag), hipGetErrorString(e_),     \
                                __FILE__, __LINE__);                                                            \
    } while (0)

int gl_set_error(gl_ctx *ctx, int code, const char *fmt, ...);

// Host <-> device copies of caller-owned (pageable) buffers on the context's stream.  Up to
// GL_STAGE_BYTES they go through the context's pinned, device-mapped staging area and a copy kernel (a
// hipMemcpyAsync of a few tens of KB costs ~0.4 ms on this platform); larger ones are left to the runtime.
// gl_h2d is asynchronous (the staging area is fenced by a stream sync before its next use), gl_d2h
// returns with the data in place.
constexpr size_t GL_STATUS_BYTES = 1 << 16;  // pinned[0 .. 64 KiB): status words of the kernels
constexpr size_t GL_STAGE_BYTES = 1 << 20;   // pinned[64 KiB ..): staging, GL_STAGE_SLOTS slots of this size
constexpr int GL_STAGE_SLOTS = 4;
int gl_h2d(gl_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
void gl_stage_idle(gl_ctx *ctx);  // after a synchronisation of the stream: no staged upload is in flight
int gl_d2h(gl_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

#define GL_HIP(ctx, call)                                                                   \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return gl_set_error((ctx), e_ == hipErrorOutOfMemory ? GL_E_NOMEM : GL_E_HIP,   \
                                "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),      \
                                __FILE__, __LINE__);                                        \
    } while (0)

#define GL_TRY(expr)              \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != GL_OK) return rc_; \
    } while (0)

// ---------------------------------------------------------------------------
// Tiled layout of the upper-triangular edge lattice (DESIGN.md "Data layout").
//   TB x TB tiles over block pairs (I <= J), tile id t(I,J) = I*nb - I(I-1)/2 + (J-I),
//   each tile row-major TB*TB doubles; element (r,c) <-> edge (I*TB+r, J*TB+c),
//   meaningful iff i < j < n, otherwise stored as 0.
constexpr int GL_TB = 64;
constexpr int GL_TILE_ELEMS = GL_TB * GL_TB;

static inline int64_t gl_num_blocks(int64_t n) { return (n + GL_TB - 1) / GL_TB; }
static inline int64_t gl_num_tiles(int64_t n) {
    int64_t nb = gl_num_blocks(n);
    return nb * (nb + 1) / 2;
}

__host__ __device__ static inline int64_t gl_tile_id(int64_t I, int64_t J, int64_t nb) {
    return I * nb - I * (I - 1) / 2 + (J - I);
}

// inverse of gl_tile_id: largest I with I*nb - I(I-1)/2 <= t
__host__ __device__ static inline void gl_tile_decode(int64_t t, int64_t nb, int &I, int &J) {
    // solve I^2 - (2nb+1) I + 2t >= 0  ->  I = floor(((2nb+1) - sqrt((2nb+1)^2 - 8t)) / 2)
    double b = 2.0 * (double)nb + 1.0;
    int64_t i = (int64_t)((b - sqrt(b * b - 8.0 * (double)t)) * 0.5);
    if (i < 0) i = 0;
    if (i > nb - 1) i = nb - 1;
    while (i > 0 && i * nb - i * (i - 1) / 2 > t) --i;
    while (i + 1 < nb && (i + 1) * nb - (i + 1) * i / 2 <= t) ++i;
    I = (int)i;
    J = (int)(t - (i * nb - i * (i - 1) / 2) + i);
}

__host__ __device__ static inline int64_t gl_condensed_index(int64_t i, int64_t j, int64_t n) {
    return i * (2 * n - i - 1) / 2 + (j - i - 1);
}

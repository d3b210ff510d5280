// Context, error reporting and the stream timer of libgraphlearn_amd.
#include <cstdarg>

#include <algorithm>

#include "gl_common.h"

int gl_set_error(gl_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

GL_API int gl_version(void) { return GL_VERSION; }

GL_API const char *gl_strerror(int code) {
    switch (code) {
        case GL_OK: return "ok";
        case GL_E_INVALID: return "invalid argument";
        case GL_E_NOMEM: return "out of memory";
        case GL_E_HIP: return "HIP runtime error";
        case GL_E_NONFINITE: return "non-finite value";
        case GL_E_NODEVICE: return "no usable gfx950 device";
        case GL_E_COMM: return "RCCL error";
        case GL_E_STATE: return "call sequence error";
        default: return "unknown error";
    }
}

GL_API int gl_device_count(int *count) {
    if (!count) return GL_E_INVALID;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *count = 0;
        (void)hipGetLastError();
        return GL_OK;  // "no device" is an answer, not an error
    }
    *count = c;
    return GL_OK;
}

GL_API int gl_ctx_create(int device, gl_ctx **out) {
    if (!out) return GL_E_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return GL_E_NODEVICE;
    }
    if (device < 0 || device >= count) return GL_E_INVALID;
    gl_ctx *ctx = new (std::nothrow) gl_ctx();
    if (!ctx) return GL_E_NOMEM;
    ctx->device = device;
    auto fail = [&](int code) {
        delete ctx;
        return code;
    };
    if (hipSetDevice(device) != hipSuccess) return fail(GL_E_HIP);
    if (hipGetDeviceProperties(&ctx->prop, device) != hipSuccess) return fail(GL_E_HIP);
    // The code objects in this library are gfx950 only.
    if (strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) return fail(GL_E_NODEVICE);
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return fail(GL_E_HIP);
    if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess)
        return fail(GL_E_HIP);
    ctx->pinned_bytes = GL_STATUS_BYTES;
    if (hipHostMalloc(&ctx->pinned, GL_STATUS_BYTES + GL_STAGE_SLOTS * GL_STAGE_BYTES, hipHostMallocMapped) != hipSuccess)
        return fail(GL_E_NOMEM);
    for (int k = 0; k < GL_STAGE_SLOTS; ++k)
        if (hipEventCreateWithFlags(&ctx->stage_ev[k], hipEventDisableTiming) != hipSuccess) return fail(GL_E_HIP);
    if (hipHostGetDevicePointer(&ctx->pinned_dev, ctx->pinned, 0) != hipSuccess) return fail(GL_E_HIP);
    *out = ctx;
    return GL_OK;
}

GL_API int gl_ctx_destroy(gl_ctx *ctx) {
    if (!ctx) return GL_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->batch_cruise && ctx->batch_cruise_free) ctx->batch_cruise_free(ctx->batch_cruise);
    for (const gl_ctx::PoolBlock &b : ctx->pool) (void)hipFree(b.p);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    for (int k = 0; k < GL_STAGE_SLOTS; ++k)
        if (ctx->stage_ev[k]) (void)hipEventDestroy(ctx->stage_ev[k]);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return GL_OK;
}

size_t gl_ctx_trim_retained(gl_ctx *ctx) {
    const bool cruise = ctx->batch_cruise && ctx->batch_cruise_free && !ctx->batch_cruise_busy;
    if (ctx->pool.empty() && !cruise) return 0;
    (void)hipStreamSynchronize(ctx->stream);  // (a released block may still be read by work queued before its release)
    size_t freed = ctx->pool_bytes;
    for (const gl_ctx::PoolBlock &b : ctx->pool) (void)hipFree(b.p);
    ctx->pool.clear();
    ctx->pool_bytes = 0;
    if (cruise) {
        ctx->batch_cruise_free(ctx->batch_cruise);
        ctx->batch_cruise = nullptr;
        freed += 1;  // (its size is the owner's business: "something")
    }
    return freed;
}

hipError_t gl_ctx_malloc(gl_ctx *ctx, void **out, size_t bytes, const char *tag) {
    *out = nullptr;
    if (const char *hook = getenv("GRAPHLEARN_AMD_TEST_ALLOC_FAIL"))
        if (tag && !strcmp(hook, tag)) return hipErrorOutOfMemory;
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipSuccess) return e;
    (void)hipGetLastError();
    *out = nullptr;
    if (gl_ctx_trim_retained(ctx) == 0) return e;
    e = hipMalloc(out, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
    }
    return e;
}

GL_API int gl_ctx_trim(gl_ctx *ctx) {
    if (!ctx) return GL_E_INVALID;
    GL_HIP(ctx, hipSetDevice(ctx->device));
    (void)gl_ctx_trim_retained(ctx);
    if (ctx->scratch) {  // (between calls nobody holds the scratch)
        GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->scratch);
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    return GL_OK;
}

int gl_ctx_alloc(gl_ctx *ctx, size_t bytes, void **out) {
    for (size_t k = ctx->pool.size(); k-- > 0;)
        if (ctx->pool[k].bytes == bytes) {
            *out = ctx->pool[k].p;
            ctx->pool_bytes -= bytes;
            ctx->pool.erase(ctx->pool.begin() + (long)k);
            return GL_OK;
        }
    GL_MALLOC(ctx, out, bytes, "session");  // (gives the pool back first when the device is full)
    return GL_OK;
}

void gl_ctx_release(gl_ctx *ctx, void *p, size_t bytes) {
    if (!p) return;
    // what the pool may hold: a sixteenth of the device's memory, 8 GiB at most (beyond it blocks are freed as before)
    const size_t cap = std::min<size_t>((size_t)8 << 30, ctx->prop.totalGlobalMem / 16);
    if (ctx->pool_bytes + bytes > cap || ctx->pool.size() >= 4096) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(p);
        return;
    }
    ctx->pool.push_back(gl_ctx::PoolBlock{p, bytes});
    ctx->pool_bytes += bytes;
}

GL_API const char *gl_last_error(gl_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

GL_API int gl_ctx_device_name(gl_ctx *ctx, char *buf, int buflen) {
    if (!ctx || !buf || buflen <= 0) return GL_E_INVALID;
    snprintf(buf, (size_t)buflen, "%s (%s, %d CUs)", ctx->prop.name, ctx->prop.gcnArchName,
             ctx->prop.multiProcessorCount);
    return GL_OK;
}

GL_API int gl_ctx_sync(gl_ctx *ctx) {
    if (!ctx) return GL_E_INVALID;
    GL_HIP(ctx, hipSetDevice(ctx->device));
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GL_OK;
}

int gl_ctx_scratch(gl_ctx *ctx, size_t bytes, void **out) {
    if (ctx->scratch_bytes < bytes) {
        if (ctx->scratch) {
            GL_HIP(ctx, hipStreamSynchronize(ctx->stream));  // somebody may still be using the old one
            (void)hipFree(ctx->scratch);
        }
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
        const size_t want = (bytes + ((size_t)1 << 20) - 1) >> 20 << 20;
        GL_MALLOC(ctx, &ctx->scratch, want, "scratch");
        ctx->scratch_bytes = want;
    }
    *out = ctx->scratch;
    return GL_OK;
}

GL_API int gl_ctx_timer_start(gl_ctx *ctx) {
    if (!ctx) return GL_E_INVALID;
    GL_HIP(ctx, hipSetDevice(ctx->device));
    GL_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return GL_OK;
}

GL_API int gl_ctx_timer_stop(gl_ctx *ctx, double *elapsed_ms) {
    if (!ctx || !elapsed_ms) return GL_E_INVALID;
    GL_HIP(ctx, hipSetDevice(ctx->device));
    GL_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    // (polled, not slept on: hipEventSynchronize wakes the caller several microseconds after the event has completed, and
    //  the callers of this timer bracket calls of a few hundred microseconds with the wall clock too)
    hipError_t q;
    while ((q = hipEventQuery(ctx->ev1)) == hipErrorNotReady) {
    }
    GL_HIP(ctx, q);
    float ms = 0.f;
    GL_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *elapsed_ms = (double)ms;
    return GL_OK;
}

// Small staged copies are done by a copy KERNEL that reads / writes the pinned (device-mapped) staging area
// directly: hipMemcpyAsync hands anything above ~16 KB to the SDMA engine, whose fixed latency on this
// platform is ~0.4 ms per copy (measured: 40 KB upload 0.47 ms, the same bytes through this kernel 0.06).
__global__ void k_copy_words(const unsigned long long *__restrict__ src, unsigned long long *__restrict__ dst, size_t nwords) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}
static int stage_copy(gl_ctx *ctx, const void *src, void *dst, size_t bytes) {
    const size_t nwords = (bytes + 7) / 8;  // every buffer involved is a multiple of 8 bytes long
    const unsigned blocks = (unsigned)std::min<size_t>(64, (nwords + 255) / 256);
    k_copy_words<<<blocks, 256, 0, ctx->stream>>>((const unsigned long long *)src, (unsigned long long *)dst, nwords);
    GL_HIP(ctx, hipGetLastError());
    return GL_OK;
}

// after a synchronisation of the stream: no staged upload is in flight
void gl_stage_idle(gl_ctx *ctx) {
    ctx->stage_busy = false;
    for (int k = 0; k < GL_STAGE_SLOTS; ++k) ctx->stage_pending[k] = false;
}

int gl_h2d(gl_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) {
    if (bytes == 0) return GL_OK;
    if (bytes > GL_STAGE_BYTES || (bytes & 7)) {
        GL_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
        return GL_OK;
    }
    const int slot = ctx->stage_next;
    ctx->stage_next = (slot + 1) % GL_STAGE_SLOTS;
    if (ctx->stage_pending[slot]) {  // (the copy that last read this slot: four uploads ago)
        GL_HIP(ctx, hipEventSynchronize(ctx->stage_ev[slot]));
        ctx->stage_pending[slot] = false;
    }
    const size_t off = GL_STATUS_BYTES + (size_t)slot * GL_STAGE_BYTES;
    memcpy((char *)ctx->pinned + off, src_host, bytes);
    GL_TRY(stage_copy(ctx, (char *)ctx->pinned_dev + off, dst_dev, bytes));
    GL_HIP(ctx, hipEventRecord(ctx->stage_ev[slot], ctx->stream));
    ctx->stage_pending[slot] = true;
    ctx->stage_busy = true;
    return GL_OK;
}

int gl_d2h(gl_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes) {
    if (bytes == 0) return GL_OK;
    if (bytes > GL_STAGE_BYTES || (bytes & 7)) {
        GL_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        gl_stage_idle(ctx);
        return GL_OK;
    }
    char *stage = (char *)ctx->pinned + GL_STATUS_BYTES;  // (slot 0: behind whatever upload still reads it, same stream)
    GL_TRY(stage_copy(ctx, src_dev, (char *)ctx->pinned_dev + GL_STATUS_BYTES, bytes));
    GL_HIP(ctx, hipStreamSynchronize(ctx->stream));  // also fences any earlier staged upload (same stream)
    gl_stage_idle(ctx);
    memcpy(dst_host, stage, bytes);
    return GL_OK;
}

// Context, error reporting and the stream timer of libgraphlearn_amd.
#include <cstdarg>

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
    ctx->pinned_bytes = 1 << 16;
    if (hipHostMalloc(&ctx->pinned, ctx->pinned_bytes, hipHostMallocDefault) != hipSuccess)
        return fail(GL_E_NOMEM);
    *out = ctx;
    return GL_OK;
}

GL_API int gl_ctx_destroy(gl_ctx *ctx) {
    if (!ctx) return GL_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return GL_OK;
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
    GL_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    GL_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *elapsed_ms = (double)ms;
    return GL_OK;
}

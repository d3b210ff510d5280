// RCCL over xGMI, used ONLY as barrier / reduce-of-a-scalar / gather for the
// batch-of-graphs case (one independent problem per GPU): nothing here is on the
// iteration path.  librccl is opened lazily (dlopen) so that single-GPU users never
// load it.
//
// The reference has no communication layer at all; this is the
// one collective step BASELINE config 4 adds: gather the learned adjacencies.
//
// No call here can hang a node.  A collective only completes when EVERY rank enters it, so
//   * a rank that hits a local error on its way into one (an allocation, a copy, the solver's staging) ABORTS the
//     communicator (ncclCommAbort) before it returns the error: it will never enter, and says so;
//   * every wait for a collective is bounded by the wall clock (gl_comm_set_timeout, default 300 s, or
//     GRAPHLEARN_AMD_COMM_TIMEOUT_S): a rank whose peer never arrives aborts the communicator too and returns
//     GL_E_COMM instead of waiting for the launcher's kill;
//   * an aborted communicator fails every later call at once (GL_E_COMM); the process is expected to exit non-zero.
#include <dlfcn.h>
#include <time.h>

#include <chrono>

#include "gl_logdeg.h"

namespace {

// Minimal subset of the RCCL API (ABI-compatible with rccl.h of ROCm 7.x).
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
typedef int ncclResult_t;
enum { ncclSuccess = 0 };
enum { ncclInt32 = 2, ncclFloat64 = 8 };  // ncclDataType_t
enum { ncclSum = 0, ncclMax = 2 };         // ncclRedOp_t

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                      // (optional)
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;  // (optional)
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi *rccl() {
    static RcclApi api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names) {
            api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (api.handle) {
            api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.handle, "ncclGetUniqueId");
            api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.handle, "ncclCommInitRank");
            api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
            api.AllGather = (decltype(api.AllGather))dlsym(api.handle, "ncclAllGather");
            api.AllReduce = (decltype(api.AllReduce))dlsym(api.handle, "ncclAllReduce");
            api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
            api.CommAbort = (decltype(api.CommAbort))dlsym(api.handle, "ncclCommAbort");
            api.CommGetAsyncError = (decltype(api.CommGetAsyncError))dlsym(api.handle, "ncclCommGetAsyncError");
            if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.AllReduce) {
                dlclose(api.handle);
                api.handle = nullptr;
            }
        }
    }
    return api.handle ? &api : nullptr;
}

}  // namespace

struct gl_comm {
    gl_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    double *scratch = nullptr;  // a few device words for barrier / scalar reductions
    double *gather = nullptr;   // device receive buffer of the last gather
    size_t gather_bytes = 0;
    double timeout_s = 300.0;   // wall-clock bound of every wait for a collective
    bool dead = false;          // aborted: a local error in front of a collective, or a peer that never arrived
};

#define GL_NCCL(ctx, call)                                                                        \
    do {                                                                                          \
        ncclResult_t r_ = (call);                                                                 \
        if (r_ != ncclSuccess)                                                                    \
            return gl_set_error((ctx), GL_E_COMM, "%s failed: %s", #call,                         \
                                rccl()->GetErrorString ? rccl()->GetErrorString(r_) : "rccl error"); \
    } while (0)

namespace {

// This rank will not take part in the collectives to come (or gave up waiting for one): tear the communicator down so that
// nothing of it stays queued on the stream, and remember.  Returns `code` (the error that led here; ctx->err is the caller's).
int comm_fail(gl_comm *c, int code) {
    if (!c->dead) {
        c->dead = true;
        RcclApi *api = rccl();
        if (c->comm && api) {
            if (api->CommAbort) api->CommAbort(c->comm);  // (also ends a collective kernel that waits for a peer)
            else api->CommDestroy(c->comm);
            c->comm = nullptr;
        }
    }
    return code;
}

int comm_alive(gl_comm *c) {
    if (!c->dead) return GL_OK;
    return gl_set_error(c->ctx, GL_E_COMM, "rank %d: the communicator was aborted (an earlier error on this rank, or a peer that never arrived)", c->rank);
}

// Waits for everything queued on the context's stream -- a collective among it -- for at most timeout_s.
int comm_wait(gl_comm *c, const char *what) {
    gl_ctx *ctx = c->ctx;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        const hipError_t e = hipStreamQuery(ctx->stream);
        if (e == hipSuccess) {
            gl_stage_idle(ctx);  // (the stream has drained: no staged upload is in flight any more)
            return GL_OK;
        }
        if (e != hipErrorNotReady) {
            (void)hipGetLastError();
            gl_set_error(ctx, GL_E_HIP, "%s: %s", what, hipGetErrorString(e));
            return comm_fail(c, GL_E_HIP);
        }
        if ((spins & 255u) == 255u) {
            RcclApi *api = rccl();
            ncclResult_t ar = ncclSuccess;
            if (api->CommGetAsyncError && c->comm && api->CommGetAsyncError(c->comm, &ar) == ncclSuccess && ar != ncclSuccess) {
                gl_set_error(ctx, GL_E_COMM, "%s: %s", what, api->GetErrorString ? api->GetErrorString(ar) : "rccl asynchronous error");
                return comm_fail(c, GL_E_COMM);
            }
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (el > c->timeout_s) {
                gl_set_error(ctx, GL_E_COMM, "%s: rank %d of %d waited %.0f s for the other ranks (one of them is missing); communicator aborted",
                             what, c->rank, c->nranks, el);
                return comm_fail(c, GL_E_COMM);
            }
        }
        if (spins > 2000u) {  // (the first ~2000 polls back to back: a barrier among live ranks takes microseconds)
            const timespec ts{0, 50000};
            nanosleep(&ts, nullptr);
        }
    }
}

// the receive buffer of the gathers
int comm_gather_buffer(gl_comm *c, size_t bytes) {
    gl_ctx *ctx = c->ctx;
    if (c->gather_bytes >= bytes) return GL_OK;
    if (c->gather) {
        GL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(c->gather);
    }
    c->gather = nullptr;
    c->gather_bytes = 0;
    GL_MALLOC(ctx, (void **)&c->gather, bytes, "comm");
    c->gather_bytes = bytes;
    return GL_OK;
}

}  // namespace

GL_API int gl_comm_unique_id(void *out128) {
    if (!out128) return GL_E_INVALID;
    RcclApi *api = rccl();
    if (!api) return GL_E_COMM;
    ncclUniqueId id;
    if (api->GetUniqueId(&id) != ncclSuccess) return GL_E_COMM;
    memcpy(out128, id.internal, sizeof id.internal);
    return GL_OK;
}

GL_API int gl_comm_create(gl_ctx *ctx, int rank, int nranks, const void *unique_id128, gl_comm **out) {
    if (!ctx || !out || !unique_id128) return GL_E_INVALID;
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return gl_set_error(ctx, GL_E_INVALID, "bad rank %d of %d", rank, nranks);
    RcclApi *api = rccl();
    if (!api) return gl_set_error(ctx, GL_E_COMM, "librccl could not be loaded: %s", dlerror());
    GL_HIP(ctx, hipSetDevice(ctx->device));
    gl_comm *c = new (std::nothrow) gl_comm();
    if (!c) return GL_E_NOMEM;
    c->ctx = ctx;
    c->rank = rank;
    c->nranks = nranks;
    if (const char *e = getenv("GRAPHLEARN_AMD_COMM_TIMEOUT_S")) c->timeout_s = std::max(0.01, atof(e));
    ncclUniqueId id;
    memcpy(id.internal, unique_id128, sizeof id.internal);
    ncclResult_t r = api->CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return gl_set_error(ctx, GL_E_COMM, "ncclCommInitRank failed: %s",
                            api->GetErrorString ? api->GetErrorString(r) : "rccl error");
    }
    if (gl_ctx_malloc(ctx, (void **)&c->scratch, 64, "comm_create") != hipSuccess) {
        if (api->CommAbort) api->CommAbort(c->comm);  // (the peers' first collective then fails instead of waiting for this rank)
        else api->CommDestroy(c->comm);
        delete c;
        return gl_set_error(ctx, GL_E_NOMEM, "hipMalloc failed");
    }
    *out = c;
    return GL_OK;
}

GL_API int gl_comm_set_timeout(gl_comm *c, double seconds) {
    if (!c || !(seconds > 0)) return GL_E_INVALID;
    c->timeout_s = seconds;
    return GL_OK;
}

// This rank cannot go on (its caller failed in front of a collective): nobody must wait for it.
GL_API int gl_comm_abort(gl_comm *c) {
    if (!c) return GL_E_INVALID;
    (void)hipSetDevice(c->ctx->device);
    (void)comm_fail(c, GL_OK);
    return GL_OK;
}

GL_API int gl_comm_destroy(gl_comm *c) {
    if (!c) return GL_OK;
    if (c->ctx) {
        (void)hipSetDevice(c->ctx->device);
        (void)hipStreamSynchronize(c->ctx->stream);
    }
    if (c->comm && rccl()) rccl()->CommDestroy(c->comm);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->gather) (void)hipFree(c->gather);
    delete c;
    return GL_OK;
}

// Device-side barrier: every rank's stream has reached this point when it returns.
GL_API int gl_comm_barrier(gl_comm *c) {
    if (!c) return GL_E_INVALID;
    gl_ctx *ctx = c->ctx;
    GL_TRY(comm_alive(c));
    const int rc = [&]() -> int {
        GL_HIP(ctx, hipSetDevice(ctx->device));
        GL_HIP(ctx, hipMemsetAsync(c->scratch, 0, 8, ctx->stream));
        GL_NCCL(ctx, rccl()->AllReduce(c->scratch, c->scratch + 1, 1, ncclInt32, ncclSum, c->comm, ctx->stream));
        return GL_OK;
    }();
    if (rc != GL_OK) return comm_fail(c, rc);
    return comm_wait(c, "gl_comm_barrier");
}

GL_API int gl_comm_allreduce_max(gl_comm *c, double *value) {
    if (!c || !value) return GL_E_INVALID;
    gl_ctx *ctx = c->ctx;
    GL_TRY(comm_alive(c));
    const int rc = [&]() -> int {
        GL_HIP(ctx, hipSetDevice(ctx->device));
        GL_HIP(ctx, hipMemcpyAsync(c->scratch + 2, value, 8, hipMemcpyHostToDevice, ctx->stream));
        GL_NCCL(ctx, rccl()->AllReduce(c->scratch + 2, c->scratch + 3, 1, ncclFloat64, ncclMax, c->comm, ctx->stream));
        GL_HIP(ctx, hipMemcpyAsync(value, c->scratch + 3, 8, hipMemcpyDeviceToHost, ctx->stream));
        return GL_OK;
    }();
    if (rc != GL_OK) return comm_fail(c, rc);
    return comm_wait(c, "gl_comm_allreduce_max");
}

// Gather of the learned adjacencies (BASELINE config 4): every rank contributes its
// condensed, thresholded edge vector (E doubles, device-resident, straight from the
// solver session) and receives all of them; host_out (nranks x E) may be NULL to
// keep the result on the device only.
GL_API int gl_comm_allgather_weights(gl_comm *c, gl_logdeg *s, double edge_tol, double *host_out) {
    if (!c || !s) return GL_E_INVALID;
    gl_ctx *ctx = c->ctx;
    if (s->ctx != ctx) return gl_set_error(ctx, GL_E_INVALID, "session and communicator live on different contexts");
    GL_TRY(comm_alive(c));
    const size_t bytes = (size_t)c->nranks * s->n_edges * 8;
    const int rc = [&]() -> int {
        GL_HIP(ctx, hipSetDevice(ctx->device));
        // the session's staging buffer receives the condensed, thresholded weights (device only)
        GL_TRY(gl_logdeg_get_weights_condensed(s, edge_tol, nullptr));
        GL_TRY(comm_gather_buffer(c, bytes));
        GL_NCCL(ctx, rccl()->AllGather(s->stage, c->gather, (size_t)s->n_edges, ncclFloat64, c->comm, ctx->stream));
        if (host_out) GL_HIP(ctx, hipMemcpyAsync(host_out, c->gather, bytes, hipMemcpyDeviceToHost, ctx->stream));
        return GL_OK;
    }();
    if (rc != GL_OK) return comm_fail(c, rc);  // (whatever failed on the way in: this rank does not enter, and says so)
    return comm_wait(c, "gl_comm_allgather_weights");
}

// All-gather of a small fp64 vector given in host memory (count doubles per rank -> nranks x count on
// every rank): per-problem scalars of a batch, e.g. theta and the iteration count.
GL_API int gl_comm_allgather_f64(gl_comm *c, const double *host_in, int64_t count, double *host_out) {
    if (!c || !host_in || !host_out || count < 1) return GL_E_INVALID;
    gl_ctx *ctx = c->ctx;
    GL_TRY(comm_alive(c));
    const size_t bytes = (size_t)(c->nranks + 1) * (size_t)count * 8;  // [send | recv]
    const int rc = [&]() -> int {
        GL_HIP(ctx, hipSetDevice(ctx->device));
        GL_TRY(comm_gather_buffer(c, bytes));
        GL_HIP(ctx, hipMemcpyAsync(c->gather, host_in, (size_t)count * 8, hipMemcpyHostToDevice, ctx->stream));
        GL_NCCL(ctx, rccl()->AllGather(c->gather, c->gather + count, (size_t)count, ncclFloat64, c->comm, ctx->stream));
        GL_HIP(ctx, hipMemcpyAsync(host_out, c->gather + count, (size_t)c->nranks * (size_t)count * 8, hipMemcpyDeviceToHost,
                                   ctx->stream));
        return GL_OK;
    }();
    if (rc != GL_OK) return comm_fail(c, rc);
    return comm_wait(c, "gl_comm_allgather_f64");
}

// The gather payload of SURVEY.md section 8 row f-4: every rank contributes ONE self-describing record --
// the 64-byte header of graph_learning_amd/wire.py (magic "GLAM", version 1, n_nodes, n_edges, problem id,
// iterations run, theta, edge_tol, flags) followed by its condensed, thresholded weights -- so the header
// travels THROUGH the collective with the data it describes.  host_out is nranks x (8 + E) doubles.
// flags: bit 0 = the solver converged, bit 1 = this rank FAILED to solve its problem (its weights are whatever the
// session holds -- zeros after a fresh init -- and every rank learns of the failure from the record itself).
GL_API int gl_comm_allgather_records(gl_comm *c, gl_logdeg *s, double edge_tol, int64_t problem_id, int64_t n_iter,
                                     double theta, int flags, double *host_out) {
    if (!c || !s) return GL_E_INVALID;
    gl_ctx *ctx = c->ctx;
    if (s->ctx != ctx) return gl_set_error(ctx, GL_E_INVALID, "session and communicator live on different contexts");
    GL_TRY(comm_alive(c));
    struct Header {  // little-endian, 64 bytes: the layout of wire.HEADER ("<4sIqqqqddQ")
        char magic[4];
        uint32_t version;
        int64_t n_nodes, n_edges, problem_id, n_iter;
        double theta, edge_tol;
        uint64_t flags;
    } h;
    static_assert(sizeof(Header) == 64, "record header must be 64 bytes");
    memcpy(h.magic, "GLAM", 4);
    h.version = 1;
    h.n_nodes = s->n;
    h.n_edges = s->n_edges;
    h.problem_id = problem_id;
    h.n_iter = n_iter;
    h.theta = theta;
    h.edge_tol = edge_tol;
    h.flags = (uint64_t)(unsigned)flags & 3u;
    const size_t rec = (size_t)s->n_edges + 8;
    const size_t bytes = (size_t)c->nranks * rec * 8;
    const int rc = [&]() -> int {
        GL_HIP(ctx, hipSetDevice(ctx->device));
        GL_TRY(gl_logdeg_get_weights_condensed(s, edge_tol, nullptr));  // condensed weights -> s->stage (device only)
        GL_TRY(gl_h2d(ctx, s->stage_rec, &h, sizeof h));
        GL_TRY(comm_gather_buffer(c, bytes));
        GL_NCCL(ctx, rccl()->AllGather(s->stage_rec, c->gather, rec, ncclFloat64, c->comm, ctx->stream));
        if (host_out) GL_HIP(ctx, hipMemcpyAsync(host_out, c->gather, bytes, hipMemcpyDeviceToHost, ctx->stream));
        return GL_OK;
    }();
    if (rc != GL_OK) return comm_fail(c, rc);
    return comm_wait(c, "gl_comm_allgather_records");
}

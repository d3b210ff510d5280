"""ctypes binding of libgraphlearn_amd.so (the C ABI in include/graphlearn_amd.h).

Host deps: numpy + ctypes only -- no PyTorch.  There is no CPU fallback: if the
shared library is missing, or no gfx950 device is visible, using the estimators
raises (loudly) instead of computing somewhere else.
"""

from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libgraphlearn_amd.so"

GL_OK = 0
GL_E_INVALID = -1
GL_E_NOMEM = -2
GL_E_HIP = -3
GL_E_NONFINITE = -4
GL_E_NODEVICE = -5
GL_E_COMM = -6
GL_E_STATE = -7


class GraphLearnError(RuntimeError):
    """A non-argument failure reported by the native library."""


_c_double_p = C.POINTER(C.c_double)
_c_int64_p = C.POINTER(C.c_int64)
_c_int_p = C.POINTER(C.c_int)

# name -> (restype, argtypes); mirrors include/graphlearn_amd.h one to one
SIGNATURES = {
    "gl_version": (C.c_int, []),
    "gl_strerror": (C.c_char_p, [C.c_int]),
    "gl_device_count": (C.c_int, [_c_int_p]),
    "gl_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "gl_ctx_destroy": (C.c_int, [C.c_void_p]),
    "gl_last_error": (C.c_char_p, [C.c_void_p]),
    "gl_ctx_device_name": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "gl_ctx_sync": (C.c_int, [C.c_void_p]),
    "gl_ctx_trim": (C.c_int, [C.c_void_p]),
    "gl_ctx_timer_start": (C.c_int, [C.c_void_p]),
    "gl_ctx_timer_stop": (C.c_int, [C.c_void_p, _c_double_p]),
    "gl_logdeg_create": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]),
    "gl_logdeg_destroy": (C.c_int, [C.c_void_p]),
    "gl_logdeg_set_signals": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, _c_double_p]),
    "gl_logdeg_theta": (C.c_int, [C.c_void_p, _c_double_p]),
    "gl_logdeg_set_distances": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gl_logdeg_get_distances": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gl_logdeg_init": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p]),
    "gl_logdeg_iterate": (C.c_int, [C.c_void_p, C.c_int64, C.c_double, _c_int64_p, _c_int_p, _c_double_p]),
    "gl_logdeg_iterate_async": (C.c_int, [C.c_void_p, C.c_int64]),
    "gl_logdeg_get_weights_condensed": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p]),
    "gl_logdeg_batch_get_weights": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_void_p]),
    "gl_logdeg_get_weights_square": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p]),
    "gl_logdeg_get_dual": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gl_logdeg_total_iterations": (C.c_int64, [C.c_void_p]),
    "gl_logdeg_time_edge_pass": (C.c_int, [C.c_void_p, C.c_int, _c_double_p]),
    "gl_logdeg_engine": (C.c_int, [C.c_void_p]),
    "gl_logdeg_batch_gram": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]),
    "gl_logdeg_batch_iterate": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_double, _c_int64_p, _c_int_p,
                                          _c_double_p]),
    "gl_log_degrees_fit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_void_p,
                                     C.c_int64, C.c_double, C.c_double, C.c_double, C.c_void_p, _c_int64_p,
                                     _c_double_p]),
    "gl_logmodel_fit_signals": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p,
                                          C.c_int64, C.c_double, C.c_double, C.c_double, C.c_void_p, _c_double_p,
                                          _c_int64_p, _c_double_p]),
    "gl_square_to_vec": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "gl_laplacian_from_vec": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "gl_laplacian_adjoint": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    "gl_squared_pdiffs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "gl_theta_from_distances": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, _c_double_p]),
    "gl_theta_from_matrix": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, _c_double_p]),
    "gl_relative_error": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, _c_double_p]),
    "gl_nonneg_qp_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_int,
                                     C.c_void_p, _c_int_p]),
    "gl_lgmrf_fit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_int, C.c_double,
                               C.c_double, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, _c_int_p, _c_int_p]),
    "gl_lgmrf_time_node_updates": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_int,
                                             C.c_void_p, C.c_int, C.c_int64, C.c_int64, _c_double_p]),
    "gl_graphdict_create": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "gl_graphdict_destroy": (None, [C.c_void_p]),
    "gl_graphdict_set_signals": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gl_graphdict_set_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gl_graphdict_get_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gl_graphdict_set_params": (C.c_int, [C.c_void_p] + [C.c_double] * 8),
    "gl_graphdict_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "gl_graphdict_get_steps": (C.c_int, [C.c_void_p, _c_double_p]),
    "gl_graphdict_engine": (C.c_int, [C.c_void_p]),
    "gl_graphdict_fit": (C.c_int, [C.c_void_p, C.c_int64, C.c_double, C.c_int, C.c_void_p, _c_int64_p, _c_int64_p,
                                   _c_int_p]),
    "gl_graphdict_predict": (C.c_int, [C.c_void_p, C.c_int64, C.c_double, C.c_void_p, _c_int64_p, _c_int_p]),
    "gl_graphdict_time_gradients": (C.c_int, [C.c_void_p, C.c_int, _c_double_p, _c_double_p]),
    "gl_comm_unique_id": (C.c_int, [C.c_void_p]),
    "gl_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "gl_comm_set_timeout": (C.c_int, [C.c_void_p, C.c_double]),
    "gl_comm_abort": (C.c_int, [C.c_void_p]),
    "gl_comm_destroy": (C.c_int, [C.c_void_p]),
    "gl_comm_barrier": (C.c_int, [C.c_void_p]),
    "gl_comm_allreduce_max": (C.c_int, [C.c_void_p, _c_double_p]),
    "gl_comm_allgather_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]),
    "gl_comm_allgather_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gl_comm_allgather_records": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_int64, C.c_int64, C.c_double,
                                            C.c_int, C.c_void_p]),
}

_lib = None
_lib_lock = threading.Lock()


def lib_path() -> str:
    return os.environ.get("GRAPHLEARN_AMD_LIB", os.path.join(_HERE, LIB_NAME))


def load():
    """Load the shared library (once) and declare every prototype."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        path = lib_path()
        if not os.path.exists(path):
            raise ImportError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C graph_learning_amd/csrc`. There is no CPU fallback.")
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def device_count() -> int:
    n = C.c_int(0)
    load().gl_device_count(C.byref(n))
    return n.value


def ptr(a: np.ndarray | None):
    """Host pointer of a C-contiguous float64 array (or NULL)."""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def as_f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


class Context:
    """One GPU + one stream.  Not thread-safe; use one per thread/process."""

    def __init__(self, device: int | None = None):
        self._lib = load()
        if device is None:
            device = int(os.environ.get("GRAPHLEARN_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        h = C.c_void_p()
        rc = self._lib.gl_ctx_create(int(device), C.byref(h))
        if rc != GL_OK:
            msg = self._lib.gl_strerror(rc).decode()
            if rc == GL_E_INVALID:
                raise ValueError(f"gl_ctx_create(device={device}): {msg}")
            raise GraphLearnError(f"gl_ctx_create(device={device}) failed: {msg} -- this library needs an "
                                  f"MI355X (gfx950); there is no CPU fallback")
        self._h = h
        self.device = int(device)

    def check(self, rc: int, what: str = ""):
        if rc == GL_OK:
            return
        detail = self._lib.gl_last_error(self._h).decode(errors="replace")
        msg = f"{what}: {self._lib.gl_strerror(rc).decode()}: {detail}"
        if rc == GL_E_INVALID:
            raise ValueError(msg)
        if rc == GL_E_NOMEM:
            raise MemoryError(msg)
        raise GraphLearnError(msg)

    @property
    def handle(self):
        return self._h

    def name(self) -> str:
        buf = C.create_string_buffer(256)
        self.check(self._lib.gl_ctx_device_name(self._h, buf, 256), "device_name")
        return buf.value.decode()

    def sync(self):
        self.check(self._lib.gl_ctx_sync(self._h), "sync")

    def trim(self):
        """Give back the device memory the context retains between calls (session pool, the batched solver's
        stand-in session and ring, scratch)."""
        self.check(self._lib.gl_ctx_trim(self._h), "trim")

    def timer_start(self):
        self.check(self._lib.gl_ctx_timer_start(self._h), "timer_start")

    def timer_stop(self) -> float:
        ms = C.c_double(0.0)
        self.check(self._lib.gl_ctx_timer_stop(self._h, C.byref(ms)), "timer_stop")
        return ms.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gl_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


_default_ctx = None
_default_lock = threading.Lock()


def default_context() -> Context:
    """Process-wide context on GRAPHLEARN_AMD_DEVICE / LOCAL_RANK / device 0."""
    global _default_ctx
    with _default_lock:
        if _default_ctx is None:
            _default_ctx = Context()
        return _default_ctx


class LogDegreeSession:
    """Staged form of the log-degree solver (device-resident problem)."""

    def __init__(self, n_nodes: int, ctx: Context | None = None):
        self.ctx = ctx or default_context()
        self._lib = self.ctx._lib
        self.n = int(n_nodes)
        self.n_edges = self.n * (self.n - 1) // 2
        h = C.c_void_p()
        self.ctx.check(self._lib.gl_logdeg_create(self.ctx.handle, self.n, C.byref(h)), "logdeg_create")
        self._h = h

    def set_signals(self, x, avg_degree=None, deferred=False):
        """-> theta; with deferred=True nothing comes back (no host round trip): ask ``theta()`` later."""
        x = as_f64(x)
        if x.ndim != 2 or x.shape[1] != self.n:
            raise ValueError(f"x must be (n_samples, {self.n}), got {x.shape}")
        theta = C.c_double(1.0)
        k = -1 if avg_degree is None else int(avg_degree)
        self.ctx.check(self._lib.gl_logdeg_set_signals(self._h, ptr(x), x.shape[0], x.shape[1], k,
                                                       None if deferred else C.byref(theta)), "set_signals")
        return None if deferred else theta.value

    def theta(self) -> float:
        out = C.c_double(1.0)
        self.ctx.check(self._lib.gl_logdeg_theta(self._h, C.byref(out)), "theta")
        return out.value

    def set_distances(self, z):
        z = as_f64(z)
        if z.shape != (self.n_edges,):
            raise ValueError(f"distances must have shape ({self.n_edges},), got {z.shape}")
        self.ctx.check(self._lib.gl_logdeg_set_distances(self._h, ptr(z)), "set_distances")

    def get_distances(self) -> np.ndarray:
        out = np.empty(self.n_edges)
        self.ctx.check(self._lib.gl_logdeg_get_distances(self._h, ptr(out)), "get_distances")
        return out

    def init(self, alpha=1.0, beta=1.0, step_size=0.5, edge_init=None):
        if edge_init is not None:
            edge_init = as_f64(edge_init)
            if edge_init.shape != (self.n_edges,):
                raise ValueError(f"edge_init must have shape ({self.n_edges},), got {edge_init.shape}")
        self.ctx.check(self._lib.gl_logdeg_init(self._h, float(alpha), float(beta), float(step_size), ptr(edge_init)),
                       "init")

    def iterate(self, maxit, tol=0.0):
        n_it = C.c_int64(0)
        conv = C.c_int(0)
        rel = (C.c_double * 2)()
        self.ctx.check(self._lib.gl_logdeg_iterate(self._h, int(maxit), float(tol), C.byref(n_it), C.byref(conv), rel),
                       "iterate")
        return n_it.value, bool(conv.value), (rel[0], rel[1])

    def iterate_async(self, n_iter):
        self.ctx.check(self._lib.gl_logdeg_iterate_async(self._h, int(n_iter)), "iterate_async")

    def weights_condensed(self, edge_tol=-np.inf) -> np.ndarray:
        out = np.empty(self.n_edges)
        self.ctx.check(self._lib.gl_logdeg_get_weights_condensed(self._h, float(edge_tol), ptr(out)), "get_weights")
        return out

    def weights_square(self, edge_tol=-np.inf) -> np.ndarray:
        out = np.empty((self.n, self.n))
        self.ctx.check(self._lib.gl_logdeg_get_weights_square(self._h, float(edge_tol), ptr(out)), "get_weights")
        return out

    def dual(self) -> np.ndarray:
        out = np.empty(self.n)
        self.ctx.check(self._lib.gl_logdeg_get_dual(self._h, ptr(out)), "get_dual")
        return out

    @property
    def engine(self) -> str:
        """"resident", "small" or "stream": the engine long iterate calls of this session run on."""
        return {1: "resident", 2: "small", 3: "resident"}.get(self._lib.gl_logdeg_engine(self._h), "stream")

    @property
    def engine_kernel(self) -> str:
        """The kernel behind ``engine``: "k_tile" (tagged hand-offs, gl_tile.hip) or "k_resident" (grid barriers,
        gl_resident.hip) for the resident family, "k_small", or "k_edge" for the streaming launches."""
        return {1: "k_resident", 2: "k_small", 3: "k_tile"}.get(self._lib.gl_logdeg_engine(self._h), "k_edge")

    def time_edge_pass(self, reps=200) -> float:
        """Average edge-pass duration in microseconds (HIP events; state unchanged)."""
        us = C.c_double(0.0)
        self.ctx.check(self._lib.gl_logdeg_time_edge_pass(self._h, int(reps), C.byref(us)), "time_edge_pass")
        return us.value

    @property
    def total_iterations(self) -> int:
        return int(self._lib.gl_logdeg_total_iterations(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gl_logdeg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class GraphDictSession:
    """Device-resident graph-dictionary problem (gl_graphdict_*): signals, (C, W, dual) state."""

    def __init__(self, n_nodes: int, n_samples: int, n_atoms: int, window_size: int = 1, ctx: Context | None = None):
        self.ctx = ctx or default_context()
        self._lib = self.ctx._lib
        self.n, self.t, self.k, self.window = int(n_nodes), int(n_samples), int(n_atoms), int(window_size)
        self.n_edges = self.n * (self.n - 1) // 2
        h = C.c_void_p()
        self.ctx.check(self._lib.gl_graphdict_create(self.ctx.handle, self.n, self.t, self.k, self.window, C.byref(h)),
                       "graphdict_create")
        self._h = h

    def set_signals(self, x):
        x = as_f64(x)
        if x.shape != (self.t, self.n):
            raise ValueError(f"x must be ({self.t}, {self.n}), got {x.shape}")
        self.ctx.check(self._lib.gl_graphdict_set_signals(self._h, ptr(x)), "graphdict_set_signals")

    def set_state(self, coefficients, weights, dual=None):
        c, w = as_f64(coefficients), as_f64(weights)
        if c.shape != (self.k, self.t) or w.shape != (self.k, self.n_edges):
            raise ValueError(f"state shapes: coefficients {(self.k, self.t)}, weights {(self.k, self.n_edges)}; "
                             f"got {c.shape}, {w.shape}")
        if dual is not None:
            dual = as_f64(dual)
            if dual.shape != (self.t // self.window, self.n):
                raise ValueError(f"dual must be {(self.t // self.window, self.n)}, got {dual.shape}")
        self.ctx.check(self._lib.gl_graphdict_set_state(self._h, ptr(c), ptr(w), ptr(dual)), "graphdict_set_state")

    def get_state(self):
        c, w = np.empty((self.k, self.t)), np.empty((self.k, self.n_edges))
        d = np.empty((self.t // self.window, self.n))
        self.ctx.check(self._lib.gl_graphdict_get_state(self._h, ptr(c), ptr(w), ptr(d)), "graphdict_get_state")
        return c, w, d

    def set_params(self, l1_w=0.0, ortho_w=0.0, l1_c=0.0, log_c=0.0, l1_diff_c=0.0, step_c=1.0, step_w=1.0,
                   step_dual=1.0):
        self.ctx.check(self._lib.gl_graphdict_set_params(self._h, *[float(v) for v in (
            l1_w, ortho_w, l1_c, log_c, l1_diff_c, step_c, step_w, step_dual)]), "graphdict_set_params")

    def set_precision(self, bits: int):
        """32: the two O(K T E) contractions of a step run in packed fp32 (folds and state stay fp64); 64: default."""
        self.ctx.check(self._lib.gl_graphdict_set_precision(self._h, int(bits)), "graphdict_set_precision")

    def steps(self):
        out = (C.c_double * 3)()
        self.ctx.check(self._lib.gl_graphdict_get_steps(self._h, out), "graphdict_get_steps")
        return out[0], out[1], out[2]

    @property
    def engine(self) -> str:
        """"small" (whole loop in one launch of one workgroup) or "launch" (one graph of kernels per step)."""
        return "small" if self._lib.gl_graphdict_engine(self._h) == 1 else "launch"

    def fit(self, max_iter, tol, reduce_step_on_plateau=False):
        """-> (history (max_iter, 2), steps_run, converged_, flags)."""
        hist = np.empty((max(int(max_iter), 0), 2))
        ran, conv, flags = C.c_int64(0), C.c_int64(-1), C.c_int(0)
        self.ctx.check(self._lib.gl_graphdict_fit(self._h, int(max_iter), float(tol), int(bool(reduce_step_on_plateau)),
                                                  ptr(hist), C.byref(ran), C.byref(conv), C.byref(flags)), "graphdict_fit")
        return hist, ran.value, conv.value, flags.value

    def predict(self, max_iter, tol):
        out = np.empty((self.k, self.t))
        ran, flags = C.c_int64(0), C.c_int(0)
        self.ctx.check(self._lib.gl_graphdict_predict(self._h, int(max_iter), float(tol), ptr(out), C.byref(ran),
                                                      C.byref(flags)), "graphdict_predict")
        return out, ran.value, flags.value

    def time_gradients(self, reps=20):
        a, b = C.c_double(0), C.c_double(0)
        self.ctx.check(self._lib.gl_graphdict_time_gradients(self._h, int(reps), C.byref(a), C.byref(b)),
                       "graphdict_time_gradients")
        return a.value, b.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gl_graphdict_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

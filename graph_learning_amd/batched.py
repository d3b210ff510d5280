"""Batched (2-D) path of ``gsp_learn_graph_log_degrees`` -- B graphs with a COUPLED stopping rule.

Reference semantics (graph_learn/smooth_learning.py:128-129,158-159,174): with ``distances`` of
shape (B, E) every array in the loop is (E, B) / (n, B), ``np.linalg.norm(M, ord=2)`` is then the
SPECTRAL norm of the matrix, and all B graphs stop at the same iteration.  Used by
``WindowLogModel`` (temporal/smooth.py) and by the clustering M-steps.

Here each graph is a device-resident solver session; the lockstep loop runs inside the library
(``gl_logdeg_batch_iterate``): per iteration 2 B launches, then only the B diagonal Gram entries come
back -- lambda_max of a Gram matrix lies between its largest diagonal entry and its trace, which
decides most iterations -- and the full B x B Gram matrices / eigenvalues are computed only when
those bounds straddle ``tol`` and for the values reported at the end.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _n_nodes_from_edges(n_edges: int) -> int:
    return int(np.round((1 + np.sqrt(1 + 8 * n_edges)) / 2))


def _spectral(gram: np.ndarray) -> float:
    """Largest singular value of a matrix given the Gram matrix of its columns."""
    return float(np.sqrt(max(np.linalg.eigvalsh(gram)[-1], 0.0)))


def _run_sessions(sessions, ctx, maxit, tol, edge_tol):
    """The coupled loop over initialised sessions -> ((B, E) thresholded weights, info)."""
    n_batch = len(sessions)
    handles = (C.c_void_p * n_batch)(*[s._h for s in sessions])
    n_it, conv, rel = C.c_int64(0), C.c_int(0), (C.c_double * 2)(np.inf, np.inf)
    ctx.check(ctx._lib.gl_logdeg_batch_iterate(handles, n_batch, int(maxit), float(tol), C.byref(n_it), C.byref(conv), rel),
              "batch_iterate")
    out = np.empty((n_batch, sessions[0].n_edges))  # (B, E) == edge_w.T (:174): one copy, one synchronisation
    ctx.check(ctx._lib.gl_logdeg_batch_get_weights(handles, n_batch, float(edge_tol), _lib.ptr(out)), "batch_get_weights")
    return out, {"n_iter": n_it.value, "rel_primal": float(rel[0]), "rel_dual": float(rel[1]),
                 "converged": bool(conv.value)}


def learn_log_degrees_from_signals(signal_blocks, avg_degree, alpha, beta, edge_init=None, maxit=1000, tol=1e-5,
                                   step_size=0.5, edge_tol=1e-3, ctx=None):
    """The batched solver fed with signals: block b (samples x nodes) gives graph b's squared distances (MFMA
    Gram kernel) and theta on the device, where they stay -- no distance matrix travels to the host and back.
    Used by ``WindowLogModel``.  -> (weights (B, E), thetas (B,), info)."""
    blocks = [_lib.as_f64(b) for b in signal_blocks]
    n = blocks[0].shape[1]
    n_edges = n * (n - 1) // 2
    n_batch = len(blocks)
    if edge_init is not None:
        edge_init = _lib.as_f64(edge_init)
        if edge_init.shape != (n_edges, n_batch):
            raise ValueError(f"edge_init must have shape ({n_edges}, {n_batch}) for batched input")
    ctx = ctx or _lib.default_context()
    sessions = [_lib.LogDegreeSession(n, ctx) for _ in range(n_batch)]
    try:
        for b, s in enumerate(sessions):
            s.set_signals(blocks[b], avg_degree, deferred=True)  # distances scaled by theta, resident; nothing comes back
            s.init(alpha, beta, step_size, None if edge_init is None else np.ascontiguousarray(edge_init[:, b]))
        out, info = _run_sessions(sessions, ctx, maxit, tol, edge_tol)
        thetas = [s.theta() for s in sessions]
    finally:
        for s in sessions:
            s.close()
    return out, np.array(thetas), info


def learn_log_degrees_batch(distances, alpha, beta, edge_init=None, maxit=1000, tol=1e-5, step_size=0.5,
                            edge_tol=1e-3, ctx=None, return_info=False):
    distances = _lib.as_f64(distances)
    if distances.ndim != 2:
        raise ValueError("Distances must be vector or matrix")
    n_batch, n_edges = distances.shape
    n = _n_nodes_from_edges(n_edges)
    if edge_init is not None:
        # the reference uses edge_init as the (E, B) iterate directly
        edge_init = _lib.as_f64(edge_init)
        if edge_init.shape != (n_edges, n_batch):
            raise ValueError(f"edge_init must have shape ({n_edges}, {n_batch}) for batched input")
    ctx = ctx or _lib.default_context()
    lib = ctx._lib
    sessions = [_lib.LogDegreeSession(n, ctx) for _ in range(n_batch)]
    try:
        for b, s in enumerate(sessions):
            s.set_distances(distances[b])
            s.init(alpha, beta, step_size, None if edge_init is None else np.ascontiguousarray(edge_init[:, b]))
        out, info = _run_sessions(sessions, ctx, maxit, tol, edge_tol)
    finally:
        for s in sessions:
            s.close()
    if return_info:
        return out, info
    return out

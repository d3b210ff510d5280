"""Laplacian <-> edge-vector operators on the GPU -- drop-in surface.

Mirrors the operators of ``graph_learn/operators.py`` that are on the hot path
(same names and shapes): ``square_to_vec`` (:39-41), ``laplacian_squareform``
(:44-55), ``laplacian_squareform_vec`` (:58-70), ``laplacian_squareform_adj``
(:73-82), ``laplacian_squareform_adj_vec`` (:85-94), ``op_adj_dual`` (:176-180),
``squared_pdiffs`` (:224-226).  The spectral prox / operator-norm helpers of that
module belong to the dictionary models and are out of scope.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _ctx(ctx):
    return ctx or _lib.default_context()


def _n_from_edges(n_edges: int) -> int:
    return int(np.sqrt(2 * n_edges + 0.25) + 0.5)  # reference :60


def square_to_vec(x, *, ctx=None) -> np.ndarray:
    """(B, n, n) -> (B, E): strict upper triangle in condensed order."""
    x = _lib.as_f64(x)
    if x.ndim != 3 or x.shape[1] != x.shape[2]:
        raise ValueError(f"expected (B, n, n), got {x.shape}")
    b, n, _ = x.shape
    out = np.empty((b, n * (n - 1) // 2))
    c = _ctx(ctx)
    c.check(c._lib.gl_square_to_vec(c.handle, _lib.ptr(x), b, n, _lib.ptr(out)), "square_to_vec")
    return out


def laplacian_squareform_vec(x, *, ctx=None) -> np.ndarray:
    """(B, E) edge weights -> (B, n, n) Laplacians."""
    x = _lib.as_f64(x)
    if x.ndim != 2:
        raise ValueError(f"expected (B, E), got {x.shape}")
    b, e = x.shape
    n = _n_from_edges(e)
    if n * (n - 1) // 2 != e:
        raise ValueError(f"{e} is not a valid number of edges")
    out = np.empty((b, n, n))
    c = _ctx(ctx)
    c.check(c._lib.gl_laplacian_from_vec(c.handle, _lib.ptr(x), b, n, _lib.ptr(out)), "laplacian_from_vec")
    return out


def laplacian_squareform(x, *, ctx=None) -> np.ndarray:
    """(E,) edge weights -> (n, n) Laplacian."""
    x = _lib.as_f64(x)
    if x.ndim != 1:
        raise ValueError(f"expected (E,), got {x.shape}")
    return laplacian_squareform_vec(x[None, :], ctx=ctx)[0]


def _adjoint(y, order, ctx):
    y = _lib.as_f64(y)
    if y.ndim != 3 or y.shape[1] != y.shape[2]:
        raise ValueError(f"expected (B, n, n), got {y.shape}")
    b, n, _ = y.shape
    out = np.empty((b, n * (n - 1) // 2))
    c = _ctx(ctx)
    c.check(c._lib.gl_laplacian_adjoint(c.handle, _lib.ptr(y), b, n, order, _lib.ptr(out)), "laplacian_adjoint")
    return out


def laplacian_squareform_adj_vec(laplacians, *, ctx=None) -> np.ndarray:
    """Adjoint of laplacian_squareform_vec: (B, n, n) -> (B, E)."""
    return _adjoint(laplacians, 0, ctx)


def laplacian_squareform_adj(laplacian, *, ctx=None) -> np.ndarray:
    """Adjoint of laplacian_squareform: (n, n) -> (E,)."""
    lap = _lib.as_f64(laplacian)
    if lap.ndim != 2:
        raise ValueError(f"expected (n, n), got {lap.shape}")
    return _adjoint(lap[None], 0, ctx)[0]


def op_adj_dual(dualv, *, ctx=None) -> np.ndarray:
    """(T, n, n) -> (T, E): Y_ii + Y_jj - Y_ij - Y_ji (the dictionary models' dual adjoint)."""
    return _adjoint(dualv, 1, ctx)


def squared_pdiffs(x, *, ctx=None) -> np.ndarray:
    """(T, n) signals -> (T, E) squared node-wise differences."""
    x = _lib.as_f64(x)
    if x.ndim != 2:
        raise ValueError(f"expected (T, n), got {x.shape}")
    t, n = x.shape
    out = np.empty((t, n * (n - 1) // 2))
    c = _ctx(ctx)
    c.check(c._lib.gl_squared_pdiffs(c.handle, _lib.ptr(x), t, n, _lib.ptr(out)), "squared_pdiffs")
    return out


def theta_from_square(sq_pdists, avg_degree: int, *, ctx=None) -> float:
    """get_theta, plain branch, on a (rows, cols) matrix of squared distances (square for LogModel,
    rectangular for the block pairs of the blocks branch)."""
    sq = _lib.as_f64(sq_pdists)
    if sq.ndim != 2:
        raise ValueError(f"sq_pdists must be a matrix, got shape {sq.shape}")
    theta = C.c_double(0.0)
    c = _ctx(ctx)
    c.check(c._lib.gl_theta_from_matrix(c.handle, _lib.ptr(sq), sq.shape[0], sq.shape[1], int(avg_degree),
                                        C.byref(theta)), "get_theta")
    return theta.value

"""Laplacian-constrained Gaussian Markov Random Field on MI355X -- drop-in surface.

Mirrors ``graph_learn/lgmrf.py`` of the reference (same names, signatures, defaults,
attributes and exceptions): ``nonnegative_qp_solver`` (:19-71) and ``LGMRF`` (:74-235).
The block-coordinate descent and the inner ADMM run in HIP kernels behind the C ABI.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

try:
    from sklearn.base import BaseEstimator
except Exception:  # pragma: no cover
    from ._compat import BaseEstimator

_LAPLACIAN_SET = ["g", "generalized", "dd", "diagonally-dominant", "c", "combinatorial"]


def nonnegative_qp_solver(mat, vec, step: float = 0.1, tol: float = 1e-6, max_iter=100, *, ctx=None,
                          return_iters: bool = False):
    r"""Non-negative solution of  min 1/2 b'Qb - b'p,  b >= 0  by ADMM (reference lgmrf.py:19-71).

    Args:
        mat: symmetric (m, m) matrix Q.
        vec: offset p, shape (m,).
        step, tol, max_iter: as in the reference.
    Returns:
        the non-negative minimiser z, shape (m,).
    """
    mat = _lib.as_f64(mat)
    vec = _lib.as_f64(vec)
    if mat.ndim != 2 or mat.shape[0] != mat.shape[1] or vec.shape != (mat.shape[0],):
        raise ValueError(f"expected (m, m) and (m,), got {mat.shape} and {vec.shape}")
    c = ctx or _lib.default_context()
    z = np.empty(mat.shape[0])
    iters = C.c_int(0)
    c.check(c._lib.gl_nonneg_qp_solve(c.handle, _lib.ptr(mat), _lib.ptr(vec), mat.shape[0], float(step), float(tol),
                                      int(max_iter), _lib.ptr(z), C.byref(iters)), "nonnegative_qp_solver")
    return (z, iters.value) if return_iters else z


class LGMRF(BaseEstimator):
    """Graph learning from a Laplacian-constrained GMRF (Egilmez et al. 2017, GGL).

    Same constructor and attributes as the reference estimator (lgmrf.py:109-132).

    Args:
        norm_par (float): normalisation that controls sparsity. Default 1.
        prob_tol (float): stop when the relative Frobenius change of the Laplacian is lower. Default 1e-3.
        max_cycle (int): maximum block-coordinate cycles. Default 50.
        qp_step (float): ADMM step of the inner QP. Default 0.1.
        qp_tol (float): inner QP tolerance. Default 1e-6.
        regularization_type (int): 1 -> H = 2 Id - 11', 2 -> H = Id - 11'. Default 1.
        laplacian_set (str): only 'generalized' / 'g' is implemented (as in the reference).
        adj_mask (ndarray): prior on connectivity, default fully connected.

    Attributes:
        n_nodes_, laplacian_, inv_laplacian_, converged_ (cycle index, -1 if not converged);
        n_cycles_ (extra).
    """

    def __init__(self, norm_par: float = 1, prob_tol=1e-3, max_cycle=50, qp_step=0.1, qp_tol=1e-6,
                 regularization_type=1, laplacian_set: str = "generalized", adj_mask=None) -> None:
        self.norm_par = norm_par
        self.prob_tol = prob_tol
        self.qp_step = qp_step
        self.qp_tol = qp_tol
        self.max_cycle = max_cycle
        self.regularization_type = regularization_type
        self.laplacian_set = laplacian_set
        self.adj_mask = adj_mask

    def _check_parameters(self, x):  # reference :136-150 (incl. filling in the default mask)
        if self.laplacian_set not in _LAPLACIAN_SET:
            raise ValueError(
                f"invalid Laplacian set, got {self.laplacian_set}, must be one of [{_LAPLACIAN_SET}]")
        if self.adj_mask is None:
            self.adj_mask = np.ones_like(x)
        if self.adj_mask.shape != x.shape:
            raise ValueError("Prior shape does not match data")
        if self.regularization_type not in (1, 2):
            raise ValueError(f"regularization type can be either 1 or 2, got {self.regularization_type}")

    def fit(self, x, _y=None, *, ctx=None, path: int = 0):
        """x is the (n_nodes, n_nodes) sample covariance / kernel (not modified)."""
        x = _lib.as_f64(x)
        if x.ndim != 2 or x.shape[0] != x.shape[1]:
            raise ValueError(f"x must be a square matrix, got {x.shape}")
        self._check_parameters(x)
        self.n_nodes_ = x.shape[0]
        if self.laplacian_set not in ("g", "generalized"):
            raise NotImplementedError(f"{self.laplacian_set} is not available yet")
        n = self.n_nodes_
        c = ctx or _lib.default_context()
        mask = _lib.as_f64(self.adj_mask)
        lap = np.empty((n, n))
        inv = np.empty((n, n))
        conv = C.c_int(-1)
        cyc = C.c_int(0)
        c.check(c._lib.gl_lgmrf_fit(c.handle, _lib.ptr(x), n, float(self.norm_par), float(self.prob_tol),
                                    int(self.max_cycle), float(self.qp_step), float(self.qp_tol),
                                    int(self.regularization_type), _lib.ptr(mask), int(path), _lib.ptr(lap),
                                    _lib.ptr(inv), C.byref(conv), C.byref(cyc)), "LGMRF.fit")
        self.laplacian_ = lap
        self.inv_laplacian_ = inv
        self.converged_ = conv.value
        self.n_cycles_ = cyc.value
        return self


def time_node_updates(x, first_node: int, count: int, *, norm_par=1.0, qp_step=0.1, qp_tol=1e-6, regularization_type=1,
                      adj_mask=None, path: int = 0, ctx=None) -> float:
    """Measurement only (bench.py, tools/): seconds the node updates ``first_node .. first_node + count - 1`` of the first
    block-coordinate cycle (reference lgmrf.py:175-219) take on the device (HIP events on the context's stream).
    Returns no matrices; :meth:`LGMRF.fit` itself cannot be truncated."""
    x = _lib.as_f64(x)
    c = ctx or _lib.default_context()
    mask = None if adj_mask is None else _lib.as_f64(adj_mask)
    out = C.c_double(0.0)
    c.check(c._lib.gl_lgmrf_time_node_updates(c.handle, _lib.ptr(x), x.shape[0], float(norm_par), float(qp_step),
                                              float(qp_tol), int(regularization_type), _lib.ptr(mask), int(path),
                                              int(first_node), int(count), C.byref(out)), "time_node_updates")
    return out.value

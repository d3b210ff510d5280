"""Learning graphs from smooth signals on MI355X -- drop-in surface.

Mirrors the public surface of ``graph_learn/smooth_learning.py`` of the
reference (same names, signatures, defaults, attributes, exceptions):

  * ``gsp_learn_graph_log_degrees``  (reference :47-174)
  * ``get_theta``                    (reference :177-235, blocks=None branch)
  * ``LogModel``                     (reference :238-300)

All arithmetic runs in HIP kernels behind ``libgraphlearn_amd.so``; this module
only validates arguments and moves NumPy buffers across the C ABI.
"""

from __future__ import annotations

from typing import Optional

import numpy as np

from . import _lib

try:  # sklearn is optional: same get_params/set_params/clone behaviour when present
    from sklearn.base import BaseEstimator
except Exception:  # pragma: no cover
    from ._compat import BaseEstimator


def _n_nodes_from_edges(n_edges: int) -> int:
    # reference :133
    return int(np.round((1 + np.sqrt(1 + 8 * n_edges)) / 2))


def gsp_learn_graph_log_degrees(
    distances: np.ndarray,
    alpha: float,
    beta: float,
    edge_init: Optional[np.ndarray] = None,
    maxit: int = 1000,
    tol: float = 1e-5,
    step_size: float = 0.5,
    edge_tol: float = 1e-3,
    *,
    ctx: Optional[_lib.Context] = None,
    return_info: bool = False,
) -> np.ndarray:
    r"""Weighted adjacency from squared pairwise distances (Kalofolias log-degree model).

    Solves  min_W  sum_ij W_ij Z_ij - alpha sum_i log(sum_j W_ij) + beta/2 ||W||_F^2
    with the forward-backward-forward primal-dual iteration of the reference
    (smooth_learning.py:146-167), on the GPU.

    Args:
        distances: vectorised (condensed) squared distances, shape (E,); or a
            (B, E) batch (see below).
        alpha, beta: log-barrier and Frobenius prior constants.
        edge_init: starting point, condensed (E,). Default zeros.
        maxit, tol, step_size, edge_tol: as in the reference.
        ctx: GPU context (default: process-wide context).
        return_info: also return ``{"n_iter", "rel_primal", "rel_dual", "converged"}``.

    Returns:
        (n, n) symmetric adjacency for 1-D input (``squareform`` of the edge
        vector, :171-172); (B, E) for 2-D input (:174).

    Raises:
        ValueError: ``distances`` has more than 2 dimensions (:130-131).
    """
    distances = np.asarray(distances)
    n_dim = distances.ndim
    if n_dim == 2:
        from .batched import learn_log_degrees_batch

        return learn_log_degrees_batch(distances, alpha, beta, edge_init=edge_init, maxit=maxit, tol=tol,
                                       step_size=step_size, edge_tol=edge_tol, ctx=ctx, return_info=return_info)
    if n_dim > 2:
        raise ValueError("Distances must be vector or matrix")
    if n_dim != 1:
        raise ValueError("Distances must be vector or matrix")

    n = _n_nodes_from_edges(distances.shape[-1])
    with _lib.LogDegreeSession(n, ctx) as s:
        s.set_distances(distances)
        s.init(alpha, beta, step_size, edge_init)
        n_iter, conv, rel = s.iterate(maxit, tol)
        out = s.weights_square(edge_tol)
    if return_info:
        return out, {"n_iter": n_iter, "rel_primal": rel[0], "rel_dual": rel[1], "converged": conv}
    return out


def get_theta(sq_pdists: np.ndarray, avg_degree, blocks=None, *, ctx: Optional[_lib.Context] = None):
    """theta that gives the desired average degree (Kalofolias & Perraudin 2019).

    ``sq_pdists`` is the SQUARE (n, n) matrix of squared distances; rows are ranked with the zero
    diagonal included, as the reference does (smooth_learning.py:202-203).  With ``blocks`` (node ->
    block labels; reference :220-235, used by the clustering M-step) the plain rule is applied to
    every block pair ``sq_pdists[block_a, block_b]`` -- ``avg_degree`` may then be a dict
    ``{(a, b): degree}`` -- and the result is the condensed (E,) vector of per-edge thetas.
    """
    from . import operators

    sq = _lib.as_f64(sq_pdists)
    if blocks is None:
        if sq.ndim != 2 or sq.shape[0] != sq.shape[1]:
            raise ValueError(f"sq_pdists must be a square matrix, got {sq.shape}")
        return operators.theta_from_square(sq, int(avg_degree), ctx=ctx)
    blocks = np.asarray(blocks)
    n = len(blocks)
    thetas = np.zeros((n, n))
    labels = np.unique(blocks).tolist()
    for a_pos, a in enumerate(labels):
        for b in labels[a_pos:]:  # combinations_with_replacement(labels, 2)
            rows = np.where(blocks == a)[0]
            cols = np.where(blocks == b)[0]
            degree = avg_degree[(a, b)] if isinstance(avg_degree, dict) else avg_degree
            theta = operators.theta_from_square(sq[np.ix_(rows, cols)], int(degree), ctx=ctx)
            thetas[np.ix_(rows, cols)] = theta
            thetas[np.ix_(cols, rows)] = theta
    iu = np.triu_indices(n, k=1)
    return thetas[iu]  # squareform(thetas, checks=False)


class LogModel(BaseEstimator):
    """Graph on which the data is smooth (Kalofolias 2016 model, 2019 parameters).

    Same constructor and attributes as the reference estimator
    (smooth_learning.py:238-300).

    Parameters:
        avg_degree (int, optional): desired average number of neighbours.
        edge_init (ndarray, optional): prior on the edge weights, condensed (E,).
        maxit (int): maximum iterations. Default 1000.
        tol (float): stopping tolerance. Default 1e-5.
        step_size (float): step size in (0, 1). Default 0.5.
        edge_tol (float): weights below it are zeroed. Default 1e-3.

    Attributes:
        theta_ (float): distance rescaling.
        weights_ (ndarray): (n_nodes, n_nodes) learned adjacency.
        n_iter_, rel_primal_, rel_dual_, converged_: extra diagnostics (additive).
    """

    def __init__(
        self,
        avg_degree: int = None,
        *,
        edge_init: Optional[np.ndarray] = None,
        maxit: int = 1000,
        tol: float = 1e-5,
        step_size: float = 0.5,
        edge_tol: float = 1e-3,
    ) -> None:
        self.avg_degree = avg_degree
        self.edge_init = edge_init
        self.maxit = maxit
        self.tol = tol
        self.step_size = step_size
        self.edge_tol = edge_tol

    def _session(self, x: np.ndarray, ctx=None) -> _lib.LogDegreeSession:
        """Device-resident problem: distances (A1), theta (A2), z*theta laid out."""
        x = _lib.as_f64(x)
        if x.ndim != 2:
            raise ValueError(f"x must be (n_samples, n_nodes), got shape {x.shape}")
        s = _lib.LogDegreeSession(x.shape[1], ctx)
        try:
            theta = s.set_signals(x, self.avg_degree)
        except Exception:
            s.close()
            raise
        # reference :279-280 keeps the int 1 when avg_degree is None
        self.theta_ = 1 if self.avg_degree is None else theta
        return s

    def _initialize(self, x) -> np.ndarray:
        """Squared pairwise distances (condensed), sets ``theta_`` (reference :277-284)."""
        with self._session(x) as s:
            return s.get_distances()

    def fit(self, x: np.ndarray, ctx: Optional[_lib.Context] = None):
        """x is (n_samples, n_nodes): distances are taken between COLUMNS (:278)."""
        with self._session(x, ctx) as s:
            s.init(1, 1, self.step_size, self.edge_init)  # alpha = beta = 1 (:291-292)
            n_iter, conv, rel = s.iterate(self.maxit, self.tol)
            self.weights_ = s.weights_square(self.edge_tol)
        self.n_iter_ = n_iter
        self.converged_ = conv
        self.rel_primal_, self.rel_dual_ = rel
        return self

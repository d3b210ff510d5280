"""Synthetic smooth-signal data (host side, NumPy only).

Restates the sampling recipe of the reference so that benches and tests use the
same kind of input:

  * ``er_laplacian``    <- graph_learn/sampling/graphs.py:17-47  (sample_er_laplacian)
  * ``smooth_signals``  <- graph_learn/sampling/signals.py:7-28  (sample_lgmrf)

For small graphs ``smooth_signals`` draws exactly like the reference
(``multivariate_normal`` with ``pinv(L)``); for large ones it avoids the O(n^3)
pseudo-inverse + SVD by a Cholesky factor of ``L + eps I`` and removes the
constant component, which gives the same distribution up to O(eps).
"""

from __future__ import annotations

import numpy as np
from numpy.random import default_rng


def laplacian_from_condensed(w: np.ndarray) -> np.ndarray:
    e = w.shape[-1]
    n = int(np.sqrt(2 * e + 0.25) + 0.5)
    r, c = np.triu_indices(n, k=1)
    lap = np.zeros((n, n))
    lap[r, c] = -w
    lap[c, r] = -w
    lap[np.arange(n), np.arange(n)] = -lap.sum(axis=-1)
    return lap


def er_laplacian(n_nodes: int, edge_p: float, edge_w_min: float, edge_w_max: float, seed=None) -> np.ndarray:
    """Erdos-Renyi(n, p) graph with U[w_min, w_max] weights -> combinatorial Laplacian.

    Draw order as in the reference: first the mask uniforms, then the weights."""
    rng = default_rng(seed)
    e = (n_nodes**2 - n_nodes) // 2
    w = (rng.uniform(size=(1, e)) < edge_p) * rng.uniform(edge_w_min, edge_w_max, size=(1, e))
    return laplacian_from_condensed(w[0])


def smooth_signals(laplacian: np.ndarray, n_samples: int, seed=None, exact_below: int = 600,
                   eps: float = 1e-3) -> np.ndarray:
    """Signals X (n_samples, n_nodes) ~ N(0, pinv(L))."""
    rng = default_rng(seed)
    n = laplacian.shape[0]
    if n <= exact_below:
        return rng.multivariate_normal(mean=np.zeros(n), cov=np.linalg.pinv(laplacian), size=n_samples)
    from scipy.linalg import solve_triangular

    chol = np.linalg.cholesky(laplacian + eps * np.eye(n))        # L + eps I = C C^T
    g = rng.standard_normal((n, n_samples))
    x = solve_triangular(chol, g, lower=True, trans="T").T          # cov = (L + eps I)^-1
    x -= x.mean(axis=1, keepdims=True)                              # drop the 1/eps constant mode
    return np.ascontiguousarray(x)


def logmodel_problem(n_nodes: int, n_samples: int, seed: int, avg_degree: float | None = None):
    """(X, L): ER graph with p = avg_degree / n (default 10/n, 6/n below 1000 nodes), weights U[0.3, 2]."""
    if avg_degree is None:
        avg_degree = 10 if n_nodes >= 1000 else 6
    rng = default_rng(seed)
    lap = er_laplacian(n_nodes, avg_degree / n_nodes, 0.3, 2.0, seed=rng)
    return smooth_signals(lap, n_samples, seed=rng), lap

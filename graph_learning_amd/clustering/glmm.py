"""M-step of the graph-Laplacian mixture: one graph per cluster from responsibility-weighted signals.

Drop-in for ``graph_learn.clustering.glmm._estimate_gauss_laplacian_parameters``
(reference clustering/glmm.py:18-85), the routine behind ``GLMM._m_step`` (:222-236) and
``KGraphs`` (kgraphs.py:96-108).  Per cluster k:

    y_k = resp[:, k, None] * (x - mean_k)                      (weighted, centred signals)
    sq_dist_k[i, j] = sum_t (y_k[t, i] - y_k[t, j])^2          -> MFMA Gram kernel (rows A1 of SURVEY 8a)
    theta_k = get_theta(sq_dist_k, avg_degree, blocks)         -> radix-select kernel (row A2)
    W = delta * gsp_learn_graph_log_degrees(sq_dist * theta)   -> batched (K, E) solver, coupled stop

Everything quadratic in n runs on the GPU through the existing sessions; the (K x n) means and the
per-cluster scalars are host arithmetic.
"""

from __future__ import annotations

from typing import Optional

import numpy as np

from .. import _lib
from ..operators import laplacian_squareform_vec, square_to_vec
from ..smooth_learning import get_theta, gsp_learn_graph_log_degrees


def _estimate_gauss_laplacian_parameters(x, resp, delta: float, *, theta=None, avg_degree: Optional[int | dict] = None,
                                         blocks=None, laplacians=None, ctx=None):
    """-> (weights (K,), means (K, n), laplacians (K, n, n)); same arguments and quirks as the reference
    (``weights`` is divided by the number of NODES, glmm.py:62)."""
    if (theta is None) == (avg_degree is None):
        raise ValueError("Exactly one of theta and avg_degree should be provided")
    x = _lib.as_f64(x)
    resp = _lib.as_f64(resp)
    _n_samples, n_nodes = x.shape
    n_comp = resp.shape[1]
    if laplacians is None:
        edge_init = None
    else:  # (E, K): the batched solver takes the iterate with edges on the first axis (:54-56)
        edge_init = -square_to_vec(_lib.as_f64(laplacians), ctx=ctx).T
    weights = np.sum(resp, axis=0)
    means = resp.T @ x / weights[:, np.newaxis]
    weights = weights / n_nodes

    if avg_degree is not None and blocks is None:
        # common case (KGraphs / GLMM with a scalar degree): distances and thetas of every cluster stay on the
        # device and go straight into the batched solver
        from ..batched import learn_log_degrees_from_signals

        ys = [resp[:, k, np.newaxis] * (x - means[k]) for k in range(n_comp)]  # :65
        out, _thetas, _info = learn_log_degrees_from_signals(ys, int(avg_degree), 1, 1, edge_init=edge_init, ctx=ctx)
        return weights, means, laplacian_squareform_vec(delta * out, ctx=ctx)

    n_edges = n_nodes * (n_nodes - 1) // 2
    sq_dist = np.empty((n_comp, n_edges))
    theta_inv = None if avg_degree is None else []
    scalar_degree = avg_degree is not None and blocks is None
    with _lib.LogDegreeSession(n_nodes, ctx) as s:
        for k in range(n_comp):
            y = resp[:, k, np.newaxis] * (x - means[k])                       # :65
            th = s.set_signals(y, int(avg_degree) if scalar_degree else None)  # Gram -> distances (+ theta)
            sq_dist[k] = s.get_distances()
            if scalar_degree:
                theta_inv.append(th)
            elif avg_degree is not None:  # per-edge thetas of the block pairs (:70-71)
                sq = np.zeros((n_nodes, n_nodes))
                iu = np.triu_indices(n_nodes, k=1)
                sq[iu] = sq_dist[k]
                sq += sq.T
                theta_inv.append(get_theta(sq, avg_degree, blocks=blocks, ctx=ctx))
    if avg_degree is not None:
        theta_inv = np.array(theta_inv)
        if theta_inv.ndim < 2:
            theta_inv = theta_inv[:, np.newaxis]
    else:
        theta_inv = 1 / theta
    edge_weights = delta * gsp_learn_graph_log_degrees(sq_dist * theta_inv, alpha=1, beta=1, edge_init=edge_init,
                                                       ctx=ctx)
    return weights, means, laplacian_squareform_vec(edge_weights, ctx=ctx)


estimate_gauss_laplacian_parameters = _estimate_gauss_laplacian_parameters

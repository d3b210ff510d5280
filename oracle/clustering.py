"""Oracle (CPU restatement) of the clustering M-step and of get_theta's blocks branch.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

  * ``theta_blocks``                        <- graph_learn/smooth_learning.py:220-235
  * ``estimate_gauss_laplacian_parameters`` <- graph_learn/clustering/glmm.py:18-85
"""

from __future__ import annotations

from itertools import combinations_with_replacement

import numpy as np
from scipy.spatial.distance import squareform

from .logdeg import learn_log_degrees, theta_for_degree
from .operators import laplacian_from_vec, square_to_vec


def theta_blocks(sq, avg_degree, blocks):
    thetas = np.zeros((len(blocks), len(blocks)))
    for a, b in combinations_with_replacement(np.unique(blocks).tolist(), 2):
        rows = np.where(blocks == a)[0][:, None]
        cols = np.where(blocks == b)[0][None, :]
        deg = avg_degree[(a, b)] if isinstance(avg_degree, dict) else avg_degree
        th = theta_for_degree(sq[rows, cols], deg)
        thetas[rows, cols] = thetas[cols.T, rows.T] = th
    return squareform(thetas, checks=False)


def estimate_gauss_laplacian_parameters(x, resp, delta, *, theta=None, avg_degree=None, blocks=None, laplacians=None):
    if (theta is None) == (avg_degree is None):
        raise ValueError("Exactly one of theta and avg_degree should be provided")
    n_nodes = x.shape[1]
    edge_init = None if laplacians is None else -square_to_vec(laplacians).T
    weights = np.sum(resp, axis=0)
    means = resp.T @ x / weights[:, None]
    weights = weights / n_nodes
    y = resp[:, :, None] * (x[:, None, :] - means[None, ...])
    sq_dist = np.sum((y[..., None] - y[..., None, :]) ** 2, axis=0)
    if avg_degree is not None:
        if blocks is None:
            theta_inv = np.array([theta_for_degree(sqd, avg_degree) for sqd in sq_dist])
        else:
            theta_inv = np.array([theta_blocks(sqd, avg_degree, blocks) for sqd in sq_dist])
        if theta_inv.ndim < 2:
            theta_inv = theta_inv[:, None]
    else:
        theta_inv = 1 / theta
    w = delta * learn_log_degrees(square_to_vec(sq_dist) * theta_inv, 1, 1, edge_init=edge_init)
    return weights, means, laplacian_from_vec(w)

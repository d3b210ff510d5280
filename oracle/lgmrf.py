"""Oracle (CPU restatement) of the Laplacian-constrained GMRF solver.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

  * ``relative_error``   <- graph_learn/evaluation.py:8-25
  * ``nonneg_qp_admm``   <- graph_learn/lgmrf.py:19-71  (nonnegative_qp_solver)
  * ``LGMRFOracle``      <- graph_learn/lgmrf.py:109-235 (_check_parameters,
                            _initialize, _ggl_fit, fit)

The explicit inverse, the fancy-index gathers and the aliasing quirk of the
ADMM start (lgmrf.py:55) are reproduced as-is so that iterates match the
reference bit for bit on the same BLAS.
"""

from __future__ import annotations

import numpy as np

LAPLACIAN_SETS = ("g", "generalized", "dd", "diagonally-dominant", "c", "combinatorial")


def relative_error(y_true, y_pred):
    """||pred - true||_F / ||true||_F, or the plain norm when true ~ 0."""
    err = np.linalg.norm(y_pred - y_true)
    if np.allclose(y_true, 0):
        return err
    return err / np.linalg.norm(y_true)


def nonneg_qp_admm(mat, vec, step=0.1, tol=1e-6, max_iter=100, return_iters=False):
    """ADMM for  min 1/2 b'Qb - b'p,  b >= 0  (lgmrf.py:53-71).

    The three start vectors are one aliased array, exactly as in the reference
    (x_k = z_k = u_k = zeros): the in-place ``u += x - z`` of the first pass
    therefore also changes what ``x_k``/``z_k`` name, which only affects the
    iteration-0 relative changes (ignored, ``it > 0``).
    """
    m = mat.shape[0]
    kinv = np.linalg.inv(np.eye(m) + step * mat)
    x_old = z_old = u = np.zeros(m)
    it = -1
    for it in range(max_iter):
        x_new = kinv @ ((z_old - u) + step * vec)
        z_new = x_new + u
        z_new[z_new < 0] = 0
        u += x_new - z_new
        dx = relative_error(x_old, x_new)
        dz = relative_error(z_old, z_new)
        x_old, z_old = x_new, z_new
        if it > 0 and dx < tol and dz < tol:
            break
    return (z_old, it + 1) if return_iters else z_old


class LGMRFOracle:
    """Block-coordinate descent for the generalized-Laplacian GMRF (Egilmez GGL)."""

    def __init__(self, norm_par=1, prob_tol=1e-3, max_cycle=50, qp_step=0.1, qp_tol=1e-6,
                 regularization_type=1, laplacian_set="generalized", adj_mask=None):
        self.norm_par = norm_par
        self.prob_tol = prob_tol
        self.max_cycle = max_cycle
        self.qp_step = qp_step
        self.qp_tol = qp_tol
        self.regularization_type = regularization_type
        self.laplacian_set = laplacian_set
        self.adj_mask = adj_mask

    # lgmrf.py:136-150
    def _check(self, k):
        if self.laplacian_set not in LAPLACIAN_SETS:
            raise ValueError(f"invalid Laplacian set, got {self.laplacian_set}")
        if self.adj_mask is None:
            self.adj_mask = np.ones_like(k)
        if self.adj_mask.shape != k.shape:
            raise ValueError("Prior shape does not match data")
        if self.regularization_type not in (1, 2):
            raise ValueError(f"regularization type can be either 1 or 2, got {self.regularization_type}")

    # lgmrf.py:152-167
    def _start(self, k):
        self._check(k)
        n = self.n_nodes_ = k.shape[0]
        shift = 2.0 if self.regularization_type == 1 else 1.0
        k += shift * np.eye(n) - np.ones(n)
        self.inv_laplacian_ = np.diag(np.diag(k))
        self.laplacian_ = np.diag(1 / np.diag(k))
        self.converged_ = -1
        return k

    # lgmrf.py:169-226
    def _descend(self, k):
        n = self.n_nodes_
        idx = np.arange(n)
        self.qp_iters_ = []
        for cycle in range(self.max_cycle):
            before = self.laplacian_.copy()
            for u in idx:
                rest = idx[idx != u]
                k_u, k_uu = k[rest, u], k[u, u]
                c_u, c_uu = self.inv_laplacian_[rest, u], self.inv_laplacian_[u, u]
                m_inv = self.inv_laplacian_[rest[:, None], rest[None, :]] - np.outer(c_u, c_u) / c_uu
                free = idx[:-1][self.adj_mask[rest, u] == 1]
                col = np.zeros_like(rest, dtype=float)
                sol, nit = nonneg_qp_admm(m_inv[free[:, None], free[None, :]], k_u[free] / k_uu,
                                          step=self.qp_step, tol=self.qp_tol, return_iters=True)
                self.qp_iters_.append(nit)
                col[free] = -sol
                self.laplacian_[u, u] = (1.0 / k_uu) + col.T @ m_inv @ col
                self.laplacian_[rest, u] = col
                self.laplacian_[u, rest] = col
                g = (m_inv @ col) * k_uu
                self.inv_laplacian_[u, u] = k_uu
                self.inv_laplacian_[u, rest] = -g
                self.inv_laplacian_[rest, u] = -g
                self.inv_laplacian_[rest[:, None], rest[None, :]] = m_inv + np.outer(g, g) / k_uu
            self.n_cycles_ = cycle + 1
            if cycle > 4 and relative_error(before, self.laplacian_) < self.prob_tol:
                self.converged_ = cycle
                break
        return self

    # lgmrf.py:228-235
    def fit(self, x, _y=None):
        k = x / np.mean(x) * self.norm_par
        k = self._start(k)
        if self.laplacian_set in ("g", "generalized"):
            return self._descend(k)
        raise NotImplementedError(f"{self.laplacian_set} is not available yet")

"""Oracle (CPU restatement) of the log-degree graph-learning solver.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

Restates, with NumPy/SciPy and the same implementation strategy (two CSR
operators + dense temporaries, so that its speed is comparable to the
reference and it can serve as the ``cpu_baseline`` of ``bench.py``):

  * ``degree_operator``      <- graph_learn/smooth_learning.py:13-44  (sum_squareform)
  * ``FBFSolver``            <- graph_learn/smooth_learning.py:125-167 (loop body)
  * ``learn_log_degrees``    <- graph_learn/smooth_learning.py:47-174
  * ``theta_for_degree``     <- graph_learn/smooth_learning.py:201-219 (get_theta, blocks=None)
  * ``LogModelOracle``       <- graph_learn/smooth_learning.py:238-300

Edge order everywhere is SciPy's condensed order, edge (i, j), i < j, at
position  i*(2n-i-1)/2 + (j-i-1).
"""

from __future__ import annotations

import numpy as np
from scipy import sparse
from scipy.spatial.distance import pdist, squareform


def n_nodes_from_edges(n_edges: int) -> int:
    """Invert E = n(n-1)/2 (smooth_learning.py:133)."""
    return int(np.round((1 + np.sqrt(1 + 8 * n_edges)) / 2))


def condensed_index(i, j, n):
    """Position of edge (i, j), i < j, in the condensed vector."""
    return i * (2 * n - i - 1) // 2 + (j - i - 1)


def degree_operator(n: int):
    """Return (S, S^T) as CSR: (S w)_i = sum_j W_ij, (S^T d)_(i,j) = d_i + d_j.

    Same sparse structure as the reference builder (smooth_learning.py:13-44):
    the COO triplets are listed node-major -- for node i first the edges
    (j, i), j < i, then the edges (i, j), j > i -- and converted with the same
    SciPy calls, so matvec summation order (hence rounding) is identical.  The
    triplets are produced with vectorised index arithmetic instead of Python
    lists (the reference's O(n^2) list build is an artefact).
    """
    node = np.repeat(np.arange(n), n - 1)                       # "J" of the reference
    other = np.arange(n - 1)[None, :] + (np.arange(n - 1)[None, :] >= np.arange(n)[:, None])
    other = other.ravel()                                       # the j != i, ascending
    lo = np.minimum(node, other)
    hi = np.maximum(node, other)
    edge = lo * (2 * n - lo - 1) // 2 + (hi - lo - 1)           # "I" of the reference
    nnz = n * (n - 1)
    st = sparse.coo_array((np.ones(nnz), (edge, node)), shape=(nnz // 2, n))
    return st.T.tocsr(), st.tocsr()


class FBFSolver:
    """Forward-backward-forward primal-dual iteration, one step at a time.

    State after construction = smooth_learning.py:135-144; ``step()`` =
    one pass of the loop body :147-164 and returns the two relative changes
    (inf at the first step, :156-161).  ``distances`` is the 1-D condensed
    vector (already multiplied by theta) or the 2-D (E, B) array of the batched
    path (the caller transposes, as :128-129 does).
    """

    def __init__(self, distances, alpha, beta, step_size=0.5, edge_init=None, ops=None):
        n_edges = distances.shape[0]
        self.n = n_nodes_from_edges(n_edges)
        self.S, self.St = ops if ops is not None else degree_operator(self.n)
        self.z = distances
        self.alpha = alpha
        self.beta = beta
        # :135-137  gamma = step / (2 beta + sqrt(2 (n-1)))
        self.gamma = step_size / (2 * beta + np.sqrt(2 * (self.S.shape[0] - 1)))
        self.w = edge_init if edge_init is not None else np.zeros_like(distances)
        self.d = self.S @ self.w
        self.it = 0

    def step(self):
        g, a, b = self.gamma, self.alpha, self.beta
        w, d, z = self.w, self.d, self.z
        # forward (:147-148)
        y = (1 - g * 2 * b) * w - g * (self.St @ d)
        yb = d + g * (self.S @ w)
        # prox (:150-151)
        t = y - 2 * g * z
        p = np.where(t > 0, t, 0)
        pb = (yb - np.sqrt(yb**2 + 4 * a * g)) / 2
        # second forward (:153-154)
        q = (1 - g * 2 * b) * p - g * (self.St @ pb)
        qb = pb + g * (self.S @ p)
        # relative changes, denominators = previous iterates (:156-161)
        if self.it > 0:
            rp = np.linalg.norm(-y + q, ord=2) / np.linalg.norm(w, ord=2)
            rd = np.linalg.norm(-yb + qb, ord=2) / np.linalg.norm(d, ord=2)
        else:
            rp = rd = np.inf
        # update (:163-164)
        self.w = w - y + q
        self.d = d - yb + qb
        self.it += 1
        return rp, rd


def learn_log_degrees(
    distances,
    alpha,
    beta,
    edge_init=None,
    maxit=1000,
    tol=1e-5,
    step_size=0.5,
    edge_tol=1e-3,
    return_info=False,
    ops=None,
):
    """Restatement of gsp_learn_graph_log_degrees (smooth_learning.py:47-174).

    1-D input -> (n, n) adjacency; 2-D (B, E) input -> (B, E); >2-D raises
    ValueError (:128-131).  With ``return_info`` also returns a dict with the
    raw (un-thresholded) iterate, the dual variable, the iteration count and
    the last relative changes -- the quantities the parity tests compare.
    """
    distances = np.asarray(distances)
    ndim = distances.ndim
    if ndim == 2:
        distances = distances.T
    elif ndim > 2:
        raise ValueError("Distances must be vector or matrix")

    solver = FBFSolver(distances, alpha, beta, step_size, edge_init, ops=ops)
    rp = rd = np.inf
    for _ in range(maxit):
        rp, rd = solver.step()
        if rp < tol and rd < tol:          # :166-167
            break

    w = solver.w
    raw = w.copy()
    w[w < edge_tol] = 0                    # :169
    out = squareform(w) if ndim == 1 else w.T
    if return_info:
        return out, {"raw": raw, "dual": solver.d, "n_iter": solver.it, "rel": (rp, rd),
                     "gamma": solver.gamma}
    return out


def theta_for_degree(sq_dists_square, k):
    """get_theta, blocks=None branch (smooth_learning.py:201-219).

    Rows of the square matrix are sorted *including* the zero diagonal."""
    srt = np.sort(sq_dists_square, 1)
    head = srt[:, :k].sum(1)
    lo = np.mean((k * srt[:, k] ** 2 - head * srt[:, k]) ** (-0.5))
    hi = np.mean((k * srt[:, k - 1] ** 2 - head * srt[:, k - 1]) ** (-0.5))
    return np.sqrt(lo * hi).item()


def signal_distances(x):
    """Squared pairwise distances between the COLUMNS of x (d, n): :278."""
    return pdist(np.asarray(x).T) ** 2


class LogModelOracle:
    """LogModel.fit restated (smooth_learning.py:277-300): alpha = beta = 1."""

    def __init__(self, avg_degree=None, *, edge_init=None, maxit=1000, tol=1e-5,
                 step_size=0.5, edge_tol=1e-3):
        self.avg_degree = avg_degree
        self.edge_init = edge_init
        self.maxit = maxit
        self.tol = tol
        self.step_size = step_size
        self.edge_tol = edge_tol

    def fit(self, x):
        z = signal_distances(x)
        self.theta_ = 1 if self.avg_degree is None else theta_for_degree(squareform(z), self.avg_degree)
        self.weights_, self.info_ = learn_log_degrees(
            z * self.theta_, 1, 1, edge_init=self.edge_init, maxit=self.maxit, tol=self.tol,
            step_size=self.step_size, edge_tol=self.edge_tol, return_info=True)
        return self

"""Oracle (CPU restatement) of the graph-dictionary model with log-barrier degrees.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

  * ``init_coefficients``  <- graph_learn/dictionary/base_model.py:159-227
  * ``init_weights``       <- graph_learn/dictionary/base_model.py:229-297
  * ``coefficient_step``   <- base_model.py:355-410 (+ log_model.py:33-34 for the dual term)
  * ``weight_step``        <- base_model.py:427-461 (+ log_model.py:36-37)
  * ``dual_step``          <- log_model.py:39-50
  * ``GraphDictLogOracle`` <- base_model.py:485-527 (_fit_step), :529-587 (_single_fit),
                              :589-643 (fit), :647-676 (predict), :680-731 (score)

The model: T samples on n nodes, K atoms (graphs with condensed weights W[k], E = n(n-1)/2),
coefficients C (K x T, piecewise constant over windows of ``window_size`` samples) and one dual
vector per window (T/window x n).  One step is a primal-dual splitting update of (C, W) followed
by the log-barrier dual prox of the instantaneous degrees.  The dense (T x E) matrix of squared
pairwise differences is formed explicitly, exactly as the reference does -- this is the CPU
strategy whose results the HIP path (which never materialises it) is compared with.

Quirks kept on purpose (each is what the reference computes):
  * the "uniform"/"discrete" coefficient draws have shape (K, T) even when window_size > 1, and
    only the first T/window columns survive the repeat-and-cut (base_model.py:175,182,224-226);
  * the log_c reset writes through a strided VIEW of ``coefficients_`` (base_model.py:385-387 on
    the argument built at :497), so with window_size > 1 only the first sample of a window is
    changed in the array the weight step reads afterwards;
  * ``converged_`` is tested with ``> 0`` (base_model.py:575), so meeting the tolerance at step 0
    does not stop the loop.
"""

from __future__ import annotations

from time import time
from warnings import warn

import numpy as np
from numpy.random import RandomState

from .logdeg import degree_operator
from .lgmrf import relative_error
from .operators import square_to_vec


def squared_pdiffs(x):
    """(T, E): (x_ti - x_tj)^2 for i < j in condensed order (operators.py:224-226)."""
    return square_to_vec(x[:, :, None] - x[:, None, :]) ** 2


def autocorr(x):
    """Sample covariance of the rows of x (operators.py:229-233)."""
    xc = x - x.mean(0, keepdims=True)
    return xc.T @ xc / (x.shape[0] - 1)


def init_coefficients(x, n_atoms, window, strategy, prior, rs: RandomState):
    t = x.shape[0]
    c = np.ones((n_atoms, t // window))
    if strategy == "uniform":
        if not isinstance(prior, (list, np.ndarray)):
            c = rs.uniform(size=(n_atoms, t))
        if prior is not None:
            c = c * prior
    elif strategy == "discrete":
        c = (rs.uniform(size=(n_atoms, t)) < (prior or 0.5)).astype(float)
    elif strategy == "exp":
        c = rs.exponential(scale=prior or 1, size=c.shape)
    elif strategy == "exact":
        if prior is not None:
            c = c * prior
    elif strategy in ("k-means++d", "k-means++c"):
        from sklearn.cluster import kmeans_plusplus

        centers, _ = kmeans_plusplus(x, n_atoms, random_state=rs)
        c.fill(0)
        if strategy.endswith("d"):
            d2 = np.sum((x[None, :, :] - centers[:, None, :]) ** 2, axis=2)
            c[np.argmin(d2, axis=0), np.arange(t)] = 1
        else:
            corr = np.abs((x / np.linalg.norm(x, axis=1, keepdims=True))
                          @ (centers / np.linalg.norm(centers, axis=1, keepdims=True)).T)
            c[np.argmax(corr, axis=1), np.arange(t)] = 1
    else:
        raise ValueError(f"Invalid init_strategy for coefficients: {strategy}")
    c[c < 0] = 0
    c[c > 1] = 1
    if window > 1:
        c = np.repeat(c, window, axis=1)[:, :t]
    return c


def init_weights(x, n_atoms, strategy, prior, rs: RandomState, coefficients=None):
    t, n = x.shape
    e = (n * n - n) // 2
    w = np.ones((n_atoms, e))
    parts = strategy.split("_")
    if parts == ["uniform"]:
        if not isinstance(prior, (list, np.ndarray)):
            w = rs.uniform(size=(n_atoms, e))
        if prior is not None:
            w = w * prior
    elif parts == ["discrete"]:
        w = (rs.uniform(size=(n_atoms, e)) < (prior or 0.5)).astype(float)
    elif parts == ["exp"]:
        w = rs.exponential(scale=prior or 1, size=w.shape)
    elif parts == ["exact"]:
        if prior is not None:
            w = w * prior
    elif parts[0] == "correlation" and len(parts) >= 2:
        if parts[1] == "a":
            choices = np.argmax(coefficients, axis=0)
        elif parts[1] == "d":
            choices = rs.choice(n_atoms, size=t, replace=False)
        else:
            raise ValueError("Invalid choice of correlation for init_strategy")
        w = np.stack([autocorr(x[choices == k]) for k in range(n_atoms)])
        if parts[2:] == ["pinv"]:
            w = np.linalg.pinv(w)
        elif parts[2:]:
            raise ValueError(f"Additional option not recognized: {parts[2:]}")
        w = -square_to_vec(w)
    else:
        raise ValueError(f"Invalid init_strategy for weights: {strategy}")
    w[w < 0] = 0
    return w


class GraphDictLogOracle:
    """GraphDictLog restated; same constructor arguments and fitted attributes."""

    def __init__(self, n_atoms=1, *, window_size=1, l1_w=0, ortho_w=0, l1_c=0, log_c=0, l1_diff_c=0,
                 max_iter=1000, step=1, step_c=None, step_w=None, step_dual=None, tol=1e-3,
                 reduce_step_on_plateau=False, random_state=None, init_strategy="uniform",
                 init_strategy_c=None, init_strategy_w=None, n_init=1, weight_prior=None,
                 coefficient_prior=None):
        self.n_atoms, self.window_size = n_atoms, window_size
        self.l1_w, self.ortho_w, self.l1_c, self.log_c, self.l1_diff_c = l1_w, ortho_w, l1_c, log_c, l1_diff_c
        self.max_iter, self.tol = max_iter, tol
        self.step_c, self.step_w, self.step_dual = step_c or step, step_w or step, step_dual or step
        self.reduce_step_on_plateau = reduce_step_on_plateau
        self.random_state = random_state if isinstance(random_state, RandomState) else RandomState(random_state)
        self.init_strategy_c = init_strategy_c or init_strategy
        self.init_strategy_w = init_strategy_w or init_strategy
        self.n_init = n_init
        self.weight_prior, self.coefficient_prior = weight_prior, coefficient_prior

    # -- one primal-dual step ------------------------------------------------------------
    def coefficient_step(self, sq, c_win, dual, weights=None):
        """c_win is ``coefficients_[:, ::window]`` (a view: the log_c reset writes through it)."""
        weights = self.weights_ if weights is None else weights
        t = sq.shape[0]
        grad = weights @ sq.T
        if self.window_size > 1:
            grad = grad.reshape(self.n_atoms, -1, self.window_size).mean(2)
        grad += self.l1_c * t
        if self.log_c > 0:
            c_win[:, c_win.sum(0) < 1e-8] = self.log_c
        if self.l1_diff_c > 0:
            grad -= self.l1_diff_c * t * np.diff(np.sign(np.diff(c_win, axis=1)), axis=1, prepend=0, append=0)
        dual_term = weights @ (self._st @ dual.T)
        c_new = c_win - self.step_c / t * (dual_term + grad)
        c_new[c_new < 0] = 0
        c_new[c_new > 1] = 1
        if np.allclose(c_new, 0):
            warn("All coefficients dropped to 0", UserWarning)
            c_new.fill(self.log_c)
        return c_new

    def weight_step(self, sq, weights, dual_rep):
        t = sq.shape[0]
        grad = self.coefficients_ @ sq
        grad += self.l1_w * t
        if self.ortho_w > 0:
            grad += self.ortho_w * t * (weights.sum(0, keepdims=True) - weights)
        dual_term = self.coefficients_ @ (self._st @ dual_rep.T).T
        w_new = weights - self.step_w / t * (dual_term + grad)
        w_new[w_new < 0] = 0
        return w_new

    def dual_step(self, weights, c_win, dual):
        sigma = self.step_dual
        inst_deg = self._s @ weights.T @ c_win           # (n, T/window)
        dual = dual + sigma * inst_deg.T
        return (dual - np.sqrt(dual ** 2 + 4 * sigma)) / 2

    def fit_step(self, sq):
        ws = self.window_size
        c_new = np.repeat(self.coefficient_step(sq, self.coefficients_[:, ::ws], self.dual_), ws, axis=1)
        w_new = self.weight_step(sq, self.weights_, np.repeat(self.dual_, ws, 0))
        self.dual_ = self.dual_step(2 * w_new - self.weights_,
                                    2 * c_new[:, ::ws] - self.coefficients_[:, ::ws], self.dual_)
        dc = relative_error(self.coefficients_.ravel(), c_new.ravel())
        dw = relative_error(self.weights_.ravel(), w_new.ravel())
        self.coefficients_, self.weights_ = c_new, w_new
        return dc, dw

    # -- fitting ---------------------------------------------------------------------------
    def initialize(self, x):
        self.n_nodes_ = x.shape[1]
        self.coefficients_ = init_coefficients(x, self.n_atoms, self.window_size, self.init_strategy_c,
                                               self.coefficient_prior, self.random_state)
        self.weights_ = init_weights(x, self.n_atoms, self.init_strategy_w, self.weight_prior,
                                     self.random_state, self.coefficients_)
        self.dual_ = np.zeros((x.shape[0] // self.window_size, self.n_nodes_))
        self.converged_, self.fit_time_ = -1, -1
        self.history_ = -np.ones((self.max_iter, 2))
        self._s, self._st = degree_operator(self.n_nodes_)

    def single_fit(self, x):
        self.initialize(x)
        sq = squared_pdiffs(x)
        t0 = time()
        for i in range(self.max_iter):
            dc, dw = self.history_[i] = self.fit_step(sq)
            if dc < self.tol and dw < self.tol:
                self.converged_ = i
            if (self.reduce_step_on_plateau and i > 100 and dc >= self.history_[i - 10:i, 0].mean()
                    and dw >= self.history_[i - 10:i, 1].mean()):
                self.step_c *= 0.7
                self.step_w *= 0.7
                self.step_dual *= 0.7
            if self.converged_ > 0:
                break
        self.fit_time_ = time() - t0
        return self

    @staticmethod
    def pad(x, window):
        rem = x.shape[0] % window
        if rem > 0:
            x = np.concatenate((x, np.zeros((window - rem, x.shape[1]))))
        return x

    def fit(self, x, _y=None):
        t = x.shape[0]
        x = self.pad(np.asarray(x, dtype=float), self.window_size)
        if self.n_init == 1:
            self.single_fit(x)
        else:
            best = None
            for _ in range(self.n_init):
                self.single_fit(x)
                sc = self.score(x)
                if best is None or sc < best[0]:
                    best = (sc, self.weights_, self.coefficients_, self.dual_, self.converged_, self.fit_time_,
                            self.history_)
            _, self.weights_, self.coefficients_, self.dual_, self.converged_, self.fit_time_, self.history_ = best
        self.coefficients_ = self.coefficients_[:, :t]
        return self

    def predict(self, x):
        t, n = x.shape
        if n != self.n_nodes_:
            raise ValueError(f"Number of nodes mismatch, got {n} instead of {self.n_nodes_}")
        ws = self.window_size
        c = init_coefficients(x, self.n_atoms, ws, self.init_strategy_c, self.coefficient_prior, self.random_state)
        dual = np.zeros((t // ws, n))
        sq = squared_pdiffs(x)
        for _ in range(self.max_iter):
            c_new = np.repeat(self.coefficient_step(sq, c[:, ::ws], dual), ws, axis=1)
            dual = self.dual_step(self.weights_, 2 * c_new[:, ::ws] - c[:, ::ws], dual)
            if np.linalg.norm((c_new - c).ravel()) < self.tol:
                return c_new
            c = c_new
        return c[:, :t]

    def score(self, x, _y=None):
        from .operators import laplacian_from_vec

        if np.allclose(self.weights_, 0):
            warn("All weights are 0, log determinant is undefined", UserWarning)
            return np.inf
        loss_w = self.l1_w * np.abs(self.weights_).sum() + np.sum(
            (self.coefficients_.T @ self.weights_) * squared_pdiffs(x))
        if self.ortho_w > 0:
            loss_w += self.ortho_w * (np.sum(self.weights_.T @ self.weights_) - np.linalg.norm(self.weights_) ** 2)
        loss_c = self.l1_c * np.abs(self.coefficients_).sum()
        if self.log_c > 0:
            loss_c -= self.log_c * np.sum(np.log(self.coefficients_.sum(0)))
        if self.l1_diff_c > 0:
            loss_c -= self.l1_diff_c * np.linalg.norm(
                np.diff(np.sign(np.diff(self.coefficients_, axis=1)), axis=1, prepend=0, append=0))
        eig = np.linalg.eigvalsh(np.einsum("kc,knm->cnm", self.coefficients_, laplacian_from_vec(self.weights_)))
        loggdet = np.sum(np.log(np.where(eig > 0, eig, 1)).sum(-1))
        return loss_w + (loss_c - loggdet)

"""``GraphDictLog`` -- graph dictionary with log-barrier degrees, on the GPU.

Drop-in for ``graph_learn.dictionary.GraphDictLog`` (reference dictionary/log_model.py:14-50 on top
of dictionary/base_model.py:28-731): same constructor, same fitted attributes
(``coefficients_`` (K, T), ``weights_`` (K, E), ``dual_`` (T/window, n), ``n_nodes_``, ``converged_``,
``history_``, ``fit_time_``), same ``fit`` / ``predict`` / ``score``.

Division of labour: the initialisation strategies (base_model.py:159-297) draw from the legacy
``numpy.random.RandomState`` stream on the host, exactly like the reference, so a given seed gives
the same start; every step of the optimisation loop (:485-527) and of ``predict`` (:647-676) runs in
HIP kernels behind ``gl_graphdict_*``; ``score`` (:680-731, an eigendecomposition per sample) is an
evaluation helper and stays on the host.
"""

from __future__ import annotations

from time import time
from typing import Callable
from warnings import warn

import numpy as np
from numpy.random import RandomState

from .. import _lib
from .._compat import BaseEstimator as _ShimEstimator

try:  # sklearn is optional on the host
    from sklearn.base import BaseEstimator
except Exception:  # pragma: no cover
    BaseEstimator = _ShimEstimator


def _triu_vec(mats):
    """(B, n, n) -> (B, E) strict upper triangle, condensed order (operators.py:39-41)."""
    n = mats.shape[-1]
    i, j = np.triu_indices(n, k=1)
    return mats[..., i, j]


def _autocorr(x):
    """operators.py:229-233"""
    xc = x - x.mean(0, keepdims=True)
    return xc.T @ xc / (x.shape[0] - 1)


class GraphDictLog(BaseEstimator):
    """Graph dictionary learning with the log-barrier on the instantaneous degrees.

    Args: see the reference (dictionary/base_model.py:31-78).  ``n_atoms`` <= 16.
    Extra, optional: ``ctx`` argument of ``fit`` / ``predict`` (a ``graph_learning_amd.Context``).
    """

    def __init__(self, n_atoms=1, *, window_size: int = 1, l1_w: float = 0, ortho_w: float = 0, l1_c: float = 0,
                 log_c: float = 0, l1_diff_c: float = 0, max_iter: int = 1000, step: float = 1, step_c: float = None,
                 step_w: float = None, step_dual: float = None, tol: float = 1e-3, reduce_step_on_plateau: bool = False,
                 random_state: RandomState = None, init_strategy: str = "uniform", init_strategy_c: str = None,
                 init_strategy_w: str = None, n_init: int = 1, weight_prior=None, coefficient_prior=None,
                 early_stop_keyboard: bool = False, compute_dtype: str = "float64") -> None:
        super().__init__()
        # extra to the reference's signature (additive): "float32" runs the two O(K T E) contractions of a step in
        # packed fp32 on the device (BASELINE configs[4]); the state, every fold and the prox stay float64
        if compute_dtype not in ("float64", "float32"):
            raise ValueError(f"compute_dtype must be 'float64' or 'float32', got {compute_dtype!r}")
        self.compute_dtype = compute_dtype
        self.n_atoms = n_atoms
        self.window_size = window_size
        self.ortho_w = ortho_w
        self.l1_w = l1_w
        self.l1_c = l1_c
        self.log_c = log_c
        self.l1_diff_c = l1_diff_c
        self.max_iter = max_iter
        self.step = step
        self.step_c = step_c or step
        self.step_w = step_w or step
        self.step_dual = step_dual or step
        self.tol = tol
        self.reduce_step_on_plateau = reduce_step_on_plateau
        self.random_state = random_state if isinstance(random_state, RandomState) else RandomState(random_state)
        self.init_strategy = init_strategy
        self.init_strategy_c = init_strategy_c or init_strategy
        self.init_strategy_w = init_strategy_w or init_strategy
        self.n_init = n_init
        self.weight_prior = weight_prior
        self.coefficient_prior = coefficient_prior
        self.early_stop_keyboard = early_stop_keyboard

    # ---- initialisation (host, RandomState-compatible) -------------------------------------------
    def _init_coefficients(self, x):
        """base_model.py:159-227"""
        t = x.shape[0]
        k, ws, prior, rs = self.n_atoms, self.window_size, self.coefficient_prior, self.random_state
        c = np.ones((k, t // ws))
        strat = self.init_strategy_c
        if strat == "uniform":
            if not isinstance(prior, (list, np.ndarray)):
                c = rs.uniform(size=(k, t))
            if prior is not None:
                c = c * prior
        elif strat == "discrete":
            c = (rs.uniform(size=(k, t)) < (prior or 0.5)).astype(float)
        elif strat == "exp":
            c = rs.exponential(scale=prior or 1, size=c.shape)
        elif strat == "exact":
            if prior is not None:
                c = c * prior
        elif strat in ("k-means++d", "k-means++c"):
            from sklearn.cluster import kmeans_plusplus

            centers, _ = kmeans_plusplus(x, k, random_state=rs)
            c.fill(0)
            if strat == "k-means++d":
                c[np.argmin(((x[None] - centers[:, None]) ** 2).sum(2), axis=0), np.arange(t)] = 1
            else:
                xn = x / np.linalg.norm(x, axis=1, keepdims=True)
                cn = centers / np.linalg.norm(centers, axis=1, keepdims=True)
                c[np.argmax(np.abs(xn @ cn.T), axis=1), np.arange(t)] = 1
        else:
            raise ValueError(f"Invalid init_strategy for coefficients: {strat}")
        c[c < 0] = 0
        c[c > 1] = 1
        if ws > 1:
            c = np.repeat(c, ws, axis=1)[:, :t]
        return c

    def _init_weigths(self, x):  # sic: the reference's spelling (base_model.py:229)
        """base_model.py:229-297"""
        t, n = x.shape
        k, prior, rs = self.n_atoms, self.weight_prior, self.random_state
        e = (n * n - n) // 2
        w = np.ones((k, e))
        parts = self.init_strategy_w.split("_")
        if parts == ["uniform"]:
            if not isinstance(prior, (list, np.ndarray)):
                w = rs.uniform(size=(k, e))
            if prior is not None:
                w = w * prior
        elif parts == ["discrete"]:
            w = (rs.uniform(size=(k, e)) < (prior or 0.5)).astype(float)
        elif parts == ["exp"]:
            w = rs.exponential(scale=prior or 1, size=w.shape)
        elif parts == ["exact"]:
            if prior is not None:
                w = w * prior
        elif parts[0] == "correlation" and len(parts) >= 2:
            if parts[1] == "a":
                choices = np.argmax(self.coefficients_, axis=0)
            elif parts[1] == "d":
                choices = rs.choice(k, size=t, replace=False)
            else:
                raise ValueError("Invalid choice of correlation for init_strategy")
            w = np.stack([_autocorr(x[choices == a]) for a in range(k)])
            if parts[2:] == ["pinv"]:
                w = np.linalg.pinv(w)
            elif parts[2:]:
                raise ValueError(f"Additional option not recognized: {parts[2:]}")
            w = -_triu_vec(w)
        else:
            raise ValueError(f"Invalid init_strategy for weights: {self.init_strategy_w}")
        w[w < 0] = 0
        return w

    def _init_dual(self, x):
        """log_model.py:23-24"""
        return np.zeros((x.shape[0] // self.window_size, self.n_nodes_))

    def _initialize(self, x) -> None:
        """base_model.py:313-333"""
        self.n_nodes_ = x.shape[1]
        self.coefficients_ = self._init_coefficients(x)
        self.weights_ = self._init_weigths(x)
        self.dual_ = self._init_dual(x)
        self.converged_ = -1
        self.fit_time_ = -1
        self.history_ = self._history_frame(-np.ones((self.max_iter, 2)))

    def _history_frame(self, arr):
        try:
            import pandas as pd

            return pd.DataFrame(data=arr, columns=["coefficient_change", "weight_change"], index=np.arange(len(arr)))
        except Exception:  # pragma: no cover - pandas absent
            return arr

    # ---- fitting -----------------------------------------------------------------------------------
    def _params(self):
        return dict(l1_w=self.l1_w, ortho_w=self.ortho_w, l1_c=self.l1_c, log_c=self.log_c, l1_diff_c=self.l1_diff_c,
                    step_c=self.step_c, step_w=self.step_w, step_dual=self.step_dual)

    def _single_fit(self, x, _y=None, *, callback: Callable = None, ctx=None):
        """base_model.py:529-587.  Without a callback the whole loop is one C call; with one, the
        loop is driven step by step from here so that the callback sees every iterate."""
        self._initialize(x)
        with _lib.GraphDictSession(self.n_nodes_, x.shape[0], self.n_atoms, self.window_size, ctx) as s:
            if self.compute_dtype == "float32":
                s.set_precision(32)
            s.set_signals(x)
            s.set_params(**self._params())
            s.set_state(self.coefficients_, self.weights_, self.dual_)
            if callback is not None:
                callback(self, 0)
            start = time()
            hist = -np.ones((self.max_iter, 2))
            if callback is None:
                hist, ran, conv, flags = s.fit(self.max_iter, self.tol, self.reduce_step_on_plateau)
                self._warn(flags)
                self.converged_ = int(conv)
            else:
                for i in range(self.max_iter):
                    try:
                        s.set_params(**self._params())
                        h1, _, _, flags = s.fit(1, -1.0, False)  # one step; the stopping logic stays here
                        self._warn(flags & 1)
                        hist[i] = dc, dw = h1[0]
                        self.coefficients_, self.weights_, self.dual_ = s.get_state()
                        if dc < self.tol and dw < self.tol:
                            self.converged_ = i
                        if (self.reduce_step_on_plateau and i > 100 and dc >= hist[i - 10:i, 0].mean()
                                and dw >= hist[i - 10:i, 1].mean()):
                            warn(f"Divergence detected at step {i}, reducing step size", UserWarning)
                            self.step_c *= 0.7
                            self.step_w *= 0.7
                            self.step_dual *= 0.7
                        callback(self, i + 1)
                        if self.converged_ > 0:
                            break
                    except KeyboardInterrupt:
                        if self.early_stop_keyboard:
                            warn("Keyboard interrupt, stopping early")
                            break
                        raise
            self.coefficients_, self.weights_, self.dual_ = s.get_state()
            if callback is None:
                self.step_c, self.step_w, self.step_dual = s.steps()
            self.fit_time_ = time() - start
        self.history_ = self._history_frame(hist)
        return self

    @staticmethod
    def _warn(flags):
        if flags & 1:
            warn("All coefficients dropped to 0", UserWarning)
        if flags & 2:
            warn("Divergence detected, reducing step size", UserWarning)

    def fit(self, x, _y=None, *, callback: Callable = None, ctx=None):
        """base_model.py:589-643"""
        x = _lib.as_f64(x)
        if x.ndim != 2:
            raise ValueError(f"x must be (n_samples, n_nodes), got shape {x.shape}")
        n_samples = x.shape[0]
        pad = n_samples % self.window_size
        if pad > 0:
            x = np.concatenate((x, np.zeros((self.window_size - pad, x.shape[1]))))
        if self.n_init == 1:
            self._single_fit(x, callback=callback, ctx=ctx)
        else:
            best = None
            for _ in range(self.n_init):
                self._single_fit(x, callback=callback, ctx=ctx)
                sc = self.score(x)
                if best is None or sc < best[0]:
                    best = (sc, self.weights_, self.coefficients_, self.dual_, self.converged_, self.fit_time_,
                            self.history_)
            _, self.weights_, self.coefficients_, self.dual_, self.converged_, self.fit_time_, self.history_ = best
        self.coefficients_ = self.coefficients_[:, :n_samples]
        return self

    def predict(self, x, ctx=None):
        """base_model.py:647-676: coefficients of new signals under the learned atoms."""
        x = _lib.as_f64(x)
        n_samples, n_nodes = x.shape
        if n_nodes != self.n_nodes_:
            raise ValueError(f"Number of nodes mismatch, got {n_nodes} instead of {self.n_nodes_}")
        if n_samples % self.window_size:
            raise ValueError("predict: n_samples must be a multiple of window_size")  # the reference's reshape fails
        c0 = self._init_coefficients(x)
        with _lib.GraphDictSession(n_nodes, n_samples, self.n_atoms, self.window_size, ctx) as s:
            if self.compute_dtype == "float32":
                s.set_precision(32)
            s.set_signals(x)
            s.set_params(**self._params())
            s.set_state(c0, self.weights_, None)
            out, _, flags = s.predict(self.max_iter, self.tol)
        self._warn(flags)
        return out[:, :n_samples]

    def score(self, x, _y=None) -> float:
        """base_model.py:680-731 (host: one symmetric eigendecomposition per sample)."""
        x = _lib.as_f64(x)
        if np.allclose(self.weights_, 0):
            warn("All weights are 0, log determinant is undefined", UserWarning)
            return np.inf
        n = self.n_nodes_
        iu = np.triu_indices(n, k=1)
        smooth = 0.0
        cw = self.coefficients_.T @ self.weights_  # (T, E): instantaneous weights
        for t0 in range(0, x.shape[0], 256):  # chunked: never the whole (T, E) difference matrix
            xb = x[t0:t0 + 256]
            smooth += np.sum(cw[t0:t0 + 256] * (xb[:, iu[0]] - xb[:, iu[1]]) ** 2)
        weight_loss = self.l1_w * np.abs(self.weights_).sum() + smooth
        if self.ortho_w > 0:
            weight_loss += self.ortho_w * (np.sum(self.weights_.T @ self.weights_) - np.linalg.norm(self.weights_) ** 2)
        coefficient_loss = self.l1_c * np.abs(self.coefficients_).sum()
        if self.log_c > 0:
            coefficient_loss -= self.log_c * np.sum(np.log(self.coefficients_.sum(0)))
        if self.l1_diff_c > 0:
            warn("Check diff loss", UserWarning)
            coefficient_loss -= self.l1_diff_c * np.linalg.norm(
                np.diff(np.sign(np.diff(self.coefficients_, axis=1)), axis=1, prepend=0, append=0))
        # Sum of log generalised determinants (:722-729).  The null eigenvalue of every instantaneous
        # Laplacian comes out of eigvalsh as +-1e-16 and only its positive occurrences are counted
        # (log ~ -37 each), so this term depends on the rounding of the Laplacians themselves: they
        # are built with the reference's own sequence of operations (operators.py:58-70, einsum).
        lap = np.zeros((self.n_atoms, n, n))
        lap[:, iu[0], iu[1]] = -self.weights_
        lap += np.transpose(lap, (0, 2, 1))
        ar = np.arange(n)
        lap[:, ar, ar] = -lap.sum(axis=-1)
        chunk = max(1, min(cw.shape[0], (1 << 27) // (n * n)))  # <= 1 GiB of Laplacians at a time
        loggdet = 0.0
        for t0 in range(0, cw.shape[0], chunk):
            eig = np.linalg.eigvalsh(np.einsum("kc,knm->cnm", self.coefficients_[:, t0:t0 + chunk], lap))
            loggdet += np.sum(np.log(np.where(eig > 0, eig, 1)).sum(-1))
        return weight_loss + (coefficient_loss - loggdet)

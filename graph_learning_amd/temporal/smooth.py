"""``WindowLogModel`` -- LogModel on non-overlapping windows (reference graph_learn/temporal/smooth.py:12-61).

A thin caller of the hot path: windowed squared distances (one MFMA Gram per window), one theta per
window, then the batched solver with its coupled stopping rule.
"""

from __future__ import annotations

from typing import Optional

import numpy as np

from .. import _lib
from ..smooth_learning import LogModel, gsp_learn_graph_log_degrees


class WindowLogModel(LogModel):
    """Learn a sequence of graphs from signals by applying LogModel on windows.

    Same constructor as the reference (``random_state`` is accepted and unused there too).
    ``theta_`` has shape (n_windows, 1) (or is 1), ``weights_`` shape (n_windows, E).
    """

    def __init__(self, window_size: int = 10, avg_degree: int = None, *, edge_init: Optional[np.ndarray] = None,
                 maxit: int = 1000, tol: float = 1e-5, step_size: float = 0.5, edge_tol: float = 1e-3,
                 random_state=None) -> None:
        super().__init__(avg_degree=avg_degree, edge_init=edge_init, maxit=maxit, tol=tol, step_size=step_size,
                         edge_tol=edge_tol)
        self.window_size = window_size
        self.random_state = random_state  # unused, kept for compatibility with the reference

    def _initialize(self, x, ctx=None) -> np.ndarray:
        """(n_windows, E) squared distances; sets ``theta_`` (reference :41-61)."""
        x = _lib.as_f64(x)
        if x.ndim != 2:
            raise ValueError(f"x must be (n_samples, n_nodes), got shape {x.shape}")
        n_samples, n_nodes = x.shape
        if (pad := n_samples % self.window_size) > 0:  # zero rows complete the last window (:44-45)
            x = np.vstack((x, np.zeros((self.window_size - pad, n_nodes))))
        windows = x.reshape(-1, self.window_size, n_nodes)
        dists, thetas = [], []
        with _lib.LogDegreeSession(n_nodes, ctx) as s:
            for win in windows:
                thetas.append(s.set_signals(win, self.avg_degree))
                dists.append(s.get_distances())
        self.theta_ = 1 if self.avg_degree is None else np.array(thetas)[:, None]
        return np.stack(dists)

    def fit(self, x, ctx=None):
        """Same result as the reference's ``fit`` (``_initialize`` + the 2-D solver, :41-61 and
        smooth_learning.py:286-300), with the windows' distances and thetas kept on the device."""
        from ..batched import learn_log_degrees_from_signals

        x = _lib.as_f64(x)
        if x.ndim != 2:
            raise ValueError(f"x must be (n_samples, n_nodes), got shape {x.shape}")
        n_samples, n_nodes = x.shape
        if (pad := n_samples % self.window_size) > 0:  # zero rows complete the last window (:44-45)
            x = np.vstack((x, np.zeros((self.window_size - pad, n_nodes))))
        windows = x.reshape(-1, self.window_size, n_nodes)
        out, thetas, info = learn_log_degrees_from_signals(windows, self.avg_degree, 1, 1, edge_init=self.edge_init,
                                                          maxit=self.maxit, tol=self.tol, step_size=self.step_size,
                                                          edge_tol=self.edge_tol, ctx=ctx)
        self.theta_ = 1 if self.avg_degree is None else thetas[:, None]
        self.weights_ = out
        self.n_iter_ = info["n_iter"]
        self.converged_ = info["converged"]
        self.rel_primal_, self.rel_dual_ = info["rel_primal"], info["rel_dual"]
        return self

"""Oracle (CPU restatement) of the Laplacian <-> edge-vector operators.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

  * ``square_to_vec``        <- graph_learn/operators.py:39-41
  * ``laplacian_from_vec``   <- graph_learn/operators.py:58-70 (and :44-55)
  * ``laplacian_adjoint``    <- graph_learn/operators.py:85-94, :176-180 (and :73-82)
  * ``squared_pdiffs``       <- graph_learn/operators.py:224-226

Written in closed index form (no squareform round trips); batched over the
leading axis like the ``*_vec`` variants of the reference.
"""

from __future__ import annotations

import numpy as np


def _n_from_e(e):
    return int(np.sqrt(2 * e + 0.25) + 0.5)


def square_to_vec(x):
    """(B, n, n) -> (B, E): strict upper triangle in condensed order."""
    r, c = np.triu_indices(x.shape[-1], k=1)
    return x[:, r, c]


def laplacian_from_vec(w):
    """(B, E) edge weights -> (B, n, n) Laplacians L = D - W."""
    w = np.asarray(w)
    n = _n_from_e(w.shape[-1])
    r, c = np.triu_indices(n, k=1)
    out = np.zeros((w.shape[0], n, n), dtype=w.dtype)
    out[:, r, c] = -w
    out[:, c, r] = -w
    d = np.arange(n)
    out[:, d, d] = -out.sum(axis=-1)
    return out


def laplacian_adjoint(y):
    """(B, n, n) -> (B, E):  (L* Y)_(i,j) = Y_ii + Y_jj - Y_ij - Y_ji."""
    y = np.asarray(y)
    diag = np.diagonal(y, axis1=-2, axis2=-1)
    full = -y - np.transpose(y, (0, 2, 1)) + diag[:, :, None] + diag[:, None, :]
    return square_to_vec(full)


def squared_pdiffs(x):
    """(T, n) -> (T, E):  (x_ti - x_tj)^2."""
    x = np.asarray(x)
    return square_to_vec(x[:, :, None] - x[:, None, :]) ** 2

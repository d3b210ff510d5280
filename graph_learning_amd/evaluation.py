"""``relative_error`` (reference graph_learn/evaluation.py:8-25) on the GPU.

It is the stopping metric of LGMRF and of its inner QP; the edge scores of that
module are outside the accelerated hot path."""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def relative_error(y_true, y_pred, *, ctx=None) -> float:
    r"""||y_pred - y_true||_F / ||y_true||_F, or the plain norm when y_true is allclose to 0."""
    a = _lib.as_f64(y_true)
    b = _lib.as_f64(y_pred)
    if a.shape != b.shape:
        raise ValueError(f"shapes differ: {a.shape} vs {b.shape}")
    c = ctx or _lib.default_context()
    out = C.c_double(0.0)
    c.check(c._lib.gl_relative_error(c.handle, _lib.ptr(a), _lib.ptr(b), a.size, C.byref(out)), "relative_error")
    return out.value

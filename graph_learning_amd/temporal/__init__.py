"""Time-varying graphs from smooth signals (the accelerated subset of graph_learn/temporal)."""

from .smooth import WindowLogModel

__all__ = ["WindowLogModel"]

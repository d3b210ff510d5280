"""Graph-dictionary models -- only ``GraphDictLog``.

The other dictionary models of the reference (``GraphDictExact``, ``GraphDictSpectral``,
``GraphDictHier``, ``FixedCoefficients``, ``FixedWeights``) use an eigendecomposition prox and are
out of scope.
"""

from .log_model import GraphDictLog

__all__ = ["GraphDictLog"]

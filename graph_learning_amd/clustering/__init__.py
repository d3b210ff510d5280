"""Clustering M-step on the GPU.

Only the parameter estimation shared by ``GLMM._m_step`` and ``KGraphs`` is provided
(``estimate_gauss_laplacian_parameters``); the EM / k-means drivers themselves are callers of the
hot path and stay out of scope.
"""

from .glmm import _estimate_gauss_laplacian_parameters, estimate_gauss_laplacian_parameters

__all__ = ["estimate_gauss_laplacian_parameters", "_estimate_gauss_laplacian_parameters"]

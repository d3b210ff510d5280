"""Clustering M-step and get_theta(blocks=...) on the GPU against the golden vectors of the reference.

Tolerance: the distances come from the fp64 MFMA Gram kernel (not from direct differences), the
batched solver stops on spectral norms of B x B Gram matrices: Laplacians agree to 1e-8 where the
stop lands on the same iteration; entries within 1e-6 of the edge threshold are excluded.
"""

import numpy as np
import pytest

from graph_learning_amd import get_theta
from graph_learning_amd.clustering import estimate_gauss_laplacian_parameters

pytestmark = pytest.mark.gpu

CASES = {
    "deg3": dict(avg_degree=3),
    "theta05": dict(theta=0.5),
    "blocks_int": dict(avg_degree=2, blocks=True),
    "blocks_dict": dict(avg_degree={(0, 0): 2, (0, 1): 3, (1, 1): 2}, blocks=True),
}


def kwargs(g, tag):
    kw = dict(CASES[tag])
    if kw.get("blocks"):
        kw["blocks"] = g["blocks"]
    return kw


def close_laplacians(got, want, delta, edge_tol=1e-3):
    off = ~np.eye(want.shape[-1], dtype=bool)
    near = np.abs(np.abs(want) / delta - edge_tol) < 1e-6     # raw weights next to the threshold
    ok = np.isclose(got, want, rtol=1e-7, atol=1e-8) | (near & off)
    diag_touched = near.any(axis=-1)                           # a degree next to a flipped edge
    ok |= np.eye(want.shape[-1], dtype=bool) & diag_touched[..., None]
    assert ok.all(), f"max abs diff {np.max(np.abs(got - want))}"


@pytest.mark.parametrize("tag", sorted(CASES))
def test_mstep_golden(golden, tag):
    g = golden("clustering")
    w, m, lap = estimate_gauss_laplacian_parameters(g["x"], g["resp"], float(g["delta"]), **kwargs(g, tag))
    np.testing.assert_allclose(w, g[f"{tag}_weights"], rtol=1e-12)
    np.testing.assert_allclose(m, g[f"{tag}_means"], rtol=1e-12, atol=1e-14)
    close_laplacians(lap, g[f"{tag}_laplacians"], float(g["delta"]))


def test_mstep_warm_start_golden(golden):
    g = golden("clustering")
    k = g["resp"].shape[1]
    w, m, lap = estimate_gauss_laplacian_parameters(g["x"], g["resp"] * 0.9 + 0.1 / k, 2.0, avg_degree=3,
                                                    laplacians=g["deg3_laplacians"])
    close_laplacians(lap, g["warm_laplacians"], 2.0)


def test_theta_blocks_golden(golden):
    g = golden("clustering")
    np.testing.assert_allclose(get_theta(g["theta_z"], 2, blocks=g["blocks"]), g["theta_blocks_int"], rtol=1e-12)
    np.testing.assert_allclose(get_theta(g["theta_z"], {(0, 0): 2, (0, 1): 3, (1, 1): 2}, blocks=g["blocks"]),
                               g["theta_blocks_dict"], rtol=1e-12)


def test_mstep_argument_check(golden):
    g = golden("clustering")
    with pytest.raises(ValueError):
        estimate_gauss_laplacian_parameters(g["x"], g["resp"], 1.0)
    with pytest.raises(ValueError):
        estimate_gauss_laplacian_parameters(g["x"], g["resp"], 1.0, theta=1.0, avg_degree=2)

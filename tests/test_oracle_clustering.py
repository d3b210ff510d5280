"""Oracle of the clustering M-step / get_theta blocks branch against golden vectors of the reference. CPU only."""

import numpy as np
import pytest

from oracle import clustering as ocl

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


@pytest.mark.parametrize("tag", sorted(CASES))
def test_mstep_golden(golden, tag):
    g = golden("clustering")
    w, m, lap = ocl.estimate_gauss_laplacian_parameters(g["x"], g["resp"], float(g["delta"]), **kwargs(g, tag))
    np.testing.assert_allclose(w, g[f"{tag}_weights"], rtol=1e-12)
    np.testing.assert_allclose(m, g[f"{tag}_means"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(lap, g[f"{tag}_laplacians"], rtol=1e-9, atol=1e-11)


def test_mstep_warm_start_golden(golden):
    g = golden("clustering")
    k = g["resp"].shape[1]
    w, m, lap = ocl.estimate_gauss_laplacian_parameters(g["x"], g["resp"] * 0.9 + 0.1 / k, 2.0, avg_degree=3,
                                                        laplacians=g["deg3_laplacians"])
    np.testing.assert_allclose(lap, g["warm_laplacians"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(m, g["warm_means"], rtol=1e-12, atol=1e-14)


def test_theta_blocks_golden(golden):
    g = golden("clustering")
    np.testing.assert_allclose(ocl.theta_blocks(g["theta_z"], 2, g["blocks"]), g["theta_blocks_int"], rtol=1e-13)
    np.testing.assert_allclose(ocl.theta_blocks(g["theta_z"], {(0, 0): 2, (0, 1): 3, (1, 1): 2}, g["blocks"]),
                               g["theta_blocks_dict"], rtol=1e-13)


def test_mstep_argument_check(golden):
    g = golden("clustering")
    with pytest.raises(ValueError):
        ocl.estimate_gauss_laplacian_parameters(g["x"], g["resp"], 1.0)
    with pytest.raises(ValueError):
        ocl.estimate_gauss_laplacian_parameters(g["x"], g["resp"], 1.0, theta=1.0, avg_degree=2)

"""GPU operators vs the reference's own property tests (tests/test_operators.py:25-72,
209-226 restated: same seeds, shapes and np.allclose defaults) and golden outputs."""

import numpy as np
import pytest
from numpy.random import default_rng

from graph_learning_amd import get_theta
from graph_learning_amd import operators as ops

pytestmark = pytest.mark.gpu

SEEDS = range(0, 1000, 37)  # a spread of the reference's 1000 seeds (each is a host<->device round trip)


def test_laplacian_squareform_vec_matches_scalar_form():
    for seed in SEEDS:
        rng = default_rng(seed)
        n_nodes = rng.integers(20, 100)
        n_components = rng.integers(1, 10)
        weights = rng.standard_normal((n_components, (n_nodes * (n_nodes - 1)) // 2))
        assert np.allclose([ops.laplacian_squareform(x) for x in weights], ops.laplacian_squareform_vec(weights))


def test_laplacian_squareform_adj():
    for seed in SEEDS:
        rng = default_rng(seed)
        n_nodes = rng.integers(20, 100)
        weights = rng.standard_normal((n_nodes * (n_nodes - 1)) // 2)
        lapl = rng.standard_normal((n_nodes, n_nodes))
        assert np.allclose(np.sum(lapl * ops.laplacian_squareform(weights)),
                           np.sum(weights * ops.laplacian_squareform_adj(lapl)))


def test_laplacian_squareform_adj_vec():
    for seed in SEEDS:
        rng = default_rng(seed)
        n_nodes = rng.integers(20, 100)
        n_components = rng.integers(1, 10)
        weights = rng.standard_normal((n_components, (n_nodes * (n_nodes - 1)) // 2))
        dual = rng.standard_normal((n_components, n_nodes, n_nodes))
        assert np.allclose(np.sum(weights * ops.laplacian_squareform_adj_vec(dual)),
                           np.sum(dual * ops.laplacian_squareform_vec(weights)))


def test_squared_pdiffs_smoothness_identity():
    """tests/test_operators.py:209-226 (dictionary_smoothness) with the GPU pairwise diffs."""
    for seed in SEEDS:
        rng = default_rng(seed)
        n_components = rng.integers(1, 5)
        n_nodes = rng.integers(10, 20)
        n_samples = rng.integers(50, 500)
        coefficients = rng.standard_normal((n_components, n_samples))
        weights = rng.standard_normal((n_components, (n_nodes * (n_nodes - 1)) // 2))
        signals = rng.standard_normal(size=(n_samples, n_nodes))
        laplacians = ops.laplacian_squareform_vec(weights)
        smooth = np.sum((coefficients.T @ weights) * ops.squared_pdiffs(signals))
        assert np.allclose(smooth, np.einsum("knm,kt,tn,tm->", laplacians, coefficients, signals, signals))


@pytest.mark.parametrize("seed", [3, 11, 1])
def test_golden_outputs(golden, seed):
    g = golden("operators")
    np.testing.assert_allclose(ops.laplacian_squareform_vec(g[f"w_{seed}"]), g[f"lap_{seed}"], rtol=0, atol=1e-12)
    off = ~np.eye(g[f"lap_{seed}"].shape[-1], dtype=bool)
    assert np.array_equal(ops.laplacian_squareform_vec(g[f"w_{seed}"])[:, off], g[f"lap_{seed}"][:, off])
    assert np.array_equal(ops.laplacian_squareform_adj_vec(g[f"dual_{seed}"]), g[f"adj_{seed}"])
    assert np.array_equal(ops.op_adj_dual(g[f"dual_{seed}"]), g[f"opadj_{seed}"])
    assert np.array_equal(ops.square_to_vec(g[f"dual_{seed}"]), g[f"s2v_{seed}"])


def test_squared_pdiffs_golden(golden):
    g = golden("operators")
    assert np.array_equal(ops.squared_pdiffs(g["signals"]), g["sq_pdiffs"])


def test_get_theta_golden(golden):
    g = golden("theta")
    for k, ref in zip(g["k"], g["theta"]):
        got = get_theta(g["z"], int(k))
        assert (np.isinf(ref) and np.isinf(got)) or got == pytest.approx(ref, rel=1e-12)
    for k, ref in zip(g["k_ties"], g["theta_ties"]):
        got = get_theta(g["z_ties"], int(k))
        assert (np.isinf(ref) and np.isinf(got)) or got == pytest.approx(ref, rel=1e-12)


def test_shape_errors():
    with pytest.raises(ValueError):
        ops.square_to_vec(np.zeros((3, 4)))
    with pytest.raises(ValueError):
        ops.laplacian_squareform_vec(np.zeros((2, 7)))
    with pytest.raises(ValueError):
        get_theta(np.zeros((3, 4)), 1)
    with pytest.raises(ValueError):  # a block with fewer than avg_degree + 1 columns
        get_theta(np.ones((4, 4)), 3, blocks=np.array([0, 0, 1, 1]))

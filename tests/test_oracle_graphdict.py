"""Pin the GraphDictLog oracle against golden vectors produced by the real reference
(tests/golden/make_golden.py:gen_graphdict).  CPU only.

BLAS-level bit equality is what we observe in the build container (same NumPy/OpenBLAS as the
generator); elsewhere summation order inside ``@`` may differ, so the assertions use 1e-9.
"""

import warnings

import numpy as np
import pytest

from oracle.graphdict import GraphDictLogOracle

TOL = dict(rtol=1e-9, atol=1e-11)

CASE_B = dict(n_atoms=3, window_size=4, l1_w=0.02, ortho_w=0.1, l1_c=0.01, log_c=0.05, l1_diff_c=0.001,
              step_c=0.5, step_w=0.3, step_dual=0.7, tol=1e-12, max_iter=60, random_state=5)
CASE_C = dict(n_atoms=2, random_state=11, l1_w=0.05, step=0.5, max_iter=1000)
INITS = {
    "uniform_prior": dict(init_strategy="uniform", weight_prior=0.5, coefficient_prior=0.8),
    "discrete": dict(init_strategy="discrete", weight_prior=0.3),
    "exp": dict(init_strategy="exp", weight_prior=0.2, coefficient_prior=0.4),
    "exact": dict(init_strategy="exact", weight_prior=0.25, coefficient_prior=0.6),
    "kmd_corr_a": dict(init_strategy_c="k-means++d", init_strategy_w="correlation_a"),
    "kmc_corr_a_pinv": dict(init_strategy_c="k-means++c", init_strategy_w="correlation_a_pinv"),
    "window_uniform": dict(init_strategy="uniform", window_size=3),
}


@pytest.fixture(autouse=True)
def quiet():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        yield


def check_state(m, g, tag):
    np.testing.assert_allclose(m.coefficients_, g[tag + "_c"], **TOL)
    np.testing.assert_allclose(m.weights_, g[tag + "_w"], **TOL)
    np.testing.assert_allclose(m.dual_, g[tag + "_dual"], **TOL)


@pytest.mark.parametrize("k", [1, 3])
@pytest.mark.parametrize("l1", [0.0, 0.01])
def test_reference_test_shapes(golden, k, l1):
    g = golden("graphdict")
    tag = f"A_k{k}_l1w{l1}"
    m = GraphDictLogOracle(n_atoms=k, l1_w=l1, random_state=240920, max_iter=100).fit(g["A_x"])
    check_state(m, g, tag)
    np.testing.assert_allclose(m.history_, g[tag + "_history"], **TOL)
    assert m.converged_ == int(g[tag + "_converged"])


def test_windows_and_all_regularisers_stepwise(golden):
    g = golden("graphdict")
    for steps in (1, 2, 3, 20):
        m = GraphDictLogOracle(**{**CASE_B, "max_iter": steps}).fit(g["B_x"])
        t = g["B_x"].shape[0]
        np.testing.assert_allclose(m.coefficients_, g[f"B_c_it{steps}"][:, :t], **TOL)
        np.testing.assert_allclose(m.weights_, g[f"B_w_it{steps}"], **TOL)
        np.testing.assert_allclose(m.dual_, g[f"B_dual_it{steps}"], **TOL)
    m = GraphDictLogOracle(**CASE_B).fit(g["B_x"])
    check_state(m, g, "B")
    np.testing.assert_allclose(m.history_, g["B_history"], **TOL)


def test_runs_to_tolerance_predict_score(golden):
    g = golden("graphdict")
    m = GraphDictLogOracle(**CASE_C).fit(g["C_x"])
    assert m.converged_ == int(g["C_converged"]) == 466
    check_state(m, g, "C")
    assert m.score(g["C_x"]) == pytest.approx(float(g["C_score"]), rel=1e-9)
    m.random_state = np.random.RandomState(3)
    m.max_iter = 50
    np.testing.assert_allclose(m.predict(g["C_predict_x"]), g["C_predict"], **TOL)


@pytest.mark.parametrize("tag", sorted(INITS))
def test_initialisation_strategies(golden, tag):
    g = golden("graphdict")
    m = GraphDictLogOracle(n_atoms=3, random_state=42, **INITS[tag])
    m.initialize(g["D_x"])
    np.testing.assert_allclose(m.coefficients_, g[f"D_{tag}_c"], **TOL)
    np.testing.assert_allclose(m.weights_, g[f"D_{tag}_w"], **TOL)


def test_bad_strategies():
    x = np.zeros((4, 3))
    with pytest.raises(ValueError):
        GraphDictLogOracle(init_strategy_c="nope").initialize(x)
    with pytest.raises(ValueError):
        GraphDictLogOracle(init_strategy_w="correlation_x").initialize(x)

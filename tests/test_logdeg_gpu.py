"""Parity of the HIP log-degree path (through the C ABI) against the oracle and
the golden vectors of the reference.  All tests here need an MI355X.

Tolerances (fp64): the per-edge arithmetic follows NumPy operation by operation
(the tile engine re-associates two of them: gamma folded into the node values,
two fused multiply-adds) and only the order of the degree
sums / norm sums differs, so fixed-iteration iterates agree to ~1e-15 .. 1e-14.
The contract checked is SURVEY.md section 7 "Parity definition":
  (i)   fixed-iteration, un-thresholded iterates: atol 1e-10
  (ii)  thresholded output: equal except where the oracle's raw value is within
        atol of edge_tol
  (iii) converge mode stops within +-1 iteration of the oracle.
"""

import os

import numpy as np
import pytest
from scipy.spatial.distance import squareform

from graph_learning_amd import LogModel, gsp_learn_graph_log_degrees
from graph_learning_amd import _lib
from graph_learning_amd.synth import logmodel_problem
from oracle import logdeg as old

pytestmark = pytest.mark.gpu

ATOL = 1e-10
CASES = ["logmodel_n7_d23", "logmodel_n16_d40", "logmodel_n64_d256"]


# The engines the library chooses by itself run the whole file in every default session; round 1's grid-barrier engine
# (gl_resident.hip, superseded by the tile engine everywhere: DESIGN.md section 3a') runs it only with
# GRAPHLEARN_AMD_TEST_ALL=1 and keeps one smoke test by default (test_round1_resident_engine_smoke), so that the
# suite's wall time stays inside the driver's step limit.
ALL_ENGINES = os.environ.get("GRAPHLEARN_AMD_TEST_ALL", "0") not in ("", "0")
ENGINES = ["stream", "resident", "small"] + (["resident1"] if ALL_ENGINES else [])


@pytest.fixture(autouse=True, params=ENGINES)
def engine(request, monkeypatch):
    """Every test of this file runs on every iteration engine: streaming (one edge-pass + one node-pass launch
    per iteration), the resident family forced -- "resident": the tile engine (gl_tile.hip: iterate in registers,
    z in LDS, tagged hand-offs) wherever its work decomposition applies, else the first resident engine;
    "resident1" (opt-in, see above): the first resident engine (gl_resident.hip: grid barriers) -- and small
    (n <= 128: one workgroup, one launch per call; where it does not apply the automatic choice takes over, so the
    tests of larger lattices skip it: they would repeat the "resident" run)."""
    monkeypatch.setenv("GRAPHLEARN_AMD_ENGINE", request.param)
    return request.param


def skip_small_above_128(engine):
    if engine == "small":
        pytest.skip("n > 128: the small engine does not apply and the automatic choice repeats the tile engine's run")


_C2 = {}


def c2_reference():
    """The n = 4096, d = 1024 problem of BASELINE configs[1] and the oracle's iterates after 5 and 40 iterations --
    computed ONCE per session (the oracle needs ~25 s for the operators, distances and 40 iterations)."""
    if not _C2:
        x, _ = logmodel_problem(4096, 1024, seed=1)
        z_ref = old.signal_distances(x)
        theta_ref = old.theta_for_degree(squareform(z_ref), 10)
        solver = old.FBFSolver(z_ref * theta_ref, 1, 1)
        snaps = {}
        for k in range(1, 41):
            rel = solver.step()
            if k in (5, 40):
                snaps[k] = (solver.w.copy(), solver.d.copy(), rel)
        _C2.update(x=x, z=z_ref, theta=theta_ref, zt=z_ref * theta_ref, snaps=snaps)
    return _C2


def raw_iterate(z, alpha, beta, k, step_size=0.5, edge_init=None):
    n = old.n_nodes_from_edges(z.size)
    with _lib.LogDegreeSession(n) as s:
        s.set_distances(z)
        s.init(alpha, beta, step_size, edge_init)
        n_it, conv, rel = s.iterate(k, 0.0)
        assert n_it == k and not conv
        return s.weights_condensed(), s.dual(), rel


@pytest.mark.parametrize("case", CASES)
def test_fixed_iterations_match_reference_golden(golden, case):
    g = golden(case)
    zt = g["z"] * float(g[f"theta_deg{int(g['deg'])}"])
    for key in g.files:
        if key.startswith("raw_w_it"):
            k = int(key[len("raw_w_it"):])
            w, _, _ = raw_iterate(zt, 1, 1, k)
            np.testing.assert_allclose(w, g[key], rtol=0, atol=ATOL, err_msg=key)
    w, _, _ = raw_iterate(zt, 2.0, 0.5, 50, step_size=0.3)
    np.testing.assert_allclose(w, g["raw_w_a2_b05_s03_it50"], rtol=0, atol=ATOL)
    w, _, _ = raw_iterate(zt, 1, 1, 30, edge_init=g["edge_init"])
    np.testing.assert_allclose(w, g["raw_w_warm_it30"], rtol=0, atol=ATOL)


@pytest.mark.parametrize("case", CASES)
def test_distances_and_theta_from_signals(golden, case):
    g = golden(case)
    n = g["x"].shape[1]
    for key in g.files:
        if not key.startswith("theta_deg"):
            continue
        k = None if key.endswith("None") else int(key[len("theta_deg"):])
        with _lib.LogDegreeSession(n) as s:
            theta = s.set_signals(g["x"], k)
            z = s.get_distances()
        # Gram-form distances vs the reference's direct differences
        np.testing.assert_allclose(z, g["z"], rtol=1e-11, atol=1e-12 * g["z"].max())
        assert theta == pytest.approx(float(g[key]), rel=1e-10)


@pytest.mark.parametrize("case", CASES)
def test_logmodel_fit_defaults(golden, case):
    g = golden(case)
    m = LogModel(avg_degree=int(g["deg"])).fit(g["x"])
    ref = g["weights_default"]
    assert m.weights_.shape == ref.shape and m.weights_.dtype == np.float64
    # SURVEY section 7 allows +-1 iteration; every engine has been observed to stop exactly where the reference does
    assert m.n_iter_ == int(g["n_iter_default"])
    assert m.theta_ == pytest.approx(float(g[f"theta_deg{int(g['deg'])}"]), rel=1e-10)
    # (ii): entries may only differ where the reference value sits at the threshold
    o = old.LogModelOracle(avg_degree=int(g["deg"])).fit(g["x"])
    raw = squareform(o.info_["raw"])
    near = np.abs(raw - 1e-3) < 1e-7
    np.testing.assert_allclose(m.weights_[~near], ref[~near], rtol=0, atol=1e-7)
    assert np.array_equal(m.weights_, m.weights_.T) and not m.weights_.diagonal().any()


@pytest.mark.parametrize("case", CASES)
def test_converge_mode_stops_like_reference(golden, case):
    g = golden(case)
    zt = g["z"] * float(g[f"theta_deg{int(g['deg'])}"])
    w, info = gsp_learn_graph_log_degrees(zt, 1, 1, maxit=20000, tol=1e-5, edge_tol=-np.inf, return_info=True)
    assert info["converged"]
    assert info["n_iter"] == int(g["n_iter_converged"])   # (contract: +-1; observed offset: 0 on every engine)
    np.testing.assert_allclose(squareform(w, checks=False), g["raw_w_converged"], rtol=0, atol=1e-6)
    if info["n_iter"] == int(g["n_iter_converged"]):
        np.testing.assert_allclose(squareform(w, checks=False), g["raw_w_converged"], rtol=0, atol=ATOL)


def test_dual_variable_and_rel_changes_match_oracle(golden):
    g = golden("logmodel_n64_d256")
    zt = g["z"] * float(g["theta_deg6"])
    solver = old.FBFSolver(zt, 1, 1)
    for _ in range(37):
        rp, rd = solver.step()
    w, d, rel = raw_iterate(zt, 1, 1, 37)
    np.testing.assert_allclose(w, solver.w, rtol=0, atol=ATOL)
    np.testing.assert_allclose(d, solver.d, rtol=0, atol=ATOL)
    assert rel[0] == pytest.approx(rp, rel=1e-9) and rel[1] == pytest.approx(rd, rel=1e-9)


def test_resume_equals_single_run(golden):
    g = golden("logmodel_n16_d40")
    zt = g["z"] * float(g["theta_deg6"])
    with _lib.LogDegreeSession(16) as s:
        s.set_distances(zt)
        s.init(1, 1, 0.5, None)
        s.iterate(13, 0.0)
        s.iterate(7, 0.0)
        s.iterate_async(30)
        assert s.total_iterations == 50
        w = s.weights_condensed()
    # with the resident engine forced, short calls (< 4 iterations; round 1's engine: < 16) still run on the streaming
    # engine: same iterates up to the order of the degree sums
    np.testing.assert_allclose(w, raw_iterate(zt, 1, 1, 50)[0], rtol=0, atol=1e-13)
    np.testing.assert_allclose(w, g["raw_w_it50"], rtol=0, atol=ATOL)
    # each engine by itself is bit-reproducible run to run
    np.testing.assert_array_equal(raw_iterate(zt, 1, 1, 50)[0], raw_iterate(zt, 1, 1, 50)[0])
    np.testing.assert_array_equal(raw_iterate(zt, 1, 1, 12)[0], raw_iterate(zt, 1, 1, 12)[0])


@pytest.mark.parametrize("n", [2, 3, 5, 63, 64, 65, 127, 128, 129, 200])
def test_ragged_sizes_against_oracle(n):
    """Tile edges: n below / at / just above multiples of the 64-wide tile."""
    rng = np.random.default_rng(n)
    e = n * (n - 1) // 2
    z = rng.uniform(0.05, 3.0, size=e)
    ei = rng.uniform(0, 0.3, size=e) * (rng.uniform(size=e) < 0.5)
    ops = old.degree_operator(n)
    for init in (None, ei):
        _, info = old.learn_log_degrees(z, 1.3, 0.7, edge_init=None if init is None else init.copy(), maxit=25,
                                        tol=0.0, edge_tol=-np.inf, return_info=True, ops=ops)
        w, d, _ = raw_iterate(z, 1.3, 0.7, 25, edge_init=init)
        np.testing.assert_allclose(w, info["raw"], rtol=0, atol=ATOL)
        np.testing.assert_allclose(d, info["dual"], rtol=0, atol=ATOL)


def test_maxit_zero_returns_thresholded_init():
    n, e = 9, 36
    ei = np.linspace(0, 0.01, e)
    w = gsp_learn_graph_log_degrees(np.ones(e), 1, 1, edge_init=ei.copy(), maxit=0)
    exp = ei.copy()
    exp[exp < 1e-3] = 0
    np.testing.assert_array_equal(w, squareform(exp))


def test_errors():
    with pytest.raises(ValueError, match="vector or matrix"):
        gsp_learn_graph_log_degrees(np.zeros((2, 2, 3)), 1, 1)
    with pytest.raises(ValueError):
        gsp_learn_graph_log_degrees(np.ones(10), 1, 1, edge_init=np.ones(9))
    with pytest.raises(ValueError):
        LogModel(avg_degree=2).fit(np.ones(5))
    with pytest.raises(ValueError):
        LogModel(avg_degree=7).fit(np.random.default_rng(0).standard_normal((10, 7)))


def test_sklearn_surface():
    m = LogModel(avg_degree=3, maxit=10)
    assert m.get_params() == {"avg_degree": 3, "edge_init": None, "maxit": 10, "tol": 1e-5, "step_size": 0.5,
                              "edge_tol": 1e-3}
    m.set_params(tol=1e-3)
    assert m.tol == 1e-3
    x = np.random.default_rng(240920).standard_normal((23, 7))   # the reference's tests/test_models.py fixture
    m.fit(x)
    assert m.weights_.shape == (7, 7) and isinstance(m.theta_, float)
    assert LogModel().fit(x).theta_ == 1


# ------------------------------------------------------------------ BASELINE sizes
def test_c2_size_five_iterations_against_oracle(engine):
    """n=4096, d=1024 (BASELINE config 2): distances, theta and 5 iterations vs the oracle."""
    skip_small_above_128(engine)
    ref = c2_reference()
    z_ref, theta_ref = ref["z"], ref["theta"]
    with _lib.LogDegreeSession(4096) as s:
        theta = s.set_signals(ref["x"], 10)
        z = s.get_distances()
        np.testing.assert_allclose(z, z_ref, rtol=1e-10, atol=1e-12 * z_ref.max())
        assert theta == pytest.approx(theta_ref, rel=1e-10)
        # iterate on the oracle's own z*theta so that both sides see identical inputs
        s.set_distances(ref["zt"])
        s.init(1, 1, 0.5, None)
        s.iterate(5, 0.0)
        w = s.weights_condensed()
        d = s.dual()
    w5, d5, _ = ref["snaps"][5]
    np.testing.assert_allclose(w, w5, rtol=0, atol=ATOL)
    np.testing.assert_allclose(d, d5, rtol=0, atol=ATOL)


def test_c2_size_persistent_engine_against_oracle(engine):
    """The kernel bench.py times -- the persistent engine at n=4096, d=1024, all iterations of a call in ONE
    launch -- against the oracle (smooth_learning.py:146-167): 17 + 23 iterations in two launches (the
    second starts from an odd iteration count), w and the dual variable to 1e-10."""
    if engine == "stream":
        pytest.skip("the streaming engine at this size is covered by the five-iteration test")
    skip_small_above_128(engine)
    ref = c2_reference()
    with _lib.LogDegreeSession(4096) as s:
        s.set_distances(ref["zt"])
        s.init(1, 1, 0.5, None)
        assert s.engine == "resident"          # forced
        assert s.engine_kernel == ("k_resident" if engine == "resident1" else "k_tile")
        n_it, conv, rel17 = s.iterate(17, 0.0)
        assert n_it == 17 and not conv and s.engine == "resident"
        n_it, conv, rel = s.iterate(23, 0.0)
        assert n_it == 23 and not conv and s.total_iterations == 40
        w = s.weights_condensed()
        d = s.dual()
    w40, d40, rel40 = ref["snaps"][40]
    np.testing.assert_allclose(w, w40, rtol=0, atol=ATOL)
    np.testing.assert_allclose(d, d40, rtol=0, atol=ATOL)
    assert rel[0] == pytest.approx(rel40[0], rel=1e-9)
    assert rel[1] == pytest.approx(rel40[1], rel=1e-9)


def test_c2_size_properties_after_many_iterations(engine):
    """Size-independent properties at n=4096 after 300 iterations: determinism, dual = prox
    fixed-point structure (dvar < 0), symmetry of the square output, degree identity."""
    skip_small_above_128(engine)
    x, _ = logmodel_problem(4096, 256, seed=3)
    outs = []
    for _ in range(2):
        with _lib.LogDegreeSession(4096) as s:
            s.set_signals(x, 10)
            s.init(1, 1, 0.5, None)
            s.iterate(300, 0.0)
            outs.append((s.weights_condensed(), s.dual(), s.weights_square(1e-3)))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    w, d, sq = outs[0]
    assert np.isfinite(w).all() and np.isfinite(d).all()
    assert (d < 0).all()                       # dual of -log: strictly negative
    assert np.array_equal(sq, sq.T) and not sq.diagonal().any()
    wt = w.copy()
    wt[wt < 1e-3] = 0
    i, j = np.triu_indices(4096, 1)
    assert np.array_equal(sq[i, j], wt)
    raw_deg = squareform(w).sum(1)
    assert (raw_deg > 0).all()                 # log barrier: every node keeps positive (raw) degree


def test_resident_engine_falls_back_when_workgroups_do_not_meet(golden, engine, monkeypatch):
    """The resident engine's workgroups must all be on the chip at once.  When they are not (shared
    GPU), the start-up rendezvous times out before anything is modified and the call is served by
    the streaming engine -- synchronously and for an asynchronous call at its next sync point."""
    if engine not in ("resident", "resident1"):
        pytest.skip("resident family only")
    g = golden("logmodel_n64_d256")
    zt = g["z"] * float(g["theta_deg6"])
    monkeypatch.setenv("GRAPHLEARN_AMD_RES_TEST_NOT_RESIDENT", "1")  # rendezvous expects one workgroup too many
    with _lib.LogDegreeSession(64) as s:
        s.set_distances(zt)
        s.init(1, 1, 0.5, None)
        assert s.engine == "resident"
        n_it, conv, _ = s.iterate(20, 0.0)
        assert n_it == 20 and not conv and s.engine == "stream"
        s.iterate_async(30)
        w = s.weights_condensed()
        assert s.total_iterations == 50
    np.testing.assert_allclose(w, g["raw_w_it50"], rtol=0, atol=ATOL)
    with _lib.LogDegreeSession(64) as s:
        s.set_distances(zt)
        s.init(1, 1, 0.5, None)
        s.iterate_async(50)            # falls back at the next synchronising call
        w = s.weights_condensed()
        assert s.engine == "stream" and s.total_iterations == 50
    np.testing.assert_allclose(w, g["raw_w_it50"], rtol=0, atol=ATOL)


def test_rounds_fall_back_untouched_when_workgroups_do_not_meet(engine, monkeypatch):
    """Past n = 4096 the tile engine updates its streamed tiles in place -- but only after the start-up rendezvous:
    a launch whose workgroups never met has written nothing, and the call (also an asynchronous one, at its next
    synchronising call) is served by the streaming engine from the untouched state."""
    if engine != "resident":
        pytest.skip("the tile engine's rounds only")
    n = 4203
    z = np.random.default_rng(n).uniform(0.05, 3.0, size=n * (n - 1) // 2)
    with _lib.LogDegreeSession(n) as s:       # reference run: streaming engine throughout
        monkeypatch.setenv("GRAPHLEARN_AMD_ENGINE", "stream")
        s.set_distances(z)
        s.init(1.0, 1.0, 0.5, None)
        s.iterate(17, 0.0)
        s.iterate(21, 0.0)
        w_ref, d_ref = s.weights_condensed(), s.dual()
    monkeypatch.setenv("GRAPHLEARN_AMD_ENGINE", "resident")
    monkeypatch.setenv("GRAPHLEARN_AMD_RES_TEST_NOT_RESIDENT", "1")
    with _lib.LogDegreeSession(n) as s:
        s.set_distances(z)
        s.init(1.0, 1.0, 0.5, None)
        assert s.engine_kernel == "k_tile"
        s.iterate_async(17)                   # fails at the rendezvous; noticed at the next synchronising call
        n_it, conv, _ = s.iterate(21, 0.0)
        assert n_it == 21 and s.engine == "stream" and s.total_iterations == 38
        w, d = s.weights_condensed(), s.dual()
    # the same engine ran the same iterations from the same state (its degree sums are grouped by call length: 1e-16)
    np.testing.assert_allclose(w, w_ref, rtol=0, atol=1e-13)
    np.testing.assert_allclose(d, d_ref, rtol=0, atol=1e-13)


@pytest.mark.parametrize("n", [2600, 3000])
def test_resident_matches_streaming_at_ragged_large_sizes(n, engine, monkeypatch):
    """Resident engine on lattices with a ragged last block (n not a multiple of 64) and many fringe
    tiles: 60 iterations agree with the streaming engine up to summation order, and the dual too."""
    if engine not in ("resident", "resident1"):
        pytest.skip("compares each persistent engine with the streaming one")
    rng = np.random.default_rng(n)
    e = n * (n - 1) // 2
    z = rng.uniform(0.05, 3.0, size=e)
    out = {}
    for eng in ("stream", engine):
        monkeypatch.setenv("GRAPHLEARN_AMD_ENGINE", eng)
        with _lib.LogDegreeSession(n) as s:
            s.set_distances(z)
            s.init(1.2, 0.8, 0.5, None)
            assert s.engine == ("stream" if eng == "stream" else "resident")
            if eng == "resident":
                assert s.engine_kernel == "k_tile"
            n_it, conv, rel = s.iterate(60, 0.0)
            out[eng] = (s.weights_condensed(), s.dual(), rel)
    np.testing.assert_allclose(out[engine][0], out["stream"][0], rtol=0, atol=1e-13)
    np.testing.assert_allclose(out[engine][1], out["stream"][1], rtol=0, atol=1e-12)
    assert out[engine][2][0] == pytest.approx(out["stream"][2][0], rel=1e-9)
    assert out[engine][2][1] == pytest.approx(out["stream"][2][1], rel=1e-9)


@pytest.mark.parametrize("n", [12, 128, 150])
def test_deferred_theta_is_the_same_theta(n):
    """set_signals(..., deferred=True) leaves theta on the device (n <= 128: also its computation) and
    theta() fetches it later: same value, same scaled distances as the immediate form."""
    x = np.random.default_rng(n).standard_normal((3 * n, n))
    with _lib.LogDegreeSession(n) as s:
        t1 = s.set_signals(x, 5)
        s.init(1, 1, 0.5, None)
        s.iterate(20, 0.0)
        w1 = s.weights_condensed(-np.inf)
        assert s.set_signals(x, 5, deferred=True) is None
        s.init(1, 1, 0.5, None)
        s.iterate(20, 0.0)
        w2 = s.weights_condensed(-np.inf)
        assert s.theta() == t1
        assert np.array_equal(w1, w2)


def test_near_coincident_columns_keep_relative_accuracy():
    """Columns that almost coincide: the Gram form G_ii + G_jj - 2 G_ij cancels there, the reference takes
    direct differences (pdist, smooth_learning.py:278).  The device path recomputes such entries directly:
    the smallest distances -- the ones theta is built from -- keep full relative accuracy."""
    rng = np.random.default_rng(77)
    n, d = 96, 300
    x = rng.standard_normal((d, n)) * 3 + 5.0
    for (a, b, eps) in ((1, 0, 1e-9), (40, 7, 1e-6), (95, 94, 1e-3), (60, 61, 0.0)):
        x[:, a] = x[:, b] + eps * rng.standard_normal(d)
    z_ref = old.signal_distances(x)
    with _lib.LogDegreeSession(n) as s:
        theta = s.set_signals(x, 4)
        z = s.get_distances()
    sq, sq_ref = squareform(z), squareform(z_ref)
    assert sq[60, 61] == 0.0 == sq_ref[60, 61]
    for (a, b) in ((1, 0), (40, 7), (95, 94)):
        assert sq[a, b] == pytest.approx(sq_ref[a, b], rel=1e-12), (a, b)
    mask = z_ref > 0
    assert np.max(np.abs(z[mask] - z_ref[mask]) / z_ref[mask]) < 1e-11
    assert theta == pytest.approx(old.theta_for_degree(sq_ref, 4), rel=1e-10)


def test_set_distances_after_deferred_theta_resets_theta():
    """A theta left on the device by set_signals(deferred) belongs to those distances: after set_distances the
    session's theta is 1 again."""
    x = np.random.default_rng(4).standard_normal((60, 20))
    with _lib.LogDegreeSession(20) as s:
        assert s.set_signals(x, 3, deferred=True) is None
        s.set_distances(np.ones(190))
        assert s.theta() == 1.0


@pytest.mark.parametrize("n", [4160, 4203, 4500])
def test_just_past_the_register_files_against_oracle(engine, n):
    """n = 4160 (65 node blocks, 2080 bulk tiles), the ragged n = 4203 and n = 4500: one to six blocks past what the register
    files hold.  The tile engine keeps one tile per wave in registers and streams the rest -- here as "split extras": the
    tiles the chip has no wave for (diagonal tiles included) are streamed by all eight waves of a workgroup together, four
    row pairs each, one or two per workgroup, under a validity mask; the streaming engine is forced once.  18 iterations
    in one call against the oracle."""
    if engine not in ("stream", "resident"):
        pytest.skip("once forced to the streaming engine, once on the tile engine")
    if n == 4500 and engine == "stream":
        pytest.skip("the streaming engine is covered at the two other sizes")
    rng = np.random.default_rng(n)
    e = n * (n - 1) // 2
    z = rng.uniform(0.05, 3.0, size=e)
    with _lib.LogDegreeSession(n) as s:
        s.set_distances(z)
        s.init(1.1, 0.9, 0.5, None)
        if engine == "stream":
            assert s.engine == "stream" and s.engine_kernel == "k_edge"
        else:
            assert s.engine == "resident" and s.engine_kernel == "k_tile"
        n_it, conv, rel = s.iterate(18, 0.0)
        w, d = s.weights_condensed(), s.dual()
    _, info = old.learn_log_degrees(z, 1.1, 0.9, maxit=18, tol=0.0, edge_tol=-np.inf, return_info=True)
    np.testing.assert_allclose(w, info["raw"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(d, info["dual"], rtol=0, atol=ATOL)
    assert rel[0] == pytest.approx(info["rel"][0], rel=1e-9) and rel[1] == pytest.approx(info["rel"][1], rel=1e-9)


@pytest.mark.parametrize("n", [5000, 9300])
def test_rounds_of_the_tile_engine_in_converge_mode_against_the_streaming_engine(engine, monkeypatch, n):
    """n = 5000 (ragged, 3 rounds) and n = 9300 (ragged, 7 rounds, 15 node blocks per workgroup): run to tol on the tile
    engine and on the streaming engine (itself checked against the oracle above): same stopping iteration, same ratios,
    same iterate; and a second call continues the first."""
    if engine != "resident":
        pytest.skip("compares the two engines itself")
    z = np.random.default_rng(n).uniform(0.05, 3.0, size=n * (n - 1) // 2)
    out = {}
    for eng in ("stream", "resident"):
        monkeypatch.setenv("GRAPHLEARN_AMD_ENGINE", eng)
        with _lib.LogDegreeSession(n) as s:
            s.set_distances(z)
            s.init(1.0, 1.0, 0.5, None)
            assert s.engine_kernel == ("k_edge" if eng == "stream" else "k_tile")
            first = s.iterate(40, 0.0)
            rest = s.iterate(100000, 1e-3)
            out[eng] = (first, rest, s.weights_condensed(), s.dual())
    a, b = out["stream"], out["resident"]
    assert a[0][0] == b[0][0] == 40 and a[1][0] == b[1][0] and a[1][1] and b[1][1]
    assert b[1][2][0] == pytest.approx(a[1][2][0], rel=1e-9) and b[1][2][1] == pytest.approx(a[1][2][1], rel=1e-9)
    np.testing.assert_allclose(b[2], a[2], rtol=0, atol=1e-12)
    np.testing.assert_allclose(b[3], a[3], rtol=0, atol=1e-12)


def test_large_lattice_n8192_against_dense_restatement(engine):
    """n = 8192 (beyond the Infinity Cache: 3 x 268 MB): the streaming engine (3 iterations) and the tile engine (5 rounds
    per iteration; 8 iterations) against a dense NumPy restatement of :146-167 (S w = row sums of
    the symmetric matrix, S'd = d_i + d_j), which needs no sparse operator at this size."""
    if engine not in ("stream", "resident"):
        pytest.skip("once on the streaming engine, once on the tile engine")
    iters = 3 if engine == "stream" else 8
    n = 8192
    rng = np.random.default_rng(5)
    iu = np.triu_indices(n, 1)
    zsq = np.zeros((n, n))
    zsq[iu] = rng.uniform(0.05, 3.0, size=iu[0].size)
    zsq = zsq + zsq.T
    alpha, beta, step = 1.0, 1.0, 0.5
    g = step / (2 * beta + np.sqrt(2 * (n - 1)))
    wm = np.zeros((n, n))
    d = np.zeros(n)
    off = ~np.eye(n, dtype=bool)
    for _ in range(iters):
        y = (1 - 2 * g * beta) * wm - g * (d[:, None] + d[None, :])
        yb = d + g * wm.sum(1)
        p = np.where(y - 2 * g * zsq > 0, y - 2 * g * zsq, 0) * off
        pb = (yb - np.sqrt(yb**2 + 4 * alpha * g)) / 2
        q = (1 - 2 * g * beta) * p - g * (pb[:, None] + pb[None, :])
        qb = pb + g * p.sum(1)
        wm = (wm - y + q) * off
        d = d - yb + qb
    with _lib.LogDegreeSession(n) as s:
        s.set_distances(np.ascontiguousarray(zsq[iu]))
        s.init(alpha, beta, step, None)
        assert s.engine_kernel == ("k_edge" if engine == "stream" else "k_tile")
        s.iterate(iters, 0.0)
        w, dd = s.weights_condensed(), s.dual()
    np.testing.assert_allclose(w, wm[iu], rtol=0, atol=1e-12)
    np.testing.assert_allclose(dd, d, rtol=0, atol=1e-11)


# ------------------------------------------------------------------ the instantiations bench.py times
def test_c2_size_timed_instantiation_against_oracle(engine):
    """bench.py times gl_logdeg_iterate_async -> k_tile<32, true, false, 0> (the variant WITHOUT the norms of the
    stopping rule) at n=4096, d=1024; the driver runs it as `--steps 20`.  Exactly that: two asynchronous calls of 20
    iterations against 40 iterations of the oracle (smooth_learning.py:146-167), w and the dual variable to 1e-10."""
    if engine != "resident":
        pytest.skip("the tile engine's asynchronous variant")
    ref = c2_reference()
    with _lib.LogDegreeSession(4096) as s:
        s.set_distances(ref["zt"])
        s.init(1, 1, 0.5, None)
        assert s.engine == "resident" and s.engine_kernel == "k_tile"
        s.iterate_async(20)
        s.iterate_async(20)
        assert s.total_iterations == 40
        w = s.weights_condensed()
        d = s.dual()
        assert s.engine == "resident"          # no fall-back happened: the tile engine produced these
    w40, d40, _ = ref["snaps"][40]
    np.testing.assert_allclose(w, w40, rtol=0, atol=ATOL)
    np.testing.assert_allclose(d, d40, rtol=0, atol=ATOL)


@pytest.mark.parametrize("n,iters", [(4096, 2000), (8192, 48), (3000, 301), (2048, 500), (1030, 333)])
def test_async_variant_equals_synchronous_variant_bit_for_bit(engine, n, iters):
    """The asynchronous calls run the kernel variant without the stopping rule's norms, the synchronous ones the variant
    with them: the iterate's arithmetic is the same, so from the same start the results are EQUAL (not close).  Sizes:
    the BASELINE lattice (U = 32, everything in registers), n = 8192 (rounds, non-temporal streams:
    k_tile<32,false,false,2>), and one lattice per shorter chunk length (ragged / U = 16 / U = 8)."""
    if engine != "resident":
        pytest.skip("tile engine only")
    z = np.random.default_rng(n).uniform(0.05, 3.0, size=n * (n - 1) // 2)
    out = []
    for mode in ("async", "sync"):
        with _lib.LogDegreeSession(n) as s:
            s.set_distances(z)
            s.init(1.0, 1.0, 0.5, None)
            assert s.engine_kernel == "k_tile"
            if mode == "async":
                s.iterate_async(iters)
            else:
                n_it, conv, _ = s.iterate(iters, 0.0)
                assert n_it == iters and not conv
            out.append((s.weights_condensed(), s.dual()))
            assert s.engine == "resident" and s.total_iterations == iters
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])


def test_tile_engine_falls_back_after_the_session_has_advanced_on_the_streaming_engine(golden, engine, monkeypatch):
    """a synchronous tile-engine call whose workgroups never meet must fall back to the streaming engine also
    when the session has already advanced (a short call, < 4 iterations, runs on the streaming engine and leaves the
    counters where a stale comparison would mistake the failure for a mid-run one)."""
    if engine != "resident":
        pytest.skip("tile engine only")
    g = golden("logmodel_n64_d256")
    zt = g["z"] * float(g["theta_deg6"])
    monkeypatch.setenv("GRAPHLEARN_AMD_RES_TEST_NOT_RESIDENT", "1")
    with _lib.LogDegreeSession(64) as s:
        s.set_distances(zt)
        s.init(1, 1, 0.5, None)
        s.iterate(3, 0.0)                      # streaming engine (short call)
        s.iterate(3, 0.0)
        s.iterate(3, 0.0)
        s.iterate(1, 0.0)
        assert s.engine == "resident" and s.total_iterations == 10
        n_it, conv, _ = s.iterate(40, 0.0)     # tile engine: rendezvous fails, nothing touched, streaming engine redoes it
        assert n_it == 40 and not conv and s.engine == "stream" and s.total_iterations == 50
        w = s.weights_condensed()
    np.testing.assert_allclose(w, g["raw_w_it50"], rtol=0, atol=ATOL)


@pytest.mark.parametrize("n", [100, 200, 700])
def test_rule_that_fires_at_the_head_of_a_call_fires_at_the_same_iteration_on_every_engine(engine, monkeypatch, n):
    """the tile engine looks at the stopping rule with a lag of one iteration while the ratios are far from
    tol -- but never during the first iterations of a call, which it judges exactly.  With tol set just above the ratios of
    iteration 1 (iteration 0 is never judged, smooth_learning.py:156-161; its ratios are inf, so this is the steepest drop
    there is) every engine stops after exactly two iterations, as the reference's loop does."""
    z = np.random.default_rng(n).uniform(0.05, 3.0, size=n * (n - 1) // 2)
    monkeypatch.setenv("GRAPHLEARN_AMD_ENGINE", "stream")
    with _lib.LogDegreeSession(n) as s:
        s.set_distances(z)
        s.init(1.0, 1.0, 0.5, None)
        _, _, rel = s.iterate(2, 0.0)          # ratios of iteration 1
    tol = 1.001 * max(rel)
    res = {}
    for eng in ("stream", engine):
        monkeypatch.setenv("GRAPHLEARN_AMD_ENGINE", eng)
        with _lib.LogDegreeSession(n) as s:
            s.set_distances(z)
            s.init(1.0, 1.0, 0.5, None)
            res[eng] = s.iterate(1000, tol)[:2] + (s.weights_condensed(-np.inf),)
    assert res["stream"][:2] == (2, True)
    assert res[engine][:2] == (2, True)
    np.testing.assert_allclose(res[engine][2], res["stream"][2], rtol=0, atol=1e-12)


# ------------------------------------------------------------------ round 4: converge mode at mid sizes, against the ORACLE
_OPS = {}


def oracle_trajectory(zt, iters, edge_init=None, snaps=()):
    """The oracle's ratios after every iteration (smooth_learning.py:156-161) and its state after the iterations in `snaps`
    (number of iterations done)."""
    n = old.n_nodes_from_edges(zt.size)
    if n not in _OPS:
        _OPS.clear()                  # (one size at a time: the operators of n = 3500 take 200 MB)
        _OPS[n] = old.degree_operator(n)
    solver = old.FBFSolver(zt, 1, 1, edge_init=None if edge_init is None else edge_init.copy(), ops=_OPS[n])
    rel, states = [], {}
    for k in range(1, iters + 1):
        rel.append(solver.step())
        if k in snaps:
            states[k] = (solver.w.copy(), solver.d.copy())
    return rel, states, solver


def pick_stopping_iteration(rel, lo, hi):
    """An iteration index K (0-based, lo <= K < hi) and a tol such that the reference's rule (:166-167) fires at K and
    nowhere before: max(ratios) of K is below that of EVERY earlier iteration (iteration 0 is never judged), and tol is the
    geometric mean of the two -- late in a run the ratios fall by 1/k per iteration, so the gap is small, but the engines
    reproduce the ratios to ~1e-9."""
    m = [max(a, b) for a, b in rel]
    for K in range(hi - 1, lo - 1, -1):      # the latest one (early in a run the primal ratio rises again for a while)
        before = min(m[1:K])
        if before > m[K] * (1 + 1e-5):
            return K, float(np.sqrt(before * m[K]))
    raise AssertionError("no usable stopping iteration in the window")


@pytest.mark.parametrize("n,iters", [(1200, 200), (2200, 100), (3000, 30)])
def test_converge_mode_at_mid_sizes_stops_where_the_oracle_stops(engine, n, iters):
    """VERDICT r03 'missing 2': converge mode on the TILE engine against the oracle's own stopping iteration beyond the
    golden sizes -- n = 1200 (chunks of 8 row pairs), n = 2200 (16, ragged) and n = 3000 (32, ragged) on smooth signals, tol set so
    that the reference's rule (smooth_learning.py:156-167) fires after some two hundred / eighty / thirty iterations:
    the same stopping iteration, the iterate and the dual variable of THAT iteration to 1e-10, the same ratios."""
    if engine != "resident":
        pytest.skip("the tile engine; the streaming engine's rule is checked at the golden sizes and against it at n = 5000 / 9300")
    x, _ = logmodel_problem(n, 256, seed=n)
    z = old.signal_distances(x)
    zt = z * old.theta_for_degree(squareform(z), 10)
    rel, states, _ = oracle_trajectory(zt, iters, snaps=(iters,))
    try:
        K, tol = pick_stopping_iteration(rel, iters - 1, iters)  # the last iteration, if the ratios are at their minimum there
    except AssertionError:                                       # else the latest earlier one, at the price of a second run
        K, tol = pick_stopping_iteration(rel, 8, iters)
        _, states, _ = oracle_trajectory(zt, K + 1, snaps=(K + 1,))
    with _lib.LogDegreeSession(n) as s:
        s.set_distances(zt)
        s.init(1, 1, 0.5, None)
        assert s.engine_kernel == "k_tile"
        n_it, conv, r = s.iterate(100000, tol)
        w, d = s.weights_condensed(), s.dual()
        assert s.engine == "resident"
    print(f"[converge n={n}] the oracle stops after {K + 1} iterations at tol {tol:.3e}; the tile engine after {n_it}")
    assert conv and n_it == K + 1
    np.testing.assert_allclose(w, states[K + 1][0], rtol=0, atol=ATOL)
    np.testing.assert_allclose(d, states[K + 1][1], rtol=0, atol=ATOL)
    assert r[0] == pytest.approx(rel[K][0], rel=1e-8) and r[1] == pytest.approx(rel[K][1], rel=1e-8)


@pytest.mark.parametrize("n", [700, 1000])
def test_warm_start_that_converges_early_in_a_second_call(engine, n):
    """VERDICT r03 'weak 2': the tile engine looks at the rule with a lag of one iteration while the ratios are far from tol,
    except during the first iterations of a call.  A warm start (edge_init = an almost converged iterate, the dual variable
    restarted from S w: smooth_learning.py:140-144) makes the ratios fall steeply; the first call runs 4 fixed iterations,
    the SECOND call runs to a tol that the oracle meets a few iterations into it: same stopping iteration as the oracle,
    iterate to 1e-10 -- on every engine."""
    if engine == "small" or (engine == "stream" and n > 800):
        pytest.skip("n > 128" if engine == "small" else "the streaming engine at the smaller size")
    x, _ = logmodel_problem(n, 256, seed=3 * n)
    z = old.signal_distances(x)
    zt = z * old.theta_for_degree(squareform(z), 10)
    _, st, _ = oracle_trajectory(zt, 150, snaps=(150,))
    warm = st[150][0]
    warm[warm < 1e-4] = 0.0
    rel, _, _ = oracle_trajectory(zt, 40, edge_init=warm)
    K, tol = pick_stopping_iteration(rel, 4 + 3, 4 + 12)      # inside the second call, beyond its exactly judged head
    _, states, _ = oracle_trajectory(zt, K + 1, edge_init=warm, snaps=(K + 1,))
    with _lib.LogDegreeSession(n) as s:
        s.set_distances(zt)
        s.init(1, 1, 0.5, warm)
        s.iterate(4, 0.0)
        n_it, conv, r = s.iterate(1000, tol)
        w, d = s.weights_condensed(), s.dual()
    print(f"[warm start n={n}] ratios of the oracle around the stop: {[f'{max(a, b):.3e}' for a, b in rel[K - 2:K + 1]]}; "
          f"stops {K + 1 - 4} iterations into the second call; engine {engine}: {n_it}")
    assert conv and n_it == K + 1 - 4
    np.testing.assert_allclose(w, states[K + 1][0], rtol=0, atol=ATOL)
    np.testing.assert_allclose(d, states[K + 1][1], rtol=0, atol=ATOL)


def test_round1_resident_engine_smoke(golden, engine, monkeypatch):
    """Round 1's grid-barrier engine (gl_resident.hip) stays selectable as the comparison point (GRAPHLEARN_AMD_ENGINE=resident1;
    the whole file runs on it with GRAPHLEARN_AMD_TEST_ALL=1): one golden case (fixed iterations, converge mode) and 60
    iterations at n = 2600 against the streaming engine in every default session."""
    if engine != "resident":
        pytest.skip("once per session")
    monkeypatch.setenv("GRAPHLEARN_AMD_ENGINE", "resident1")
    g = golden("logmodel_n64_d256")
    zt = g["z"] * float(g["theta_deg6"])
    with _lib.LogDegreeSession(64) as s:
        s.set_distances(zt)
        s.init(1, 1, 0.5, None)
        assert s.engine == "resident" and s.engine_kernel == "k_resident"
        s.iterate(50, 0.0)
        np.testing.assert_allclose(s.weights_condensed(), g["raw_w_it50"], rtol=0, atol=ATOL)
    w, info = gsp_learn_graph_log_degrees(zt, 1, 1, maxit=20000, tol=1e-5, edge_tol=-np.inf, return_info=True)
    assert info["converged"] and info["n_iter"] == int(g["n_iter_converged"])
    np.testing.assert_allclose(squareform(w, checks=False), g["raw_w_converged"], rtol=0, atol=ATOL)
    n = 2600
    z = np.random.default_rng(n).uniform(0.05, 3.0, size=n * (n - 1) // 2)
    out = {}
    for eng in ("stream", "resident1"):
        monkeypatch.setenv("GRAPHLEARN_AMD_ENGINE", eng)
        with _lib.LogDegreeSession(n) as s:
            s.set_distances(z)
            s.init(1.2, 0.8, 0.5, None)
            assert s.engine_kernel == ("k_edge" if eng == "stream" else "k_resident")
            s.iterate(60, 0.0)
            out[eng] = (s.weights_condensed(), s.dual())
    np.testing.assert_allclose(out["resident1"][0], out["stream"][0], rtol=0, atol=1e-13)
    np.testing.assert_allclose(out["resident1"][1], out["stream"][1], rtol=0, atol=1e-12)


@pytest.mark.parametrize("n,iters", [(300, 1500), (500, 900), (300, 3200)])
def test_long_converge_calls_run_through_the_dual_gate(engine, monkeypatch, capfd, n, iters):
    """Round 4: a long converge-mode call on the tile engine runs in three launches -- a head with the norms, a middle WITHOUT
    the primal norms that only watches the dual ratio (the rule needs both ratios below tol, smooth_learning.py:166-167, and the
    dual one holds it up for most of a run), a tail with the norms that judges every iteration.  Against the oracle: the same
    stopping iteration, the same iterate (1e-10) and ratios; and against the same call with the gate switched off: the same
    stopping iteration, iterates equal to 1e-13.  (The case of 3200 iterations: the middle ends ~2160 iterations into the
    run, where the dual ratio moves by < 0.03 % per iteration.)"""
    if engine != "resident":
        pytest.skip("the tile engine")
    x, _ = logmodel_problem(n, 256, seed=7 * n)
    z = old.signal_distances(x)
    zt = z * old.theta_for_degree(squareform(z), 6)
    rel, states, _ = oracle_trajectory(zt, iters, snaps=(iters,))
    try:
        K, tol = pick_stopping_iteration(rel, iters - 1, iters)  # the last iteration, if the ratios are at their minimum there
    except AssertionError:                                       # else the latest earlier one, at the price of a second run
        K, tol = pick_stopping_iteration(rel, 2 * iters // 3, iters)
        _, states, _ = oracle_trajectory(zt, K + 1, snaps=(K + 1,))
    assert rel[16][1] > 2 * tol                                   # the dual ratio is well above tol after the head: the gate opens
    out = {}
    for gate in ("1", "0"):
        monkeypatch.setenv("GRAPHLEARN_AMD_TILE_GATE", gate)
        monkeypatch.setenv("GRAPHLEARN_AMD_TILE_DEBUG", "1")
        with _lib.LogDegreeSession(n) as s:
            s.set_distances(zt)
            s.init(1, 1, 0.5, None)
            assert s.engine_kernel == "k_tile"
            n_it, conv, r = s.iterate(100000, tol)
            out[gate] = (n_it, conv, r, s.weights_condensed(), s.dual())
            assert s.engine == "resident"
        err = capfd.readouterr().err
        lines = [l for l in err.splitlines() if l.startswith("[tile gate]")]
        if gate == "1":
            assert len(lines) == 1, err[-300:]
            gate_line = lines[0]
        else:
            assert not lines
    n_it, conv, r, w, d = out["1"]
    gated_to = next((k for k in range(17, K) if rel[k][1] < 1.25 * tol), K)
    assert "3 launches" in gate_line, gate_line
    print(f"[dual gate n={n}] the oracle stops after {K + 1} iterations (tol {tol:.3e}); the dual ratio is above 1.25 tol until iteration {gated_to}; "
          f"the tile engine: {n_it} with the gate, {out['0'][0]} without")
    assert conv and n_it == K + 1 and out["0"][0] == K + 1
    np.testing.assert_allclose(w, states[K + 1][0], rtol=0, atol=ATOL)
    np.testing.assert_allclose(d, states[K + 1][1], rtol=0, atol=ATOL)
    assert r[0] == pytest.approx(rel[K][0], rel=1e-8) and r[1] == pytest.approx(rel[K][1], rel=1e-8)
    # (not bit for bit: every launch re-forms gamma S w = ybar - d from the stored dual state, where a single launch carries it
    #  through its recurrence -- three launches round differently from one, at the 1e-16 level)
    np.testing.assert_allclose(w, out["0"][3], rtol=0, atol=1e-13)
    np.testing.assert_allclose(d, out["0"][4], rtol=0, atol=1e-13)

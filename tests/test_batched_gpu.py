"""2-D (batched) path of the solver and WindowLogModel against the reference's golden vectors:
spectral-norm coupled stopping rule (smooth_learning.py:158-159 on 2-D arrays) and the
reference's own smoke test shapes (tests/test_models.py:26-28)."""

from itertools import product

import numpy as np
import pytest
from numpy.random import default_rng

from graph_learning_amd import gsp_learn_graph_log_degrees
from graph_learning_amd.temporal import WindowLogModel
from oracle import logdeg as old

pytestmark = pytest.mark.gpu


def test_batched_fixed_iterations(golden):
    g = golden("batched_B3_n12")
    out = gsp_learn_graph_log_degrees(g["z"].copy(), 1, 1, maxit=40, tol=0.0, edge_tol=-np.inf)
    assert out.shape == g["z"].shape
    np.testing.assert_allclose(out, g["raw_w_it40"], rtol=0, atol=1e-10)


def test_batched_coupled_stopping_rule(golden):
    g = golden("batched_B3_n12")
    out, info = gsp_learn_graph_log_degrees(g["z"].copy(), 1, 1, maxit=5000, tol=1e-4, edge_tol=1e-3, return_info=True)
    assert info["converged"] and info["n_iter"] == int(g["n_iter_converged"])   # (contract +-1; observed 0)
    ref = g["w_converged"]
    raw, _ = old.learn_log_degrees(g["z"].copy(), 1, 1, maxit=int(g["n_iter_converged"]), tol=0.0, edge_tol=-np.inf,
                                   return_info=True)
    near = np.abs(raw - 1e-3) < 1e-6
    np.testing.assert_allclose(out[~near], ref[~near], rtol=0, atol=1e-6)


def test_window_log_model_golden(golden):
    g = golden("batched_B3_n12")
    m = WindowLogModel(window_size=5, avg_degree=2).fit(g["win_x"])
    np.testing.assert_allclose(m.theta_, g["win_theta"], rtol=1e-9)
    assert m.weights_.shape == g["win_weights"].shape
    z = WindowLogModel(window_size=5, avg_degree=2)._initialize(g["win_x"])
    np.testing.assert_allclose(z, g["win_z"], rtol=1e-10, atol=1e-12)
    diff = np.abs(m.weights_ - g["win_weights"])
    assert (diff > 1e-6).mean() < 0.02      # only entries sitting at the edge_tol threshold may flip


@pytest.mark.parametrize("window_size, avg_degree", list(product([5, 10], [2, 5])))
def test_windowlogmodel_fit_reference_smoke(window_size, avg_degree):
    """tests/test_models.py:26-28 of the reference, same data and parameters."""
    data = default_rng(240920).standard_normal((23, 7))
    m = WindowLogModel(window_size=window_size, avg_degree=avg_degree, random_state=240920).fit(data)
    n_win = -(-23 // window_size)
    assert m.weights_.shape == (n_win, 21) and np.isfinite(m.weights_).all() and (m.weights_ >= 0).all()


def test_batched_errors():
    with pytest.raises(ValueError):
        gsp_learn_graph_log_degrees(np.ones((2, 10)), 1, 1, edge_init=np.ones((2, 10)))


@pytest.mark.parametrize("engine", ["stream", "auto"])
def test_batched_paths_agree_and_fall_back(golden, engine, monkeypatch):
    """The coupled loop has two device paths: one persistent launch with a workgroup per graph (n <= 128,
    the default here) and per-iteration batched launches (any n; forced with ENGINE=stream).  Both stop at
    the reference's iteration; when the persistent one cannot gather its workgroups (shared GPU, simulated)
    nothing has been modified and the other path takes over."""
    g = golden("batched_B3_n12")
    if engine == "stream":
        monkeypatch.setenv("GRAPHLEARN_AMD_ENGINE", "stream")
    else:
        monkeypatch.setenv("GRAPHLEARN_AMD_RES_TEST_NOT_RESIDENT", "1")  # rendezvous expects one workgroup too many
    out, info = gsp_learn_graph_log_degrees(g["z"].copy(), 1, 1, maxit=5000, tol=1e-4, edge_tol=1e-3, return_info=True)
    assert info["converged"] and info["n_iter"] == int(g["n_iter_converged"])   # (contract +-1; observed 0)
    raw, _ = old.learn_log_degrees(g["z"].copy(), 1, 1, maxit=int(g["n_iter_converged"]), tol=0.0, edge_tol=-np.inf,
                                   return_info=True)
    near = np.abs(raw - 1e-3) < 1e-6
    np.testing.assert_allclose(out[~near], g["w_converged"][~near], rtol=0, atol=1e-6)


_SIZES_ORACLE = {}


@pytest.mark.parametrize("rule", ["device", "host"])
@pytest.mark.parametrize("n,b", [(2, 3), (5, 1), (5, 300), (64, 5), (65, 4), (128, 3), (129, 3), (30, 20), (64, 16),
                                 (100, 32), (128, 20), (128, 32), (256, 8), (300, 5)])
def test_batched_sizes_against_oracle(n, b, rule, monkeypatch):
    """Both sides of the small-engine limits (64 / 128 nodes), a batch of one, tiny graphs, batches on both
    sides of what the in-kernel evaluation of the coupled rule takes (B <= 32, LDS permitting): same stopping
    iteration and iterates as the oracle's 2-D branch, with the rule's brackets tightened on the device or
    every ambiguous iteration handed to the host."""
    if rule == "host":
        monkeypatch.setenv("GRAPHLEARN_AMD_BATCH_HOST_RULE", "1")
    rng = default_rng(100 * n + b)
    e = n * (n - 1) // 2
    z = rng.uniform(0.1, 3.0, size=(b, e))
    if (n, b) not in _SIZES_ORACLE:      # (the oracle's answer serves both forms of the rule: its 300-graph case alone takes 9 s)
        _SIZES_ORACLE[(n, b)] = old.learn_log_degrees(z.copy(), 1.1, 0.9, maxit=400, tol=1e-4, edge_tol=-np.inf, return_info=True)
    ref, info = _SIZES_ORACLE[(n, b)]
    out, got = gsp_learn_graph_log_degrees(z, 1.1, 0.9, maxit=400, tol=1e-4, edge_tol=-np.inf, return_info=True)
    assert got["n_iter"] == info["n_iter"]
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-10)
    assert got["rel_primal"] == pytest.approx(info["rel"][0], rel=1e-8)
    assert got["rel_dual"] == pytest.approx(info["rel"][1], rel=1e-8)


@pytest.mark.parametrize("chunk", ["1", "5", "16"])
def test_batched_launch_path_chunks_and_budget(chunk, monkeypatch):
    """Above n = 128 the coupled loop runs as chunks of batched launches with the rule's screen on the device (the node
    pass's last workgroup raises every problem's done flag; the rest of the chunk falls through).  Whatever the chunk
    length: the iteration budget is kept exactly (maxit not a multiple of the chunk), a call with tol = 0 never stops
    early, a second call continues the first, and the stopping iteration of a run to tol equals the oracle's."""
    monkeypatch.setenv("GRAPHLEARN_AMD_BATCH_CHUNK", chunk)
    n, b = 200, 3
    rng = default_rng(5)
    z = rng.uniform(0.1, 3.0, size=(b, n * (n - 1) // 2))
    ref, info = old.learn_log_degrees(z, 1.0, 1.0, maxit=37, tol=0.0, edge_tol=-np.inf, return_info=True)
    out, got = gsp_learn_graph_log_degrees(z, 1.0, 1.0, maxit=37, tol=0.0, edge_tol=-np.inf, return_info=True)
    assert got["n_iter"] == 37 and not got["converged"]
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-10)
    assert got["rel_primal"] == pytest.approx(info["rel"][0], rel=1e-8)
    ref, info = old.learn_log_degrees(z, 1.0, 1.0, maxit=1000, tol=2e-4, edge_tol=-np.inf, return_info=True)
    out, got = gsp_learn_graph_log_degrees(z, 1.0, 1.0, maxit=1000, tol=2e-4, edge_tol=-np.inf, return_info=True)
    assert got["converged"] and got["n_iter"] == info["n_iter"]
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-10)


@pytest.mark.parametrize("n,b,hook", [(256, 8, None), (256, 20, None), (160, 30, None), (1000, 3, None), (256, 8, "overshoot"),
                                      (256, 8, "not_resident"), (256, 8, "off"), (300, 20, "no_room"), (256, 8, "alloc_cruise"),
                                      (256, 8, "alloc_ring")])
def test_batched_cruise_on_the_tile_engine(n, b, hook, monkeypatch, capfd):
    """Above n = 128 a batched call first cruises on the tile engine: the B graphs as ONE block-diagonal lattice in one
    persistent launch (k_tile<..., BATCH>), stopped by the screen of the coupled rule (lambda_max of a Gram matrix lies in
    [max diagonal, trace], smooth_learning.py:158-159,166-167) with 5 % to spare; the chunked launches with the exact
    spectral norms (Gram matrices on f64 MFMA, k_batch_gram) finish the run.  Same stopping iteration and iterates as the
    oracle's 2-D branch for: the headline case; 80 node blocks in total and two groups of 16 graphs (256, 20); thirty graphs
    of three node blocks on eight CUs each; a large graph; a cruise whose result is
    discarded (the screen overshot: redone from the sessions' own state); workgroups that are not co-resident; the cruise
    switched off; a batch the chip has no room for (300 nodes x 20 graphs: one graph needs 14 workgroups, has 12 CUs); a device
    that has no memory for the B-fold stand-in session (the allocation fails, the batched launches serve the call
    with a clean last-error word, nothing half-built stays cached) or for the ring of the deciding chunks."""
    monkeypatch.setenv("GRAPHLEARN_AMD_BATCH_DEBUG", "1")
    if hook == "overshoot":
        monkeypatch.setenv("GRAPHLEARN_AMD_BATCH_TEST_OVERSHOOT", "1")
    if hook == "not_resident":
        monkeypatch.setenv("GRAPHLEARN_AMD_RES_TEST_NOT_RESIDENT", "1")
    if hook == "off":
        monkeypatch.setenv("GRAPHLEARN_AMD_BATCH_NO_CRUISE", "1")
    if hook in ("alloc_cruise", "alloc_ring"):
        from graph_learning_amd import _lib
        _lib.default_context().trim()       # (no stand-in cached from an earlier test: the allocation is really asked for)
        monkeypatch.setenv("GRAPHLEARN_AMD_TEST_ALLOC_FAIL", hook[len("alloc_"):])
    rng = default_rng(1000 * n + b)
    z = rng.uniform(0.1, 3.0, size=(b, n * (n - 1) // 2))
    maxit = 300 if n < 1000 else 120
    ref, info = old.learn_log_degrees(z, 1.1, 0.9, maxit=maxit, tol=1e-4, edge_tol=-np.inf, return_info=True)
    out, got = gsp_learn_graph_log_degrees(z, 1.1, 0.9, maxit=maxit, tol=1e-4, edge_tol=-np.inf, return_info=True)
    err = capfd.readouterr().err
    cruised = [l for l in err.splitlines() if l.startswith("[batch cruise]")]
    if hook in ("off", "no_room", "alloc_cruise"):
        assert not cruised
    elif hook == "alloc_ring":
        assert len(cruised) == 1 and cruised[0].rstrip().endswith("kept 1") and "[batch ring]" not in err
    else:
        assert len(cruised) == 1 and f"n {n} B {b}:" in cruised[0]
        kept = cruised[0].rstrip().endswith("kept 1")
        assert kept == (hook is None), cruised[0]
        if hook is None:
            ran = int(cruised[0].split("ran ")[1].split()[0])
            assert 0.5 * info["n_iter"] < ran < info["n_iter"], cruised[0]   # most of the run, never the iteration that stops it
    assert got["n_iter"] == info["n_iter"] and got["converged"] == (info["n_iter"] < maxit)
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-10)
    assert got["rel_primal"] == pytest.approx(info["rel"][0], rel=1e-8)
    assert got["rel_dual"] == pytest.approx(info["rel"][1], rel=1e-8)
    # a second call on the cached stand-in session: bit-identical
    out2 = gsp_learn_graph_log_degrees(z, 1.1, 0.9, maxit=maxit, tol=1e-4, edge_tol=-np.inf)
    assert np.array_equal(out, out2)
    if hook == "alloc_cruise":   # with memory again the very next call cruises: the failed attempt left nothing behind
        monkeypatch.delenv("GRAPHLEARN_AMD_TEST_ALLOC_FAIL")
        capfd.readouterr()
        out3 = gsp_learn_graph_log_degrees(z, 1.1, 0.9, maxit=maxit, tol=1e-4, edge_tol=-np.inf)
        assert "[batch cruise]" in capfd.readouterr().err
        np.testing.assert_allclose(out3, ref, rtol=0, atol=1e-10)


@pytest.mark.parametrize("ring_k", ["2", "7", "32", "off"])
@pytest.mark.parametrize("n,b,tol,maxit", [(256, 8, 1e-4, 2000), (200, 20, 3e-4, 2000), (256, 6, 1e-4, 150), (320, 3, 0.0, 90)])
def test_batched_deciding_chunks_on_the_ring(n, b, tol, maxit, ring_k, monkeypatch, capfd):
    """Behind the cruise the exact rule is judged in chunks: the tile engine runs K iterations per launch and writes every
    iterate to a ring, one launch (k_batch_gram_ring) judges all K exactly (Gram matrices on MFMA, rigorous brackets), and the
    first iteration that meets the rule ends the chunk -- its state is rebuilt from the chunk's start.  Whatever K (2: a replay
    nearly every time; 7: chunks that do not divide the run; 32: the default; off: the batched launches instead): the oracle's
    stopping iteration, iterates and ratios -- also when the budget ends the run inside a chunk, and with tol = 0."""
    monkeypatch.setenv("GRAPHLEARN_AMD_BATCH_DEBUG", "1")
    if ring_k == "off":
        monkeypatch.setenv("GRAPHLEARN_AMD_BATCH_NO_RING", "1")
    else:
        monkeypatch.setenv("GRAPHLEARN_AMD_BATCH_RING_K", ring_k)
    rng = default_rng(77 * n + b)
    z = rng.uniform(0.1, 3.0, size=(1, n * (n - 1) // 2)) * rng.uniform(0.85, 1.15, size=(b, n * (n - 1) // 2))   # alike: long deciding stretch
    ref, info = old.learn_log_degrees(z, 1.0, 1.0, maxit=maxit, tol=tol, edge_tol=-np.inf, return_info=True)
    out, got = gsp_learn_graph_log_degrees(z, 1.0, 1.0, maxit=maxit, tol=tol, edge_tol=-np.inf, return_info=True)
    err = capfd.readouterr().err
    rings = [l for l in err.splitlines() if l.startswith("[batch ring]")]
    assert (len(rings) == 1) == (ring_k != "off"), err[-400:]
    assert got["n_iter"] == info["n_iter"] and got["converged"] == (info["n_iter"] < maxit)
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-10)
    assert got["rel_primal"] == pytest.approx(info["rel"][0], rel=1e-8)
    assert got["rel_dual"] == pytest.approx(info["rel"][1], rel=1e-8)
    if ring_k != "off":
        assert "back to the launches 0" in rings[0] and f"at iteration {info['n_iter']}," in rings[0], rings[0]


def test_batched_deciding_chunk_that_fails_hands_over_to_the_launches(monkeypatch, capfd):
    """A deciding chunk whose launch cannot run (test hook: the second one is treated as not co-resident) puts the chunk's start
    state back and hands the batch -- both iterates -- to the batched launches, which finish the run: same result."""
    monkeypatch.setenv("GRAPHLEARN_AMD_BATCH_DEBUG", "1")
    monkeypatch.setenv("GRAPHLEARN_AMD_BATCH_TEST_RING_FAIL", "1")
    monkeypatch.setenv("GRAPHLEARN_AMD_BATCH_RING_K", "8")
    n, b = 256, 6
    rng = default_rng(3)
    z = rng.uniform(0.1, 3.0, size=(1, n * (n - 1) // 2)) * rng.uniform(0.85, 1.15, size=(b, n * (n - 1) // 2))
    ref, info = old.learn_log_degrees(z, 1.0, 1.0, maxit=2000, tol=1e-4, edge_tol=-np.inf, return_info=True)
    out, got = gsp_learn_graph_log_degrees(z, 1.0, 1.0, maxit=2000, tol=1e-4, edge_tol=-np.inf, return_info=True)
    err = capfd.readouterr().err
    assert "back to the launches 1" in err and "[batch ring] 1 chunks" in err, err[-300:]
    assert got["n_iter"] == info["n_iter"] and got["converged"]
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-10)
    assert got["rel_primal"] == pytest.approx(info["rel"][0], rel=1e-8)


def test_window_model_large_windows_theta_on_device_and_session_pool():
    """Windows of n = 200 nodes: theta never leaves the device before the distances are scaled (the finish of get_theta --
    NumPy's pairwise means, smooth_learning.py:204-211 -- by one device thread, any n), the sessions' device blocks come from and
    go back to the context's pool (two fits in a row, then another size in between): thetas and weights against the oracle."""
    from scipy.spatial.distance import pdist, squareform

    rng = default_rng(11)
    n, win, deg = 200, 64, 5
    x = rng.standard_normal((3 * win, n)).cumsum(axis=1) * 0.1 + rng.standard_normal((3 * win, n)) * 0.05
    d = np.stack([pdist(w.T) ** 2 for w in x.reshape(-1, win, n)])
    th = np.array([old.theta_for_degree(squareform(dd), deg) for dd in d])
    ref, info = old.learn_log_degrees(d * th[:, None], 1, 1, maxit=300, tol=1e-4, edge_tol=-np.inf, return_info=True)
    for rep in range(2):
        m = WindowLogModel(window_size=win, avg_degree=deg, maxit=300, tol=1e-4, edge_tol=-np.inf).fit(x)
        np.testing.assert_allclose(m.theta_[:, 0], th, rtol=1e-12)
        assert m.n_iter_ == info["n_iter"]
        np.testing.assert_allclose(m.weights_, ref, rtol=0, atol=1e-9)
        if rep == 0:  # another session size between the two fits
            gsp_learn_graph_log_degrees(rng.uniform(0.1, 3.0, size=(2, 30 * 29 // 2)), 1, 1, maxit=20, tol=0.0)


def test_batch_gram_entry_point():
    """gl_logdeg_batch_gram (diagnostic entry point): the four B x B Gram matrices of the last iteration."""
    import ctypes as C

    from graph_learning_amd import _lib

    n, b = 12, 3
    e = n * (n - 1) // 2
    z = default_rng(7).uniform(0.2, 2.0, size=(b, e))
    ctx = _lib.default_context()
    sessions = [_lib.LogDegreeSession(n, ctx) for _ in range(b)]
    try:
        prev_w, prev_d = [], []
        for s, zz in zip(sessions, z):
            s.set_distances(zz)
            s.init(1, 1, 0.5, None)
            s.iterate_async(4)
            prev_w.append(s.weights_condensed())
            prev_d.append(s.dual())
            s.iterate_async(1)
        new_w = [s.weights_condensed() for s in sessions]
        new_d = [s.dual() for s in sessions]
        handles = (C.c_void_p * b)(*[s._h for s in sessions])
        gram = np.empty((4, b, b))
        ctx.check(ctx._lib.gl_logdeg_batch_gram(handles, b, _lib.ptr(gram)), "batch_gram")
    finally:
        for s in sessions:
            s.close()
    dw = np.array(new_w) - np.array(prev_w)
    dd = np.array(new_d) - np.array(prev_d)
    for got, want in zip(gram, (dw @ dw.T, np.array(prev_w) @ np.array(prev_w).T, dd @ dd.T,
                                np.array(prev_d) @ np.array(prev_d).T)):
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-14)


def test_stopping_state_comes_out_of_the_rings_not_out_of_a_replay(monkeypatch, capfd):
    """Where a deciding chunk is stopped in mid-chunk the state of the stopping iteration -- iterate, dual variable, ybar, pbar,
    dprev and the node blocks' norm partials -- is read back from the rings the chunk wrote (round 4); before, a launch of exactly
    that many iterations from the chunk's start rebuilt it (GRAPHLEARN_AMD_BATCH_REPLAY=1).  Both ways must leave the sessions
    in the same state bit for bit: the weights, the dual variable, and -- what only shows when a session goes on alone, with
    its own lagged rule reading those norm partials -- every later iterate and the iteration a continued run stops at."""
    from graph_learning_amd import _lib
    from graph_learning_amd.batched import _run_sessions

    n, b = 256, 6
    e = n * (n - 1) // 2
    rng = default_rng(11)
    z = rng.uniform(0.05, 4.0, size=(1, e)) * rng.uniform(0.9, 1.1, size=(b, e))   # (one pattern: a long deciding stretch)
    ctx = _lib.default_context()
    results = {}
    for mode in ("rings", "replay"):
        if mode == "replay":
            monkeypatch.setenv("GRAPHLEARN_AMD_BATCH_REPLAY", "1")
        monkeypatch.setenv("GRAPHLEARN_AMD_BATCH_DEBUG", "1")
        sessions = [_lib.LogDegreeSession(n, ctx) for _ in range(b)]
        try:
            for s, zz in zip(sessions, z):
                s.set_distances(zz)
                s.init(1.0, 1.0, 0.5, None)
            out, info = _run_sessions(sessions, ctx, 5000, 1e-5, -np.inf)
            err = capfd.readouterr().err
            ring = [l for l in err.splitlines() if l.startswith("[batch ring]")]
            assert len(ring) == 1 and "finished 1" in ring[0], err
            assert (" 0 replays" in ring[0]) == (mode == "rings"), ring[0]
            duals = [s.dual() for s in sessions]
            cont = [s.iterate(400, 2e-6) for s in sessions]           # each graph goes on alone, with its own rule
            later = [s.weights_condensed() for s in sessions]
        finally:
            for s in sessions:
                s.close()
        results[mode] = (out, info, duals, cont, later)
    a, r = results["rings"], results["replay"]
    assert a[1] == r[1] and np.array_equal(a[0], r[0])
    ref, oinfo = old.learn_log_degrees(z, 1.0, 1.0, maxit=5000, tol=1e-5, edge_tol=-np.inf, return_info=True)
    assert a[1]["n_iter"] == oinfo["n_iter"]
    np.testing.assert_allclose(a[0], ref, rtol=0, atol=1e-10)
    for k in range(b):
        assert np.array_equal(a[2][k], r[2][k])
        assert a[3][k] == r[3][k] and a[3][k][0] > 0
        assert np.array_equal(a[4][k], r[4][k])

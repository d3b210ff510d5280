"""Dev tool: the batched (2-D) solver above n = 128 (cruise on the tile engine, then chunked batched launches with the coupled
rule's screen on the device): us per iteration with tol = 0 (bare loop) and run to tol, against the oracle's stopping iteration.
usage: python tools/batched_large_bench.py [n:B[:kind] ...]     kind: uniform (default) | indep | similar
  indep / similar: distances of smooth signals on B different random graphs / on one graph (windows of one process): runs of
  ~1000 iterations whose coupled rule stays undecided by the cheap bounds for 5-60 % of them"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np
from graph_learning_amd import gsp_learn_graph_log_degrees
from oracle import logdeg as old


def smooth_distances(n, b, kind, d=64):
    rng = np.random.default_rng(n + b)
    def chol(seed):
        r = np.random.default_rng(seed)
        a = np.triu((r.random((n, n)) < 6.0 / n) * r.uniform(0.3, 2, (n, n)), 1)
        a = a + a.T
        return np.linalg.cholesky(np.linalg.inv(np.diag(a.sum(1)) - a + np.eye(n)))
    iu = np.triu_indices(n, 1)
    out = []
    for g in range(b):
        x = chol(1 if kind == "similar" else 10 + g) @ rng.standard_normal((n, d))
        sq = (x * x).sum(1)
        dist = np.maximum(sq[:, None] + sq[None, :] - 2 * x @ x.T, 0)
        out.append(dist[iu] * old.theta_for_degree(dist, 6))
    return np.array(out)


cases = [a.split(":") for a in sys.argv[1:] if not a.startswith("--")] or [["256", "8"], ["256", "20"], ["512", "8"], ["1000", "4"]]
for c in cases:
    n, b, kind = int(c[0]), int(c[1]), (c[2] if len(c) > 2 else "uniform")
    z = np.random.default_rng(n).uniform(0.2, 3.0, size=(b, n * (n - 1) // 2)) if kind == "uniform" else smooth_distances(n, b, kind)
    gsp_learn_graph_log_degrees(z, 1, 1, maxit=20, tol=0.0)
    gsp_learn_graph_log_degrees(z, 1, 1, maxit=100, tol=0.0)    # (builds the cruise's stand-in session, kept by the context)
    t0 = time.perf_counter(); gsp_learn_graph_log_degrees(z, 1, 1, maxit=0); tsetup = time.perf_counter() - t0
    t0 = time.perf_counter(); gsp_learn_graph_log_degrees(z, 1, 1, maxit=1000, tol=0.0); tb = time.perf_counter() - t0
    t0 = time.perf_counter(); out, info = gsp_learn_graph_log_degrees(z, 1, 1, maxit=20000, tol=1e-4, return_info=True); tg = time.perf_counter() - t0
    line = (f"n={n} B={b} {kind}: bare loop {(tb - tsetup) / 1000 * 1e6:.1f} us/it | to tol 1e-4: {info['n_iter']} its, "
            f"{(tg - tsetup) / info['n_iter'] * 1e6:.1f} us/it, {(tg - tsetup) * 1e3:.2f} ms (set-up {tsetup * 1e3:.1f} ms)")
    if n <= 300 and "--no-oracle" not in sys.argv:
        _, oinfo = old.learn_log_degrees(z, 1, 1, maxit=20000, tol=1e-4, return_info=True)
        line += f" | oracle stops at {oinfo['n_iter']}"
    print(line, flush=True)

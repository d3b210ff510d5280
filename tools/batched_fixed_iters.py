"""Batched small-graph solver: time per iteration of the bare loop (tol = 0: the coupled rule never comes
close, no tightening, no host) next to a run to tolerance."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graph_learning_amd import gsp_learn_graph_log_degrees

for n, b in [(16, 8), (64, 10), (100, 20), (128, 8)]:
    rng = np.random.default_rng(n)
    z = rng.uniform(0.2, 3.0, size=(b, n * (n - 1) // 2))
    gsp_learn_graph_log_degrees(z, 1, 1, maxit=5)
    t0 = time.perf_counter(); gsp_learn_graph_log_degrees(z, 1, 1, maxit=0); ts = time.perf_counter() - t0
    res = []
    for maxit, tol in [(4000, 0.0), (4000, 1e-5), (4000, 1e-7)]:
        t0 = time.perf_counter()
        _, info = gsp_learn_graph_log_degrees(z, 1, 1, maxit=maxit, tol=tol, return_info=True)
        dt = time.perf_counter() - t0 - ts
        res.append(f"tol={tol:g}: {info['n_iter']} its, {dt * 1e3:.2f} ms ({dt / info['n_iter'] * 1e6:.1f} us/it)")
    print(f"n={n} B={b}: setup {ts * 1e3:.2f} ms | " + " | ".join(res), flush=True)

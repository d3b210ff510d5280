"""Dev tool: the batched (2-D) solver above n = 128 (chunked batched launches, the coupled rule's screen on the device):
us per iteration with tol = 0 (bare loop) and run to tol, against the oracle's stopping iteration.
usage: python tools/batched_large_bench.py [n:B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np
from graph_learning_amd import gsp_learn_graph_log_degrees
from oracle import logdeg as old

cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(256, 8), (256, 20), (512, 8), (1000, 4)]
for n, b in cases:
    z = np.random.default_rng(n).uniform(0.2, 3.0, size=(b, n * (n - 1) // 2))
    gsp_learn_graph_log_degrees(z, 1, 1, maxit=20, tol=0.0)
    t0 = time.perf_counter(); gsp_learn_graph_log_degrees(z, 1, 1, maxit=0); tsetup = time.perf_counter() - t0
    t0 = time.perf_counter(); gsp_learn_graph_log_degrees(z, 1, 1, maxit=1000, tol=0.0); tb = time.perf_counter() - t0
    t0 = time.perf_counter(); out, info = gsp_learn_graph_log_degrees(z, 1, 1, maxit=20000, tol=1e-4, return_info=True); tg = time.perf_counter() - t0
    line = (f"n={n} B={b}: bare loop {(tb - tsetup) / 1000 * 1e6:.1f} us/it | to tol 1e-4: {info['n_iter']} its, "
            f"{(tg - tsetup) / info['n_iter'] * 1e6:.1f} us/it (set-up {tsetup * 1e3:.1f} ms)")
    if n <= 300:
        _, oinfo = old.learn_log_degrees(z, 1, 1, maxit=20000, tol=1e-4, return_info=True)
        line += f" | oracle stops at {oinfo['n_iter']}"
    print(line, flush=True)

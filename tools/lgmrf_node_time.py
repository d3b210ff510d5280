"""Dev tool: ms per LGMRF node update (fast path) as the difference of two capped fits, narrow and wide tagged ADMM forms.
usage: python tools/lgmrf_node_time.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.linalg import solve_triangular
from graph_learning_amd import LGMRF, _lib
from graph_learning_amd.synth import er_laplacian

ctx = _lib.default_context()
for n in [int(a) for a in sys.argv[1:]] or [512, 1024, 2048]:
    d = n // 4
    rng = np.random.default_rng(7)
    lap = er_laplacian(n, 6.0 / n, 0.3, 2.0, seed=rng)
    xs = solve_triangular(np.linalg.cholesky(lap + np.eye(n)), rng.standard_normal((n, d)), lower=True, trans="T").T
    cov = xs.T @ xs / d
    for wide in ("0", "1"):
        os.environ["GRAPHLEARN_AMD_LGMRF_WIDE"] = wide
        def capped(k):
            os.environ["GRAPHLEARN_AMD_LGMRF_MAX_NODES"] = str(k)
            t0 = time.perf_counter()
            LGMRF(max_cycle=1).fit(cov, ctx=ctx)
            return time.perf_counter() - t0
        capped(8)
        lo, hi = 64, min(192, n)
        dt = min(capped(hi) - capped(lo) for _ in range(2))
        print(f"n={n} wide={wide}: {dt * 1e3 / (hi - lo):.3f} ms per node update", flush=True)
    del os.environ["GRAPHLEARN_AMD_LGMRF_MAX_NODES"], os.environ["GRAPHLEARN_AMD_LGMRF_WIDE"]

"""Dev tool: ms per LGMRF node update (fast path, tagged ADMM) with the fused node update on / off and the rows of the
ADMM kernel half in LDS / all in registers, same process.
usage: python tools/lgmrf_ab.py [n ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.linalg import solve_triangular
from graph_learning_amd import _lib
from graph_learning_amd.lgmrf import time_node_updates
from graph_learning_amd.synth import er_laplacian

ctx = _lib.default_context()
for n in [int(a) for a in sys.argv[1:]] or [512, 1024, 2048]:
    d = n // 4
    rng = np.random.default_rng(7)
    lap = er_laplacian(n, 6.0 / n, 0.3, 2.0, seed=rng)
    xs = solve_triangular(np.linalg.cholesky(lap + np.eye(n)), rng.standard_normal((n, d)), lower=True, trans="T").T
    cov = xs.T @ xs / d
    for fuse, regs in (("1", "0"), ("1", "1"), ("0", "0"), ("1", "0"), ("1", "1")):
        os.environ["GRAPHLEARN_AMD_LGMRF_FUSE3"] = fuse
        os.environ["GRAPHLEARN_AMD_LGMRF_REGS"] = regs
        time_node_updates(cov, 0, 8, ctx=ctx)
        lo, hi = 64, min(192, n)
        dt = min(time_node_updates(cov, lo, hi - lo, ctx=ctx) for _ in range(2))
        print(f"n={n} fused node update {fuse}, all rows in registers {regs}: {dt * 1e3 / (hi - lo):.3f} ms per node update", flush=True)
    del os.environ["GRAPHLEARN_AMD_LGMRF_FUSE3"], os.environ["GRAPHLEARN_AMD_LGMRF_REGS"]

"""Dev tool: ms per LGMRF node update (n = 2048 by default, fast path, fused node update) for variant builds of the library
(tools/build_variant_lgmrf.sh), one process each.   usage: python tools/lgmrf_variants.py [n] NAME[:ENV=VAL,...] ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
n = int(args.pop(0)) if args and args[0].isdigit() else 2048
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from scipy.linalg import solve_triangular
from graph_learning_amd import _lib
from graph_learning_amd.lgmrf import time_node_updates
from graph_learning_amd.synth import er_laplacian
n = %d
ctx = _lib.default_context()
rng = np.random.default_rng(7)
lap = er_laplacian(n, 6.0 / n, 0.3, 2.0, seed=rng)
xs = solve_triangular(np.linalg.cholesky(lap + np.eye(n)), rng.standard_normal((n, n // 4)), lower=True, trans="T").T
cov = xs.T @ xs / (n // 4)
time_node_updates(cov, 0, 8, ctx=ctx)
dt = min(time_node_updates(cov, 64, 128, ctx=ctx) for _ in range(3))
print(f"{dt * 1e3 / 128:.4f} ms per node update")
''' % (ROOT, n)
for spec in args:
    name, _, envs = spec.partition(":")
    env = dict(os.environ)
    for kv in filter(None, envs.split(",")):
        k, v = kv.split("=")
        env[k] = v
    if name != "shipped":
        env["GRAPHLEARN_AMD_LIB"] = os.path.join(ROOT, "graph_learning_amd", "csrc", "build", "var", name + ".so")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(f"{spec:60s} n={n}: {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)

"""Dev tool: the distance set-up (k_gram and friends) at n x d; run under rocprofv3 --kernel-trace --stats for the
per-kernel times.  usage: python tools/gram_bench.py [n=4096] [d=1024] [reps=5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graph_learning_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ctx = _lib.default_context()
x = np.random.default_rng(0).standard_normal((d, n))
with _lib.LogDegreeSession(n, ctx) as s:
    for r in range(reps):
        t0 = time.perf_counter()
        theta = s.set_signals(x, 10)
        dt = time.perf_counter() - t0
        print(f"n={n} d={d}: set_signals {dt*1e3:.2f} ms (upload + centring + Gram + distances + theta), theta {theta:.6g}", flush=True)
    z = s.get_distances()
    i, j = 17, n - 5
    ref = ((x[:, i] - x[:, j]) ** 2).sum()
    from scipy.spatial.distance import squareform
    print("spot check", squareform(z)[i, j] if n <= 2048 else "-", ref)

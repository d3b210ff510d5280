"""Dev tool: event-timed iterations/s of the resident LogModel loop (no data generation cost).
usage: python tools/iter_bench.py [n=4096] [steps=1000]   (env: GRAPHLEARN_AMD_BPC, GRAPHLEARN_AMD_NO_GRAPH)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graph_learning_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ctx = _lib.Context(0)
e = n * (n - 1) // 2
z = np.random.default_rng(0).uniform(0.5, 2.0, size=e)
s = _lib.LogDegreeSession(n, ctx)
s.set_distances(z); s.init(1, 1, 0.5, None)
s.iterate_async(128); ctx.sync()
res = []
for _ in range(3):
    t0 = time.perf_counter(); ctx.timer_start(); s.iterate_async(steps); t1 = time.perf_counter(); ms = ctx.timer_stop(); t2 = time.perf_counter()
    res.append((1e3 * ms / steps, 1e6 * (t2 - t0) / steps, 1e6 * (t1 - t0) / steps))
best = min(res)
by = 12 * n * (n - 1)
print(f"n={n} steps={steps} WPC={os.environ.get('GRAPHLEARN_AMD_WPC','8')} nograph={os.environ.get('GRAPHLEARN_AMD_NO_GRAPH','0')}: "
      f"gpu {best[0]:.2f} us/it wall {best[1]:.2f} us/it enqueue {best[2]:.2f} us/it -> {1e6/best[0]:.0f} it/s, algo {by/best[0]/1e6:.2f} TB/s ({by/best[0]/8e6*100:.1f}% of 8 TB/s)")

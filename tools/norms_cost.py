"""Dev tool: cost of the norms / of the stopping rule in the tile engine at n = 4096 (us per iteration)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graph_learning_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = _lib.default_context()
z = np.random.default_rng(n).uniform(0.05, 3.0, size=n * (n - 1) // 2)
with _lib.LogDegreeSession(n, ctx) as s:
    s.set_distances(z)
    s.init(1, 1, 0.5, None)
    s.iterate(100, 0.0)
    for name, fn in (("async (no norms)", lambda: (s.iterate_async(2000), ctx.sync())),
                     ("sync tol=0 (norms, rule only at the end)", lambda: s.iterate(2000, 0.0)),
                     ("sync tol=1e-30 (norms + rule every iteration)", lambda: s.iterate(2000, 1e-30))):
        s.init(1, 1, 0.5, None)
        fn()
        s.init(1, 1, 0.5, None)
        ctx.sync()
        t = time.perf_counter(); fn(); dt = time.perf_counter() - t
        print(f"n={n} {name}: {dt * 1e6 / 2000:.2f} us per iteration", flush=True)

"""Resident vs streaming engine: microseconds per iteration (steady state, after a warm-up call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graph_learning_amd import _lib

sizes = [int(a) for a in sys.argv[1:]] or [64, 512, 1024, 2048, 3072, 4096]
for n in sizes:
    row = []
    for engine in ("stream", "resident", "small"):
        os.environ["GRAPHLEARN_AMD_ENGINE"] = engine
        ctx = _lib.Context(0)
        e = n * (n - 1) // 2
        z = np.random.default_rng(0).uniform(0.5, 2.0, size=e)
        s = _lib.LogDegreeSession(n, ctx)
        s.set_distances(z); s.init(1, 1, 0.5, None)
        s.iterate_async(200); ctx.sync()
        ctx.timer_start(); s.iterate_async(2000); ms = ctx.timer_stop()
        row.append(ms)
        s.close(); ctx.close()
    print(f"n={n}: stream {row[0]/2:.2f} us/it   resident {row[1]/2:.2f} us/it   small (n <= 128, else automatic) {row[2]/2:.2f} us/it", flush=True)

"""Dev diagnostic: host enqueue time vs GPU completion time of iterate_async."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graph_learning_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = _lib.Context(0)
e = n * (n - 1) // 2
z = np.random.default_rng(0).uniform(0.5, 2.0, size=e)
s = _lib.LogDegreeSession(n, ctx)
s.set_distances(z); s.init(1, 1, 0.5, None)
s.iterate_async(50); ctx.sync()
for steps in (100, 200, 400, 500, 1000, 2000):
    t0 = time.perf_counter(); s.iterate_async(steps); t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
    ctx.timer_start(); s.iterate_async(steps); ms = ctx.timer_stop(); t3 = time.perf_counter()
    print(f"steps {steps:5d}: enqueue {1e6*(t1-t0)/steps:7.1f} us/it, until done {1e6*(t2-t0)/steps:7.1f} us/it | with events: gpu {1e3*ms/steps:7.1f} us/it wall {1e6*(t3-t2)/steps:7.1f} us/it")

import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graph_learning_amd import _lib
n = 4096
ctx = _lib.Context(0)
e = n * (n - 1) // 2
z = np.random.default_rng(0).uniform(0.5, 2.0, size=e)
s = _lib.LogDegreeSession(n, ctx)
s.set_distances(z); s.init(1, 1, 0.5, None)
s.iterate_async(100); ctx.sync()
for steps in (2000, 2000, 300, 1000, 2000, 4000, 64, 128, 640):
    t0 = time.perf_counter(); ctx.timer_start(); s.iterate_async(steps); ms = ctx.timer_stop(); t1 = time.perf_counter()
    print(f"steps {steps:5d}: gpu {1e3*ms/steps:7.2f} us/it  wall {1e6*(t1-t0)/steps:7.2f} us/it", flush=True)

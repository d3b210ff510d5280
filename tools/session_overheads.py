import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from graph_learning_amd import _lib
ctx = _lib.default_context()
for n in (30, 64, 100):
    e = n*(n-1)//2
    z = np.random.default_rng(0).uniform(0.5, 2, e)
    s=_lib.LogDegreeSession(n, ctx); s.close()
    t0=time.perf_counter()
    for _ in range(20):
        s=_lib.LogDegreeSession(n, ctx)
        s.close()
    t1=time.perf_counter()
    ss=[_lib.LogDegreeSession(n, ctx) for _ in range(20)]
    t2=time.perf_counter()
    for s in ss: s.set_distances(z)
    t3=time.perf_counter()
    for s in ss: s.init(1,1,0.5,None)
    t4=time.perf_counter()
    for s in ss: s.weights_condensed(1e-3)
    t5=time.perf_counter()
    for s in ss: s.close()
    t6=time.perf_counter()
    x=np.random.default_rng(1).standard_normal((10,n))
    s=_lib.LogDegreeSession(n, ctx)
    t7=time.perf_counter()
    for _ in range(20): s.set_signals(x, 3); s.get_distances()
    t8=time.perf_counter()
    s.close()
    print(f"n={n}: create+destroy {1e3*(t1-t0)/20:.3f} ms, create {1e3*(t2-t1)/20:.3f}, set_distances {1e3*(t3-t2)/20:.3f}, init {1e3*(t4-t3)/20:.3f}, weights out {1e3*(t5-t4)/20:.3f}, close {1e3*(t6-t5)/20:.3f}, set_signals+get_distances {1e3*(t8-t7)/20:.3f} ms each")

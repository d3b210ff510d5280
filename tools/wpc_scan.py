import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from graph_learning_amd import _lib
n = int(sys.argv[1])
rng = np.random.default_rng(0)
z = rng.uniform(0.2, 3.0, size=n*(n-1)//2)
ctx = _lib.default_context()
with _lib.LogDegreeSession(n, ctx) as s:
    s.set_distances(z); s.init(1, 1, 0.5, None)
    s.iterate_async(50); ctx.sync()
    ctx.timer_start(); s.iterate_async(200); ms = ctx.timer_stop()
    print(f"n={n} WPC={os.environ.get('GRAPHLEARN_AMD_WPC')}: {ms/200*1e3:.1f} us/it, edge pass alone {s.time_edge_pass(50):.1f} us")

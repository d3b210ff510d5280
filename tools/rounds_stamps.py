"""Dev tool: phase stamps of the tile engine's rounds at one size.  usage: python tools/rounds_stamps.py n"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GRAPHLEARN_AMD_RES_STAMPS"] = "1"
import numpy as np
from graph_learning_amd import _lib
n = int(sys.argv[1])
ctx = _lib.default_context()
z = np.random.default_rng(n).uniform(0.05, 3.0, size=n * (n - 1) // 2)
with _lib.LogDegreeSession(n, ctx) as s:
    s.set_distances(z); s.init(1, 1, 0.5, None)
    s.iterate_async(100); s.dual()
    ctx.timer_start(); s.iterate_async(500); print(n, s.engine_kernel, ctx.timer_stop() * 1e3 / 500, "us/it", flush=True)
    s.dual()

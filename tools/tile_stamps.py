"""Dev tool: phase stamps of the tile engine.  usage: GRAPHLEARN_AMD_RES_STAMPS=1 python tools/tile_stamps.py n [n ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GRAPHLEARN_AMD_RES_STAMPS"] = "1"
os.environ.setdefault("GRAPHLEARN_AMD_ENGINE", "tile")
import numpy as np
from graph_learning_amd import _lib
ctx = _lib.default_context()
for n in [int(a) for a in sys.argv[1:]] or [130, 2048, 4096]:
    rng = np.random.default_rng(n)
    z = rng.uniform(0.05, 3.0, size=n * (n - 1) // 2)
    with _lib.LogDegreeSession(n, ctx) as s:
        s.set_distances(z)
        s.init(1, 1, 0.5, None)
        s.iterate(100, 0.0)
        print(n, s.engine_kernel, flush=True)
        s.iterate(100, 0.0)      # stamps of this launch (with norms) are printed by the library
        print("-- with the stopping rule (tol = 1e-30):", flush=True)
        s.iterate(100, 1e-30)
        print("-- without norms:", flush=True)
        s.iterate_async(100)     # ... and of the variant without norms, at the next synchronising call
        s.dual()

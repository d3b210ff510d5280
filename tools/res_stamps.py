"""Phase breakdown of the resident engine (GRAPHLEARN_AMD_RES_STAMPS=1 must be set)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GRAPHLEARN_AMD_RES_STAMPS", "1")
os.environ.setdefault("GRAPHLEARN_AMD_ENGINE", "resident")
import numpy as np
from graph_learning_amd import _lib
for n in [int(a) for a in sys.argv[1:]] or [4096]:
    ctx = _lib.Context(0)
    z = np.random.default_rng(0).uniform(0.5, 2.0, size=n * (n - 1) // 2)
    with _lib.LogDegreeSession(n, ctx) as s:
        s.set_distances(z); s.init(1, 1, 0.5, None)
        print(f"n={n} engine={s.engine}", flush=True)
        s.iterate(200, 0.0)
    ctx.close()

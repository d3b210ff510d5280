"""Dev tool: phase stamps of the batched cruise (B graphs as one lattice on the tile engine) with the coupled rule's screen
(tol > 0) and without it (tol = 0): where the screen's cost per iteration sits.
usage: GRAPHLEARN_AMD_RES_STAMPS=1 GRAPHLEARN_AMD_BATCH_DEBUG=1 python tools/batch_stamps.py [n=256] [B=10]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graph_learning_amd import gsp_learn_graph_log_degrees

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
b = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = np.random.default_rng(3)
z = rng.uniform(0.05, 4.0, size=(1, n * (n - 1) // 2)) * rng.uniform(0.9, 1.1, size=(b, n * (n - 1) // 2))
for tol, maxit in ((0.0, 400), (1e-5, 5000), (0.0, 400), (1e-5, 5000)):
    print(f"--- tol {tol} maxit {maxit}", file=sys.stderr, flush=True)
    _, info = gsp_learn_graph_log_degrees(z, 1.0, 1.0, maxit=maxit, tol=tol, return_info=True)
    print(f"    n_iter {info['n_iter']}", file=sys.stderr, flush=True)

"""Dev tool: first call against later calls of the batched solver for several shapes (the first launch of every kernel instantiation
loads its code: ~10 ms each for the tile engine's; a new shape on loaded kernels costs ~1 ms: stand-in session, ring)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graph_learning_amd import gsp_learn_graph_log_degrees
n, b = 256, 4
z = np.random.default_rng(n).uniform(0.2, 3.0, size=(b, n * (n - 1) // 2))
gsp_learn_graph_log_degrees(z, 1, 1, maxit=5)
for k in range(3):
    t0 = time.perf_counter(); out, info = gsp_learn_graph_log_degrees(z, 1, 1, maxit=2000, tol=1e-5, return_info=True); print(k, info['n_iter'], f"{(time.perf_counter()-t0)*1e3:.2f} ms", flush=True)
z2 = np.random.default_rng(1).uniform(0.2, 3.0, size=(6, 300 * 299 // 2))
for k in range(2):
    t0 = time.perf_counter(); out, info = gsp_learn_graph_log_degrees(z2, 1, 1, maxit=2000, tol=1e-5, return_info=True); print('n=300 B=6', k, info['n_iter'], f"{(time.perf_counter()-t0)*1e3:.2f} ms", flush=True)
for (n3, b3) in [(320, 5), (200, 7), (384, 3)]:
    z3 = np.random.default_rng(n3).uniform(0.2, 3.0, size=(b3, n3 * (n3 - 1) // 2))
    for k in range(2):
        t0 = time.perf_counter(); out, info = gsp_learn_graph_log_degrees(z3, 1, 1, maxit=2000, tol=1e-5, return_info=True); print(f'n={n3} B={b3}', k, info['n_iter'], f"{(time.perf_counter()-t0)*1e3:.2f} ms", flush=True)

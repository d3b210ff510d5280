"""Dev tool: bench.py's `window_n256` case (10 windows of 256 samples of one smooth process on 256 nodes) on its own:
end-to-end time of WindowLogModel.fit, and -- with GRAPHLEARN_AMD_BATCH_DEBUG=1 -- where the batched solver spent its
iterations (cruise, chunks, the launch that rebuilds the stopping state).
usage: python tools/window_case.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graph_learning_amd.temporal import WindowLogModel

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = 256
rs = np.random.default_rng(5)
adj = np.triu((rs.random((n, n)) < 6.0 / n) * rs.uniform(0.3, 2.0, (n, n)), 1)
adj = adj + adj.T
chol = np.linalg.cholesky(np.linalg.inv(np.diag(adj.sum(1)) - adj + np.eye(n)))
xw = (chol @ rs.standard_normal((n, 10 * 256))).T
WindowLogModel(window_size=256, avg_degree=6, maxit=50, tol=1e-4).fit(xw)
for r in range(reps):
    t0 = time.perf_counter()
    wm = WindowLogModel(window_size=256, avg_degree=6, maxit=20000, tol=1e-4).fit(xw)
    dt = time.perf_counter() - t0
    print(f"rep {r}: {dt*1e3:.2f} ms end to end, {int(wm.n_iter_)} iterations, converged {bool(wm.converged_)}", flush=True)

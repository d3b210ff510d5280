import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graph_learning_amd import LGMRF
from graph_learning_amd.synth import er_laplacian
from scipy.linalg import solve_triangular
def cov_of(n, d):
    rng = np.random.default_rng(7)
    lap = er_laplacian(n, 6.0 / n, 0.3, 2.0, seed=rng)
    chol = np.linalg.cholesky(lap + np.eye(n))
    x = solve_triangular(chol, rng.standard_normal((n, d)), lower=True, trans="T").T
    return x.T @ x / d
for n, d, cyc, paths in ((256, 64, 8, (1, 2)), (512, 128, 2, (2,)), (1024, 256, 1, (2,)), (1024, 256, 2, (2,)), (2048, 512, 1, (2,))):
    cov = cov_of(n, d)
    for path in paths:
        t0 = time.perf_counter()
        m = LGMRF(max_cycle=cyc).fit(cov, path=path)
        L = m.laplacian_
        fin = np.isfinite(L).all()
        res = np.abs(L @ m.inv_laplacian_ - np.eye(n)).max() if fin else float("nan")
        print(f"n={n} d={d} path={path} cycles={m.n_cycles_} conv={m.converged_} finite={fin} |ThC-I|={res:.2e} "
              f"maxabs={np.nanmax(np.abs(L)):.3e} t={time.perf_counter()-t0:.1f}s", flush=True)

"""Dev tool / BASELINE config 3: LGMRF block-coordinate descent timing on S = X'X/d, X ~ N(0, (L+I)^-1).
usage: python tools/lgmrf_bench.py [n=2048] [d=512] [max_cycle=50] [check_oracle=0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graph_learning_amd import LGMRF, _lib
from graph_learning_amd.synth import er_laplacian
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
max_cycle = int(sys.argv[3]) if len(sys.argv) > 3 else 50
check = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rng = np.random.default_rng(7)
lap = er_laplacian(n, 6.0 / n, 0.3, 2.0, seed=rng)
chol = np.linalg.cholesky(lap + np.eye(n))
from scipy.linalg import solve_triangular
x = solve_triangular(chol, rng.standard_normal((n, d)), lower=True, trans="T").T     # cov (L + I)^-1
cov = x.T @ x / d
print(f"n={n} d={d} mean(S)={cov.mean():.3e}", flush=True)
ctx = _lib.default_context()
t0 = time.perf_counter()
m = LGMRF(max_cycle=max_cycle).fit(cov, ctx=ctx)
dt = time.perf_counter() - t0
nu = m.n_cycles_ * n
print(f"LGMRF fast path: {m.n_cycles_} cycles, converged_={m.converged_}, {dt:.2f} s total, {dt/m.n_cycles_:.3f} s/cycle, "
      f"{nu/dt:.0f} node-updates/s ({1e3*dt/nu:.3f} ms per node-update)", flush=True)
off = m.laplacian_[~np.eye(n, dtype=bool)]
print("finite:", np.isfinite(m.laplacian_).all(), " off-diag <= 0:", (off <= 0).all(),
      " |Theta C - I|max:", np.abs(m.laplacian_ @ m.inv_laplacian_ - np.eye(n)).max(), " nnz edges:", int((off < 0).sum() // 2), flush=True)
if check:
    from oracle import lgmrf as olg
    t0 = time.perf_counter()
    o = olg.LGMRFOracle(max_cycle=max_cycle).fit(cov.copy())
    dto = time.perf_counter() - t0
    err = np.abs(m.laplacian_ - o.laplacian_).max() / np.abs(o.laplacian_).max()
    print(f"oracle: {o.n_cycles_} cycles converged_={o.converged_} {dto:.1f} s ({dto/o.n_cycles_/n*1e3:.2f} ms per node-update); "
          f"rel max err laplacian {err:.2e}; speedup {dto/dt:.1f}x", flush=True)

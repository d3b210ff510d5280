"""Batched (2-D) solver path: GPU vs the oracle on the host, small graphs (WindowLogModel / M-step regime)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np
from graph_learning_amd import gsp_learn_graph_log_degrees
from oracle import logdeg as old

for n, b in [(16, 8), (64, 10), (128, 8), (256, 4)]:
    rng = np.random.default_rng(n)
    e = n * (n - 1) // 2
    z = rng.uniform(0.2, 3.0, size=(b, e))
    gsp_learn_graph_log_degrees(z, 1, 1, maxit=5)
    t0 = time.perf_counter(); out, info = gsp_learn_graph_log_degrees(z, 1, 1, maxit=2000, tol=1e-5, return_info=True); tg = time.perf_counter() - t0
    t0 = time.perf_counter(); gsp_learn_graph_log_degrees(z, 1, 1, maxit=0); tsetup = time.perf_counter() - t0
    t0 = time.perf_counter(); ref, oinfo = old.learn_log_degrees(z, 1, 1, maxit=2000, tol=1e-5, return_info=True); tc = time.perf_counter() - t0
    print(f"n={n} B={b}: GPU {info['n_iter']} its {tg*1e3:.1f} ms, setup {tsetup*1e3:.1f} ms ({(tg-tsetup)/info['n_iter']*1e6:.1f} us/it) | oracle {oinfo['n_iter']} its "
          f"{tc*1e3:.1f} ms ({tc/oinfo['n_iter']*1e6:.1f} us/it) | max diff {np.abs(out-ref).max():.1e}", flush=True)

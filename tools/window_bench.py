"""WindowLogModel (temporal/smooth.py of the reference): GPU end to end vs the oracle's batched solver on the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np
from scipy.spatial.distance import pdist
from graph_learning_amd.temporal.smooth import WindowLogModel
from oracle import logdeg as old

for (t, n, win, deg) in [(200, 30, 10, 3), (640, 64, 32, 5), (1000, 100, 50, 6)]:
    x = np.random.default_rng(t).standard_normal((t, n)).cumsum(axis=1) * 0.1
    WindowLogModel(window_size=win, avg_degree=deg).fit(x)
    t0 = time.perf_counter(); m = WindowLogModel(window_size=win, avg_degree=deg).fit(x); tg = time.perf_counter() - t0
    # the reference's path restated with the oracle: per-window pdist^2, theta, then the 2-D solver
    t0 = time.perf_counter()
    wins = x.reshape(-1, win, n)
    d = np.stack([pdist(w.T) ** 2 for w in wins])
    from scipy.spatial.distance import squareform
    th = np.array([old.theta_for_degree(squareform(dd), deg) for dd in d])[:, None]
    ref = old.learn_log_degrees(d * th, 1, 1)
    tc = time.perf_counter() - t0
    print(f"T={t} n={n} window={win} (B={t//win}): GPU {tg*1e3:.1f} ms ({m.n_iter_} its) | oracle {tc*1e3:.1f} ms | max diff {np.abs(m.weights_-ref).max():.1e}", flush=True)

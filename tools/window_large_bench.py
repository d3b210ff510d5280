"""Dev tool: WindowLogModel end to end with windows of n > 128 nodes (the batched solver's cruise + deciding stretch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graph_learning_amd.temporal.smooth import WindowLogModel
for (t, n, win, deg) in [(2560, 256, 256, 6), (5120, 256, 256, 6), (4000, 400, 500, 6)]:
    rng = np.random.default_rng(t)
    x = rng.standard_normal((t, n)).cumsum(axis=1) * 0.1 + rng.standard_normal((t, n)) * 0.05
    WindowLogModel(window_size=win, avg_degree=deg).fit(x)
    t0 = time.perf_counter(); m = WindowLogModel(window_size=win, avg_degree=deg).fit(x); tg = time.perf_counter() - t0
    print(f"T={t} n={n} window={win} (B={t//win}): GPU {tg*1e3:.1f} ms end to end, {m.n_iter_} iterations -> {tg/m.n_iter_*1e6:.1f} us per iteration all in", flush=True)

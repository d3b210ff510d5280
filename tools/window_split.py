"""Dev tool: where a batched fit from signals spends its time on the host (create / set_signals / init / run / theta / close, ms).
usage: python tools/window_split.py [T n window avg_degree]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graph_learning_amd import _lib
from graph_learning_amd.batched import _run_sessions
ctx = _lib.default_context()
t, n, win, deg = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (2560, 256, 256, 6)
rng = np.random.default_rng(t)
x = rng.standard_normal((t, n)).cumsum(axis=1) * 0.1 + rng.standard_normal((t, n)) * 0.05
blocks = [np.ascontiguousarray(b) for b in x.reshape(-1, win, n)]
for rep in range(3):
    T = {}
    t0 = time.perf_counter(); sessions = [_lib.LogDegreeSession(n, ctx) for _ in blocks]; T['create'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    for b, s in enumerate(sessions): s.set_signals(blocks[b], deg, deferred=True)
    T['set_signals'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    for s in sessions: s.init(1, 1, 0.5, None)
    ctx.sync(); T['init'] = time.perf_counter() - t0
    t0 = time.perf_counter(); out, info = _run_sessions(sessions, ctx, 1000, 1e-5, 1e-3); T['run'] = time.perf_counter() - t0
    t0 = time.perf_counter(); th = [s.theta() for s in sessions]; T['theta'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    for s in sessions: s.close()
    T['close'] = time.perf_counter() - t0
    print(rep, info['n_iter'], {k: round(v * 1e3, 2) for k, v in T.items()}, flush=True)

"""GraphDictLog: single-workgroup engine vs one graph of kernels per step, microseconds per step over a
range of small shapes (the crossover sets GS_MAX_WORK in gl_graphdict_small.hip)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter("ignore")
import numpy as np
from graph_learning_amd import _lib

SHAPES = [(7, 23, 3, 1), (7, 23, 1, 1), (16, 64, 4, 1), (16, 256, 4, 1), (16, 512, 8, 2), (24, 128, 4, 1),
          (32, 64, 4, 1), (32, 128, 8, 1), (32, 200, 4, 1), (40, 100, 6, 1), (48, 64, 4, 1), (20, 1000, 3, 4)]
STEPS = 300
for n, t, k, w in SHAPES:
    x = np.random.default_rng(1).standard_normal((t, n))
    rs = np.random.RandomState(1)
    c0 = np.repeat(rs.uniform(size=(k, t // w)), w, axis=1)
    w0 = rs.uniform(size=(k, n * (n - 1) // 2))
    row = []
    res = {}
    for eng in ("launch", "small"):
        os.environ["GRAPHLEARN_AMD_GD_ENGINE"] = eng
        with _lib.GraphDictSession(n, t, k, w) as s:
            s.set_signals(x); s.set_params(l1_w=0.01, step_c=0.5, step_w=0.5, step_dual=0.5)
            if s.engine != eng:
                row.append(f"{eng}: n/a"); continue
            s.set_state(c0, w0); s.fit(20, 0.0)
            s.set_state(c0, w0)
            t0 = time.perf_counter(); h, ran, conv, fl = s.fit(STEPS, 0.0); dt = time.perf_counter() - t0
            res[eng] = s.get_state()
            row.append(f"{eng}: {dt / STEPS * 1e6:7.1f} us/step")
    dev = ""
    if len(res) == 2:
        dev = " | max dW %.1e" % np.abs(res["launch"][1] - res["small"][1]).max()
    work = n * (n - 1) // 2 * t * (k + 2)
    print(f"n={n:3d} T={t:4d} K={k:2d} W={w} work={work:9d}  " + "  ".join(row) + dev, flush=True)

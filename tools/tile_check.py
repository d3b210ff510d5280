"""Dev tool: the tile engine (gl_tile.hip) against the streaming engine, and its time per iteration.
usage: python tools/tile_check.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graph_learning_amd import _lib

sizes = [int(a) for a in sys.argv[1:]] or [7, 64, 130, 200, 520, 1030, 2048, 2600, 3000, 4096]
ctx = _lib.default_context()
for n in sizes:
    rng = np.random.default_rng(n)
    e = n * (n - 1) // 2
    z = rng.uniform(0.05, 3.0, size=e)
    out = {}
    for eng in ("stream", "resident1", "tile"):
        os.environ["GRAPHLEARN_AMD_ENGINE"] = eng
        try:
            with _lib.LogDegreeSession(n, ctx) as s:
                s.set_distances(z)
                s.init(1.2, 0.8, 0.5, None)
                kern = s.engine_kernel
                n_it, conv, rel = s.iterate(17, 0.0)
                n_it2, conv2, rel = s.iterate(24, 0.0)
                w, d = s.weights_condensed(), s.dual()
                nit_c, conv_c, rel_c = s.iterate(100000, 1e-4)
                t_us = None
                if n >= 64:
                    s.init(1.2, 0.8, 0.5, None)
                    s.iterate_async(200)
                    ctx.sync()
                    ctx.timer_start()
                    s.iterate_async(2000)
                    t_us = ctx.timer_stop() * 1e3 / 2000
                    ctx.timer_start()
                    s.iterate_async(20)
                    t20 = ctx.timer_stop() * 1e3
                out[eng] = (kern, w, d, rel, nit_c, conv_c, t_us, t20 if n >= 64 else None)
        except Exception as exc:
            out[eng] = exc
            print(f"n={n} {eng}: FAILED {exc}", flush=True)
    ref = out["stream"]
    for eng in ("resident1", "tile"):
        o = out[eng]
        if isinstance(o, Exception) or isinstance(ref, Exception):
            continue
        dw, dd = np.abs(o[1] - ref[1]).max(), np.abs(o[2] - ref[2]).max()
        print(f"n={n:5d} {eng:9s} kernel {o[0]:10s} max|dw| {dw:.2e} max|dd| {dd:.2e} rel {o[3][0]:.6e}/{ref[3][0]:.6e} "
              f"converge its {o[4]} vs {ref[4]} ({o[5]}/{ref[5]})  us/it {o[6] if o[6] is None else round(o[6], 2)} "
              f"(stream {ref[6] if ref[6] is None else round(ref[6], 2)})  20-it call {o[7] if o[7] is None else round(o[7], 1)} us",
              flush=True)

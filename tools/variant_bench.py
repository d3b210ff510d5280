"""Dev tool: the tile engine's us per iteration for variant builds of the library (tools/build_variant.sh), one process each.
usage: python tools/variant_bench.py [n] NAME [NAME ...]      (NAME = shipped | a name under graph_learning_amd/csrc/build/var/)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
n = int(args.pop(0)) if args and args[0].isdigit() else 4096
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from graph_learning_amd import _lib
n = %d
ctx = _lib.default_context()
z = np.random.default_rng(n).uniform(0.05, 3.0, size=n * (n - 1) // 2)
with _lib.LogDegreeSession(n, ctx) as s:
    s.set_distances(z); s.init(1, 1, 0.5, None)
    s.iterate_async(200); ctx.sync()
    best = 1e9; best20 = 1e9
    for _ in range(3):
        ctx.timer_start(); s.iterate_async(2000); best = min(best, ctx.timer_stop() * 1e3 / 2000)
        ctx.timer_start(); s.iterate_async(20); best20 = min(best20, ctx.timer_stop() * 1e3)
    s.init(1, 1, 0.5, None)
    import time
    t0 = time.perf_counter(); nit, conv, _ = s.iterate(200000, 1e-5); tc = time.perf_counter() - t0
    print(f"{best:.3f} us/it   20-iteration call {best20:.1f} us   converge {nit} its {tc * 1e3:.1f} ms ({tc / nit * 1e6:.2f} us/it)")
''' % (ROOT, n)
for name in args:
    env = dict(os.environ)
    if name != "shipped":
        env["GRAPHLEARN_AMD_LIB"] = os.path.join(ROOT, "graph_learning_amd", "csrc", "build", "var", name + ".so")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(f"{name:24s} n={n}: {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)

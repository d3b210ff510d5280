"""Randomised sweep of the tile engine's rounds (n > 4096) against the streaming engine: random sizes 4097 .. 11200 (ragged and not),
parameters, warm starts, sparse distance patterns, resumed calls, fixed-iteration and converge mode.
usage: python tools/fuzz_rounds.py [seconds=120] [seed=0].  Exit code 1 on the first mismatch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graph_learning_amd import _lib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
ctx = _lib.default_context()
cases = 0
while time.time() < t_end:
    n = int(rng.integers(4097, 11201)) if rng.uniform() < 0.7 else int(rng.choice([4097, 4160, 4161, 4224, 4608, 5120, 6144, 6145, 8191, 8192, 8193, 8704, 8705, 9000, 10240, 11200]))
    e = n * (n - 1) // 2
    z = rng.uniform(0.05, 4.0, size=e)
    if rng.uniform() < 0.4:
        z *= rng.uniform(size=e) < rng.uniform(0.3, 1.0)
    a, b, step = rng.uniform(0.3, 2.0), rng.uniform(0.3, 2.0), rng.uniform(0.2, 0.9)
    ei = (rng.uniform(0, 0.3, size=e) * (rng.uniform(size=e) < 0.4)) if rng.uniform() < 0.3 else None
    converge = rng.uniform() < 0.3
    k1, k2 = int(rng.integers(16, 40)), (int(rng.integers(16, 60)) if rng.uniform() < 0.5 else 0)
    tol = float(rng.choice([3e-3, 1e-3]))
    desc = f"n={n} a={a:.3f} b={b:.3f} step={step:.2f} warm={ei is not None} converge={converge} k={k1}+{k2}"
    out = {}
    for eng in ("stream", "tile"):
        os.environ["GRAPHLEARN_AMD_ENGINE"] = eng
        with _lib.LogDegreeSession(n, ctx) as s:
            s.set_distances(z)
            s.init(a, b, step, ei)
            if s.engine_kernel != ("k_tile" if eng == "tile" else "k_edge"):
                print(f"NOT APPLICABLE {desc} -> {s.engine_kernel}", flush=True)
                sys.exit(1)
            if converge:
                res = s.iterate(3000, tol)
            else:
                s.iterate(k1, 0.0)
                if k2:
                    s.iterate_async(k2) if eng == "tile" and k2 % 2 else s.iterate(k2, 0.0)
                res = s.iterate(16, 0.0)
            out[eng] = (res, s.weights_condensed(), s.dual())
    (ra, wa, da), (rb, wb, db) = out["stream"], out["tile"]
    dw, dd = np.abs(wa - wb).max(), np.abs(da - db).max()
    bad = ra[0] != rb[0] or ra[1] != rb[1] or dw > 1e-12 or dd > 1e-11 or abs(ra[2][0] - rb[2][0]) > 1e-9 * abs(ra[2][0])
    cases += 1
    print(f"{'MISMATCH' if bad else 'ok'} {desc}: iterations {rb[0]} vs {ra[0]}, max|dw| {dw:.1e} max|dd| {dd:.1e} rel {rb[2][0]:.6e}/{ra[2][0]:.6e}", flush=True)
    if bad:
        sys.exit(1)
print(f"{cases} cases, no mismatch")

"""Randomised parity sweep of the tile engine (gl_tile.hip) against the oracle: random sizes (ragged and not), chunk
lengths, parameters, warm starts, resumed calls, fixed-iteration and converge mode.
usage: python tools/fuzz_tile.py [seconds=120] [seed=0].  Exit code 1 on the first mismatch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np
from graph_learning_amd import _lib
from oracle import logdeg as old

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
ctx = _lib.default_context()
cases = 0
while time.time() < t_end:
    big = rng.uniform() < 0.15
    n = int(rng.integers(1500, 3300)) if big else int(rng.choice([129, 130, 191, 192, 193, 256, 257, 320, 511, 512, 513, 640, 700, 1000, 1024, 1025, 1300]))
    if rng.uniform() < 0.3 and not big:
        n = int(rng.integers(129, 1400))
    e = n * (n - 1) // 2
    z = rng.uniform(0.05, 4.0, size=e) * (rng.uniform(size=e) < rng.uniform(0.3, 1.0))
    a, b, step = rng.uniform(0.3, 2.0), rng.uniform(0.3, 2.0), rng.uniform(0.2, 0.9)
    ei = (rng.uniform(0, 0.3, size=e) * (rng.uniform(size=e) < 0.4)) if rng.uniform() < 0.3 else None
    os.environ["GRAPHLEARN_AMD_ENGINE"] = "tile"
    nb = (n + 63) // 64
    F = nb * (nb - 1) // 2
    us = [u for u in (8, 16, 32) if F * (32 // u) <= 2048]
    u = int(rng.choice(us))
    os.environ["GRAPHLEARN_AMD_TILE_U"] = str(u)
    ops = old.degree_operator(n)
    converge = rng.uniform() < 0.35 and not big
    desc = f"n={n} U={u} a={a:.3f} b={b:.3f} step={step:.2f} warm={ei is not None} converge={converge}"
    with _lib.LogDegreeSession(n, ctx) as s:
        s.set_distances(z)
        s.init(a, b, step, ei)
        if s.engine_kernel != "k_tile":
            print(f"NOT APPLICABLE {desc} -> {s.engine_kernel}", flush=True)
            sys.exit(1)
        if converge:
            tol = float(rng.choice([1e-3, 1e-4]))
            n_it, conv, rel = s.iterate(4000, tol)
            _, info = old.learn_log_degrees(z, a, b, edge_init=None if ei is None else ei.copy(), maxit=4000, tol=tol, step_size=step,
                                            edge_tol=-np.inf, return_info=True, ops=ops)
            if n_it != info["n_iter"]:
                print(f"MISMATCH {desc}: iterations {n_it} vs {info['n_iter']} (tol {tol})", flush=True)
                if abs(n_it - info["n_iter"]) > 1:
                    sys.exit(1)
            total = n_it
        else:
            k1 = int(rng.integers(16, 40))
            k2 = int(rng.integers(16, 60)) if rng.uniform() < 0.5 else 0
            if big:
                k1, k2 = 17, 0
            s.iterate(k1, 0.0)
            if k2:
                if rng.uniform() < 0.5:
                    s.iterate_async(k2)
                else:
                    s.iterate(k2, 0.0)
            total = k1 + k2
            _, info = old.learn_log_degrees(z, a, b, edge_init=None if ei is None else ei.copy(), maxit=total, tol=0.0, step_size=step,
                                            edge_tol=-np.inf, return_info=True, ops=ops)
        w, d = s.weights_condensed(), s.dual()
    if total == info["n_iter"]:
        dw, dd = np.abs(w - info["raw"]).max(), np.abs(d - info["dual"]).max()
        if not (dw < 1e-10 and dd < 1e-10):
            print(f"MISMATCH {desc}: max|dw| {dw:.2e} max|dd| {dd:.2e} after {total} iterations", flush=True)
            sys.exit(1)
    cases += 1
    if cases % 10 == 0:
        print(f"{cases} cases ok (last: {desc})", flush=True)
print(f"fuzz_tile: {cases} cases, no mismatch", flush=True)

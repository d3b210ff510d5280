"""Randomised parity sweep of the batched (2-D) solver above n = 128 -- cruise on the tile engine, deciding stretch on the
batched launches -- against the oracle's 2-D branch: stopping iteration, iterates, reported ratios.
usage: python tools/fuzz_batch.py [seconds=120] [seed=0].  Exit code 1 on the first mismatch."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OMP_NUM_THREADS", "1")
warnings.simplefilter("ignore")
import numpy as np
from graph_learning_amd import gsp_learn_graph_log_degrees
from oracle import logdeg as old

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
cases = cruised = 0
os.environ["GRAPHLEARN_AMD_BATCH_DEBUG"] = "1"
while time.time() < t_end:
    n = int(rng.choice([129, 130, 191, 192, 193, 200, 256, 257, 300, 320, 385, 450, 512, 600]))
    b = int(rng.integers(1, 25 if n <= 260 else (13 if n <= 400 else 7)))
    e = n * (n - 1) // 2
    kind = rng.choice(["uniform", "sparse", "similar"])
    if kind == "uniform":
        z = rng.uniform(0.05, 4.0, size=(b, e))
    elif kind == "sparse":
        z = rng.uniform(0.05, 4.0, size=(b, e)) * (rng.uniform(size=(b, e)) < rng.uniform(0.3, 1.0))
    else:  # one pattern, perturbed per graph (the rule stays undecided by the cheap bounds for long)
        z = rng.uniform(0.05, 4.0, size=(1, e)) * rng.uniform(0.9, 1.1, size=(b, e))
    a, bt = rng.uniform(0.3, 2.0), rng.uniform(0.3, 2.0)
    if rng.uniform() < 0.3:
        kw = dict(maxit=int(rng.integers(64, 200)), tol=0.0, edge_tol=-np.inf)
    else:
        kw = dict(maxit=int(rng.choice([80, 400, 1500])), tol=float(rng.choice([1e-3, 3e-4, 1e-4])), edge_tol=-np.inf)
    desc = f"n={n} B={b} {kind} alpha={a:.3f} beta={bt:.3f} {kw}"
    ref, info = old.learn_log_degrees(z, a, bt, return_info=True, **kw)
    out, got = gsp_learn_graph_log_degrees(z, a, bt, return_info=True, **kw)
    cases += 1
    if got["n_iter"] != info["n_iter"]:
        print(f"MISMATCH {desc}: n_iter {got['n_iter']} vs oracle {info['n_iter']}", flush=True)
        sys.exit(1)
    err = np.abs(out - ref).max()
    if not err <= 1e-10:
        print(f"MISMATCH {desc}: max |w - oracle| = {err:.3e}", flush=True)
        sys.exit(1)
    if info["n_iter"] > 1:
        for k, j in (("rel_primal", 0), ("rel_dual", 1)):
            # (ratios at the rounding floor -- a converged run driven on with tol = 0 -- are differences of nearly equal numbers)
            if info["rel"][j] > 1e-8 and not np.isclose(got[k], info["rel"][j], rtol=1e-8, atol=0):
                print(f"MISMATCH {desc}: {k} {got[k]!r} vs {info['rel'][j]!r}", flush=True)
                sys.exit(1)
    if cases % 10 == 0:
        print(f"[fuzz_batch] {cases} cases ok (last: {desc}, {info['n_iter']} iterations)", flush=True)
print(f"[fuzz_batch] {cases} cases, no mismatch", flush=True)

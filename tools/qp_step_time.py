"""ADMM step time of nonnegative_qp_solver by form: (time at 2100 steps - time at 100 steps) / 2000."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graph_learning_amd import nonnegative_qp_solver
sizes = [int(a) for a in sys.argv[1:]] or [32, 64, 128, 160, 256, 1024]
for n in sizes:
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n, n + 5))
    q, p = a @ a.T / n + 0.05 * np.eye(n), rng.standard_normal(n)
    row = []
    for form in ("auto", "wave", "quad", "one", "tagged", "resident", "launch"):
        if form == "auto":
            os.environ.pop("GRAPHLEARN_AMD_LGMRF_ADMM", None)
        else:
            os.environ["GRAPHLEARN_AMD_LGMRF_ADMM"] = form
        ts = []
        for mi in (100, 2100):
            nonnegative_qp_solver(q, p, step=0.1, tol=-1.0, max_iter=mi)
            t0 = time.perf_counter(); nonnegative_qp_solver(q, p, step=0.1, tol=-1.0, max_iter=mi); ts.append(time.perf_counter() - t0)
        row.append(f"{form} {(ts[1] - ts[0]) / 2000 * 1e6:.2f}")
    print(f"n={n}: us/step  " + "  ".join(row), flush=True)

"""LogModel.fit end to end at the BASELINE size (host buffers in, n x n result out)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np
from graph_learning_amd import LogModel, _lib
from graph_learning_amd.synth import logmodel_problem
n, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 1024)
x, _ = logmodel_problem(n, d, seed=0)
LogModel(avg_degree=10, maxit=10).fit(x)
t0 = time.perf_counter(); m = LogModel(avg_degree=10, maxit=100000).fit(x); t1 = time.perf_counter()
ctx = _lib.default_context()
with _lib.LogDegreeSession(n, ctx) as s:
    ta = time.perf_counter(); th = s.set_signals(x, 10); tb = time.perf_counter()
    s.init(1, 1, 0.5, None); tc = time.perf_counter()
    it, conv, rel = s.iterate(100000, 1e-5); td = time.perf_counter()
    w = s.weights_square(1e-3) if hasattr(s, "weights_square") else None; te = time.perf_counter()
print(f"n={n} d={d}: LogModel.fit {1e3*(t1-t0):.1f} ms ({m.n_iter_} iterations) = upload+distances+theta {1e3*(tb-ta):.1f} + init {1e3*(tc-tb):.1f} "
      f"+ solve {1e3*(td-tc):.1f} + n x n result to host {1e3*(te-td):.1f} ms")

"""BASELINE configs[0]: LogModel.fit on synthetic smooth signals, n = 64, d = 256 -- GPU vs the oracle on the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np
from graph_learning_amd import LogModel
from graph_learning_amd.synth import logmodel_problem
from oracle import logdeg as old

x, _ = logmodel_problem(64, 256, seed=1234)
LogModel(avg_degree=6).fit(x)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); m = LogModel(avg_degree=6).fit(x); ts.append(time.perf_counter() - t0)
t0 = time.perf_counter(); o = old.LogModelOracle(avg_degree=6).fit(x); tc = time.perf_counter() - t0
print(f"C1 n=64 d=256: GPU fit {min(ts)*1e3:.2f} ms ({m.n_iter_} iterations, engine small), oracle fit {tc*1e3:.1f} ms "
      f"(operator build included, as in the reference); max |dW| {np.abs(m.weights_ - o.weights_).max():.1e}")

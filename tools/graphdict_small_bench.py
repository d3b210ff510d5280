"""GraphDictLog at the reference's own test shape (n = 7, T = 23) and a mid size: GPU vs the oracle on the host."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OMP_NUM_THREADS", "1")
warnings.simplefilter("ignore")
import numpy as np
from graph_learning_amd.dictionary import GraphDictLog
from oracle.graphdict import GraphDictLogOracle

for (t, n, k, steps) in [(23, 7, 3, 100), (200, 32, 4, 100), (512, 128, 8, 50)]:
    x = np.random.default_rng(240920).standard_normal((t, n))
    kw = dict(n_atoms=k, l1_w=0.01, random_state=1, max_iter=steps, tol=0.0)
    GraphDictLog(**kw).fit(x)
    t0 = time.perf_counter(); m = GraphDictLog(**kw).fit(x); tg = time.perf_counter() - t0
    t0 = time.perf_counter(); o = GraphDictLogOracle(**kw).fit(x); tc = time.perf_counter() - t0
    print(f"T={t} n={n} K={k}: GPU {tg*1e3:.1f} ms ({tg/steps*1e6:.0f} us/step) | oracle {tc*1e3:.1f} ms ({tc/steps*1e6:.0f} us/step) | "
          f"max dW {np.abs(m.weights_-o.weights_).max():.1e}", flush=True)

# where the time goes at the tiny shape: session set-up vs the steps themselves
from graph_learning_amd import _lib
t, n, k = 23, 7, 3
x = np.random.default_rng(240920).standard_normal((t, n))
rs = np.random.RandomState(1)
c0, w0 = rs.uniform(size=(k, t)), rs.uniform(size=(k, n * (n - 1) // 2))
t0 = time.perf_counter()
s = _lib.GraphDictSession(n, t, k)
s.set_signals(x); s.set_params(l1_w=0.01); s.set_state(c0, w0)
t1 = time.perf_counter()
s.fit(100, 0.0)
t2 = time.perf_counter()
s.fit(100, 0.0)
t3 = time.perf_counter()
s.close()
print(f"tiny shape: session set-up {1e3*(t1-t0):.2f} ms, first 100 steps {1e3*(t2-t1):.2f} ms, next 100 steps {1e3*(t3-t2):.2f} ms ({(t3-t2)*1e4:.1f} us/step)")

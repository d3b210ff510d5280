"""GraphDictLog (BASELINE config 5: K=8, n=1024, T=4096): seconds per step and the two gradient kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graph_learning_amd import _lib

n, t, k = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (1024, 4096, 8)
rng = np.random.default_rng(0)
x = rng.standard_normal((t, n)) * 0.3
e = n * (n - 1) // 2
rs = np.random.RandomState(0)
c0, w0 = rs.uniform(size=(k, t)), rs.uniform(size=(k, e))
bits = int(os.environ.get("GD_BITS", "64"))
with _lib.GraphDictSession(n, t, k) as s:
    s.set_precision(bits)
    s.set_signals(x)
    s.set_params(l1_w=0.01, step_c=0.5, step_w=0.5, step_dual=0.5)
    s.set_state(c0, w0)
    steps = int(os.environ.get("GD_STEPS", "50"))
    s.fit(steps, 0.0)   # (captures the step's graphs for a history of this length: a fit's one-time cost)
    t0 = time.perf_counter(); hist, ran, conv, flags = s.fit(steps, 0.0); dt = (time.perf_counter() - t0) / steps
    gc, gw = s.time_gradients(5)
    pair_samples = t * e
    lane_ops = pair_samples * (2 + 2 * k)
    print(f"fp{bits} n={n} T={t} K={k}: {dt*1e3:.3f} ms/step ({1/dt:.1f} steps/s); grad_c {gc/1e3:.3f} ms, grad_w {gw/1e3:.3f} ms; "
          f"fp64 vector rate grad_c {lane_ops/gc/1e6:.2f} TFLOP/s-equivalent (fma=2: {2*pair_samples*k/gc/1e6:.2f}), "
          f"grad_w {2*pair_samples*k/gw/1e6:.2f} TFLOP/s (fma only)")
    print("history tail", hist[-1])

"""Dev tool: where a CALL of the tile engine spends its time (entry, loads, rendezvous, first iteration, loop, write-back).
usage: python tools/call_stamps.py [n] [steps]   (sets GRAPHLEARN_AMD_RES_STAMPS=1; the driver's call is n = 4096, steps = 20)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GRAPHLEARN_AMD_RES_STAMPS"] = "1"
import numpy as np
from graph_learning_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = _lib.default_context()
z = np.random.default_rng(n).uniform(0.05, 3.0, size=n * (n - 1) // 2)
with _lib.LogDegreeSession(n, ctx) as s:
    s.set_distances(z)
    s.init(1, 1, 0.5, None)
    print(n, s.engine_kernel, flush=True)
    for rep in range(4):
        ctx.sync()
        ctx.timer_start()
        s.iterate_async(steps)
        ms = ctx.timer_stop()
        print(f"-- asynchronous call {rep} of {steps} iterations: {ms * 1e3:.1f} us between HIP events", flush=True)
        s.dual()                 # synchronising call: the library prints the stamps of the launch
    print("-- synchronous call (variant with the norms):", flush=True)
    s.iterate(steps, 0.0)

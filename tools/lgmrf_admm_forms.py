"""LGMRF: time per node-update with each form of the QP's ADMM loop (launch / resident / tagged) over n.
One cycle of the BASELINE config-3 input (S = X'X/d, d = n/4): every QP runs its 100 steps."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
sizes = [int(a) for a in sys.argv[1:]] or [128, 256, 500, 1024, 2048]
for n in sizes:
    row = []
    for form in ("launch", "resident", "tagged"):
        env = dict(os.environ, GRAPHLEARN_AMD_LGMRF_ADMM=form)
        out = subprocess.run([sys.executable, os.path.join(here, "lgmrf_bench.py"), str(n), str(max(n // 4, 32)), "1"],
                             env=env, capture_output=True, text=True, timeout=600)
        ms = [l for l in out.stdout.splitlines() if "ms per node-update" in l]
        row.append(f"{form}: " + (ms[0].split("(")[-1].split(" ms")[0] + " ms" if ms else "failed " + out.stderr[-200:]))
    print(f"n={n:5d}  " + "   ".join(row), flush=True)

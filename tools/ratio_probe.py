"""Dev tool: the two ratios of the stopping rule (smooth_learning.py:156-161) over a run of the oracle on the bench problem's
smaller siblings -- which of them holds the rule up, and for how long.   usage: python tools/ratio_probe.py n d avg_degree tol"""
import sys, numpy as np, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from graph_learning_amd.synth import logmodel_problem
from oracle.logdeg import FBFSolver, signal_distances, theta_for_degree, degree_operator
from scipy.spatial.distance import squareform
n, d, deg = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x, _ = logmodel_problem(n, d, 0, deg)
z = signal_distances(x)
th = theta_for_degree(squareform(z), deg)
s = FBFSolver(z * th, 1, 1)
tol = float(sys.argv[4])
t0 = time.time()
for it in range(20000):
    rp, rd = s.step()
    if it < 5 or it % 50 == 0 or (rp < 1.5*tol and rd < 1.5*tol):
        print(it, f"{rp:.4e} {rd:.4e}", flush=True)
    if rp < tol and rd < tol:
        break
print("stop", s.it, time.time() - t0)

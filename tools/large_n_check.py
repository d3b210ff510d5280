"""Beyond the Infinity Cache: n = 8192 (3 x 268 MB of iterate + distances) -- two iterations against a dense
NumPy restatement of smooth_learning.py:146-164, and the streaming rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from graph_learning_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.default_rng(0)
iu = np.triu_indices(n, 1)
z = rng.uniform(0.2, 3.0, size=iu[0].size)
alpha, beta, step = 1.0, 1.0, 0.5
ctx = _lib.default_context()
with _lib.LogDegreeSession(n, ctx) as s:
    s.set_distances(z)
    s.init(alpha, beta, step, None)
    s.iterate(2, 0.0)
    w_gpu, d_gpu = s.weights_condensed(), s.dual()
    s.iterate_async(50); ctx.sync()
    ctx.timer_start(); s.iterate_async(200); ms = ctx.timer_stop()
    print(f"n={n}: engine {s.engine}, {ms/200*1e3:.1f} us per iteration = {12*n*(n-1)/(ms/200*1e-3)/1e12:.2f} TB/s of algorithmic bytes", flush=True)
# dense restatement (upper triangle kept in a square matrix)
gamma = step / (2 * beta + np.sqrt(2 * (n - 1)))
Z = np.zeros((n, n)); Z[iu] = z; Z += Z.T
W = np.zeros((n, n)); d = W.sum(1)
for _ in range(2):
    y = (1 - 2 * gamma * beta) * W - gamma * (d[:, None] + d[None, :]); np.fill_diagonal(y, 0)
    yb = d + gamma * W.sum(1)
    p = np.maximum(0, y - 2 * gamma * Z); np.fill_diagonal(p, 0)
    pb = (yb - np.sqrt(yb ** 2 + 4 * alpha * gamma)) / 2
    q = (1 - 2 * gamma * beta) * p - gamma * (pb[:, None] + pb[None, :]); np.fill_diagonal(q, 0)
    qb = pb + gamma * p.sum(1)
    W = W - y + q
    d = d - yb + qb
print(f"two iterations vs dense NumPy: max |dw| {np.abs(W[iu] - w_gpu).max():.2e}, max |dd| {np.abs(d - d_gpu).max():.2e}")

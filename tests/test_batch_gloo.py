"""N > 1 path on CPU: world_size-2 gloo run of the sharding / gather logic of
graph_learning_amd.batch, with the per-problem solver replaced by the oracle (the
product solver needs a GPU; what is tested here is the distribution logic)."""

import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from graph_learning_amd.batch import SingleComm, fit_batch, shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, os.environ["GL_ROOT"])
    from graph_learning_amd.batch import GlooComm, fit_batch
    from oracle import logdeg as old

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    comm = GlooComm(rank, world)
    rng = np.random.default_rng(7)
    problems = [rng.uniform(0.1, 2.0, size=15) for _ in range(5)]      # 5 graphs of 6 nodes

    def fit_one(z):
        _, info = old.learn_log_degrees(z, 1, 1, maxit=40, tol=0.0, edge_tol=-np.inf, return_info=True)
        return info["raw"]

    comm.barrier()
    res = fit_batch(problems, fit_one, comm)
    t = comm.max_over_ranks(float(rank + 1))
    assert t == float(world), t
    np.save(os.path.join(os.environ["GL_OUT"], f"res_{rank}.npy"), np.stack(res))
    comm.barrier()
    comm.close()
""")


# A LogModelBatch-shaped run: the driver of graph_learning_amd.batch with its device session replaced by an
# oracle-backed stand-in (same methods).  Problems get DIFFERENT theta (avg_degree rule) and DIFFERENT
# iteration counts (converge mode), so a header that did not travel with its weights would show.
BATCH_WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, os.environ["GL_ROOT"])
    from scipy.spatial.distance import squareform
    from graph_learning_amd.batch import GlooComm, LogModelBatch
    from graph_learning_amd import wire
    from oracle import logdeg as old

    class OracleSession:
        def __init__(self, n, ctx=None):
            self.n, self.n_edges = n, n * (n - 1) // 2
        def __enter__(self): return self
        def __exit__(self, *a): pass
        def set_signals(self, x, k):
            self.z = old.signal_distances(x)
            theta = 1.0 if k is None else old.theta_for_degree(squareform(self.z), k)
            self.z = self.z * theta
            return theta
        def set_distances(self, z): self.z = np.asarray(z, dtype=float)
        def init(self, a, b, step, edge_init): self.par = (a, b, step)
        def iterate(self, maxit, tol):
            a, b, step = self.par
            _, info = old.learn_log_degrees(self.z, a, b, maxit=maxit, tol=tol, step_size=step, edge_tol=-np.inf,
                                            return_info=True)
            self.raw = info["raw"]
            conv = info["rel"][0] < tol and info["rel"][1] < tol
            return info["n_iter"], conv, info["rel"]
        def weights_condensed(self, edge_tol=-np.inf):
            w = self.raw.copy() if hasattr(self, "raw") else np.zeros(self.n_edges)
            w[w < edge_tol] = 0
            return w

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    comm = GlooComm(rank, world)
    rng = np.random.default_rng(11)
    xs = [rng.standard_normal((20 + 3 * k, 8)) * (1 + k) for k in range(5)]   # 5 problems, 2 ranks: one idle slot
    b = LogModelBatch(avg_degree=3, maxit=400, tol=1e-3, edge_tol=1e-3, session_factory=OracleSession).fit(xs, comm)
    recs = np.stack(b.records())
    np.savez(os.path.join(os.environ["GL_OUT"], f"batch_{rank}.npz"), weights=b.weights_, theta=b.theta_,
             n_iter=b.n_iter_, converged=b.converged_, records=recs)
    comm.barrier()
    comm.close()
""")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_assignment():
    assert shard(5, 0, 2) == [0, 2, 4] and shard(5, 1, 2) == [1, 3]
    assert shard(8, 3, 8) == [3] and shard(3, 5, 8) == []
    assert sorted(sum((shard(11, r, 4) for r in range(4)), [])) == list(range(11))


def test_single_comm_fit_batch():
    res = fit_batch([np.arange(3.0), np.arange(3.0) + 1], lambda p: p * 2, SingleComm())
    assert np.array_equal(res[0], [0, 2, 4]) and np.array_equal(res[1], [2, 4, 6])


@pytest.mark.timeout(300)
def test_world_size_2_gloo(tmp_path):
    from oracle import logdeg as old

    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GL_ROOT=ROOT, GL_OUT=str(tmp_path), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=280)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    r0 = np.load(tmp_path / "res_0.npy")
    r1 = np.load(tmp_path / "res_1.npy")
    assert np.array_equal(r0, r1) and r0.shape == (5, 15)
    rng = np.random.default_rng(7)
    for k in range(5):
        z = rng.uniform(0.1, 2.0, size=15)
        _, info = old.learn_log_degrees(z, 1, 1, maxit=40, tol=0.0, edge_tol=-np.inf, return_info=True)
        assert np.array_equal(r0[k], info["raw"])


@pytest.mark.timeout(300)
def test_world_size_2_logmodel_batch_carries_headers(tmp_path):
    """theta_ / n_iter_ / converged_ of the problems the OTHER rank solved arrive through the gathered
    record (wire.py header + weights in one all-gather), identically on both ranks."""
    from scipy.spatial.distance import squareform

    from graph_learning_amd import wire
    from oracle import logdeg as old

    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GL_ROOT=ROOT, GL_OUT=str(tmp_path), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", BATCH_WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=280)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    r0, r1 = np.load(tmp_path / "batch_0.npz"), np.load(tmp_path / "batch_1.npz")
    for key in ("weights", "theta", "n_iter", "converged", "records"):
        assert np.array_equal(r0[key], r1[key]), key
    rng = np.random.default_rng(11)
    xs = [rng.standard_normal((20 + 3 * k, 8)) * (1 + k) for k in range(5)]
    seen_iters = set()
    for k, x in enumerate(xs):   # problems 1 and 3 were solved by rank 1, the others by rank 0
        o = old.LogModelOracle(avg_degree=3, maxit=400, tol=1e-3, edge_tol=1e-3).fit(x)
        assert r0["theta"][k] == o.theta_ and r0["n_iter"][k] == o.info_["n_iter"]
        assert bool(r0["converged"][k]) == (o.info_["n_iter"] < 400)
        assert np.array_equal(squareform(r0["weights"][k]), o.weights_)
        rec = wire.unpack(r0["records"][k])
        assert (rec["problem_id"], rec["n_nodes"], rec["n_iter"], rec["theta"]) == (k, 8, o.info_["n_iter"], o.theta_)
        seen_iters.add(int(r0["n_iter"][k]))
    assert len(set(r0["theta"])) == 5 and len(seen_iters) > 1   # the headers really differ per problem


# ---------------------------------------------------------------------------------- a rank that fails
#  Every collective is entered by all ranks or the process exits non-zero: nobody hangs.
FAIL_WORKER = textwrap.dedent("""
    import os, sys, time
    import numpy as np
    sys.path.insert(0, os.environ["GL_ROOT"])
    from graph_learning_amd.batch import GlooComm, LogModelBatch, BatchError

    MODE = os.environ["GL_FAIL_MODE"]

    class Session:
        def __init__(self, n, ctx=None):
            self.n, self.n_edges = n, n * (n - 1) // 2
            self.raw = np.zeros(self.n_edges)
        def __enter__(self): return self
        def __exit__(self, *a): pass
        def set_signals(self, x, k):
            self.cur = float(np.asarray(x)[0, 0])
            return 1.0
        def set_distances(self, z): self.raw = np.zeros(self.n_edges)
        def init(self, a, b, step, edge_init): pass
        def iterate(self, maxit, tol):
            if self.cur == 1.0:                       # problem 1 = rank 1's first problem
                if MODE == "raise":
                    raise FloatingPointError("the solver of problem 1 blew up")
                os._exit(3)                           # MODE == "die": the rank is gone before the gather
            self.raw = np.full(self.n_edges, self.cur)
            return 7, True, (0.0, 0.0)
        def weights_condensed(self, edge_tol=-np.inf): return self.raw.copy()

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    comm = GlooComm(rank, world, timeout_s=float(os.environ["GL_BOUND"]))
    xs = [np.full((4, 6), float(k)) for k in range(4)]
    t0 = time.time()
    try:
        LogModelBatch(session_factory=Session).fit(xs, comm)
    except BatchError as exc:
        print(f"BatchError after {time.time() - t0:.1f}s: {exc}", flush=True)
        sys.exit(17)
    except Exception as exc:                          # the collective itself gave up (peer gone / bound reached)
        print(f"collective error after {time.time() - t0:.1f}s: {type(exc).__name__}: {exc}", flush=True)
        sys.exit(18)
    sys.exit(0)
""")


def run_fail_workers(tmp_path, mode, bound):
    import time

    port = free_port()
    procs = []
    t0 = time.time()
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GL_ROOT=ROOT, GL_OUT=str(tmp_path), OMP_NUM_THREADS="1", GL_FAIL_MODE=mode,
                   GL_BOUND=str(bound))
        env.pop("GRAPHLEARN_AMD_COMM_TIMEOUT_S", None)
        procs.append(subprocess.Popen([sys.executable, "-c", FAIL_WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    return [p.returncode for p in procs], outs, time.time() - t0


@pytest.mark.timeout(300)
def test_rank_whose_solver_raises_still_enters_the_gather(tmp_path):
    """Rank 1's solver raises on its first problem.  It still takes part in the round's gather with a record flagged
    `failed`, so rank 0 -- which solved its own problem -- gets the SAME BatchError at the same point, at once."""
    codes, outs, _ = run_fail_workers(tmp_path, "raise", bound=120)
    assert codes == [17, 17], outs
    for o in outs:
        assert "rank 1 (problem 1) failed to solve its problem" in o, o
    assert "blew up" in outs[1]


@pytest.mark.timeout(300)
def test_rank_that_dies_before_the_gather_does_not_hang_the_other(tmp_path):
    """Rank 1 is gone before the gather (os._exit in its solver).  Rank 0 must come back with an error within the
    communicator's bound (20 s here) -- not wait for the launcher's kill -- and exit non-zero."""
    codes, outs, took = run_fail_workers(tmp_path, "die", bound=20)
    assert codes[1] == 3, outs
    assert codes[0] in (17, 18), outs            # an error, never 0 and never a kill by this test's timeout
    assert took < 120, (took, outs)

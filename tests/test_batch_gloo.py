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

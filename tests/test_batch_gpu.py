"""RCCL communicator through the C ABI on the one GPU of the test box (world of 1): bootstrap,
barrier, scalar reduction, device-to-device gather; and the LogModelBatch driver."""

import numpy as np
import pytest

from graph_learning_amd import _lib
from graph_learning_amd.batch import LogModelBatch, RcclComm, SingleComm
from graph_learning_amd.synth import logmodel_problem
from oracle import logdeg as old

pytestmark = pytest.mark.gpu


def test_rccl_world_of_one(tmp_path):
    ctx = _lib.Context(0)
    comm = RcclComm(ctx, 0, 1, rendezvous_key=f"test_{tmp_path.name}")
    comm.barrier()
    assert comm.max_over_ranks(3.25) == 3.25
    x, _ = logmodel_problem(40, 80, seed=5)
    with _lib.LogDegreeSession(40, ctx) as s:
        s.set_signals(x, 4)
        s.init(1, 1, 0.5, None)
        s.iterate(60, 0.0)
        g = comm.allgather_condensed(s, 1e-3)
        w = s.weights_condensed(1e-3)
    assert g.shape == (1, 40 * 39 // 2) and np.array_equal(g[0], w)
    comm.close()
    ctx.close()


def test_logmodel_batch_single():
    xs = [logmodel_problem(30, 60, seed=s)[0] for s in range(3)]
    b = LogModelBatch(avg_degree=4, maxit=80, tol=0.0, edge_tol=-np.inf).fit(xs, SingleComm())
    for k, x in enumerate(xs):
        o = old.LogModelOracle(avg_degree=4, maxit=80, tol=0.0, edge_tol=-np.inf).fit(x)
        np.testing.assert_allclose(b.weights_[k], o.info_["raw"], rtol=0, atol=1e-10)
        assert b.theta_[k] == pytest.approx(o.theta_, rel=1e-10) and b.n_iter_[k] == 80

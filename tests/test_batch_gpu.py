"""RCCL communicator through the C ABI on the one GPU of the test box (world of 1): bootstrap,
barrier, scalar reduction, device-to-device gather; and the LogModelBatch driver."""

import numpy as np
import pytest

from graph_learning_amd import _lib, wire
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
        # the gather payload proper: header + weights through one RCCL all-gather
        recs = comm.allgather_records(s, 1e-3, problem_id=17, n_iter=60, theta=0.125, converged=True)
        assert recs.shape == (1, 8 + 40 * 39 // 2)
        rec = wire.unpack(recs[0])
        assert (rec["n_nodes"], rec["n_edges"], rec["problem_id"], rec["n_iter"]) == (40, 780, 17, 60)
        assert rec["theta"] == 0.125 and rec["edge_tol"] == 1e-3 and rec["converged"]
        assert np.array_equal(rec["weights"], w)
        assert np.array_equal(recs[0], wire.pack(w, problem_id=17, n_iter=60, theta=0.125, edge_tol=1e-3, converged=True))
        meta = comm.allgather_arrays(np.array([[1.5, 2.5], [3.5, 4.5]]))
        assert meta.shape == (1, 2, 2) and meta[0, 1, 0] == 3.5
    comm.close()
    ctx.close()


def test_logmodel_batch_single():
    xs = [logmodel_problem(30, 60, seed=s)[0] for s in range(3)]
    b = LogModelBatch(avg_degree=4, maxit=80, tol=0.0, edge_tol=-np.inf).fit(xs, SingleComm())
    for k, x in enumerate(xs):
        o = old.LogModelOracle(avg_degree=4, maxit=80, tol=0.0, edge_tol=-np.inf).fit(x)
        np.testing.assert_allclose(b.weights_[k], o.info_["raw"], rtol=0, atol=1e-10)
        assert b.theta_[k] == pytest.approx(o.theta_, rel=1e-10) and b.n_iter_[k] == 80


def test_logmodel_batch_through_rccl_records(tmp_path):
    """LogModelBatch with the RCCL communicator (world of one on this box): results come out of the gathered
    records, so theta_ / n_iter_ / converged_ are what the header carried."""
    ctx = _lib.Context(0)
    comm = RcclComm(ctx, 0, 1, rendezvous_key=f"batch_{tmp_path.name}")
    xs = [logmodel_problem(24, 50, seed=s)[0] for s in range(2)]
    b = LogModelBatch(avg_degree=3, maxit=5000, tol=1e-4).fit(xs, comm)
    for k, x in enumerate(xs):
        o = old.LogModelOracle(avg_degree=3, maxit=5000, tol=1e-4).fit(x)
        assert b.theta_[k] == pytest.approx(o.theta_, rel=1e-10)
        assert abs(int(b.n_iter_[k]) - o.info_["n_iter"]) <= 1 and b.converged_[k]
        rec = wire.unpack(b.records()[k])
        assert rec["problem_id"] == k and rec["n_iter"] == b.n_iter_[k] and rec["theta"] == b.theta_[k]
    comm.close()
    ctx.close()


def test_rank_that_fails_in_front_of_a_collective_aborts_the_communicator(tmp_path, monkeypatch):
    """a rank that meets a local error on its way INTO a collective (here: the receive buffer's
    allocation, forced to fail) must not leave the others waiting for it.  It aborts the communicator (ncclCommAbort)
    before it returns the error; every later call fails at once with GL_E_COMM instead of entering a collective that the
    peers have left; the solver session and the context stay usable."""
    ctx = _lib.Context(0)
    comm = RcclComm(ctx, 0, 1, rendezvous_key=f"abort_{tmp_path.name}")
    x, _ = logmodel_problem(40, 80, seed=5)
    with _lib.LogDegreeSession(40, ctx) as s:
        s.set_signals(x, 4)
        s.init(1, 1, 0.5, None)
        s.iterate(20, 0.0)
        monkeypatch.setenv("GRAPHLEARN_AMD_TEST_ALLOC_FAIL", "comm")
        with pytest.raises(MemoryError):
            comm.allgather_records(s, 1e-3, problem_id=0, n_iter=20, theta=1.0)
        monkeypatch.delenv("GRAPHLEARN_AMD_TEST_ALLOC_FAIL")
        with pytest.raises(_lib.GraphLearnError, match="aborted"):
            comm.barrier()
        with pytest.raises(_lib.GraphLearnError, match="aborted"):
            comm.allgather_records(s, 1e-3, problem_id=0, n_iter=20, theta=1.0)
        s.iterate(5, 0.0)                                   # the context and the session are untouched
        assert s.total_iterations == 25
    comm.close()
    ctx.close()


def test_logmodel_batch_reports_a_failed_problem_through_the_gather(tmp_path):
    """A problem whose solve raises (here: a signal matrix with the wrong number of nodes, refused by the library) still goes
    through the round's gather as a record flagged `failed`; fit raises BatchError naming it, on every rank alike."""
    from graph_learning_amd.batch import BatchError

    ctx = _lib.Context(0)
    comm = RcclComm(ctx, 0, 1, rendezvous_key=f"fail_{tmp_path.name}")
    xs = [logmodel_problem(24, 50, seed=0)[0], np.ones((50, 23))]      # the second has the wrong number of nodes
    with pytest.raises(BatchError, match=r"rank 0 \(problem 1\) failed"):
        LogModelBatch(avg_degree=3, maxit=50, tol=0.0).fit(xs, comm)
    comm.barrier()                                          # the communicator is alive: every rank entered the gather
    comm.close()
    ctx.close()

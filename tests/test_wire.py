"""Record format of the gather payload (graph_learning_amd/wire.py).  CPU only."""

import numpy as np
import pytest

from graph_learning_amd import wire


def test_round_trip_and_layout(tmp_path):
    rng = np.random.default_rng(0)
    w = rng.uniform(size=15)  # n = 6
    rec = wire.pack(w, problem_id=7, n_iter=123, theta=0.25, edge_tol=1e-3, converged=True)
    assert rec.dtype == np.float64 and rec.size == 8 + 15
    assert rec[:1].tobytes()[:4] == b"GLAM"
    got = wire.unpack(rec)
    assert (got["n_nodes"], got["n_edges"], got["problem_id"], got["n_iter"]) == (6, 15, 7, 123)
    assert got["theta"] == 0.25 and got["edge_tol"] == 1e-3 and got["converged"]
    np.testing.assert_array_equal(got["weights"], w)
    assert wire.unpack(rec.tobytes())["n_iter"] == 123
    # a gathered (world, 8 + E) buffer is a stack of records
    stack = np.stack([wire.pack(w * r, problem_id=r) for r in range(3)])
    assert [wire.unpack(row)["problem_id"] for row in stack] == [0, 1, 2]
    # file of records of different sizes
    path = tmp_path / "graphs.glam"
    wire.save(path, [rec, wire.pack(np.ones(3), problem_id=1)])
    back = wire.load(path)
    assert [b["n_nodes"] for b in back] == [6, 3]
    np.testing.assert_array_equal(back[0]["weights"], w)


def test_rejects_garbage():
    with pytest.raises(ValueError):
        wire.pack(np.ones(4))          # 4 is not n(n-1)/2
    with pytest.raises(ValueError):
        wire.pack(np.ones((2, 3)))
    with pytest.raises(ValueError):
        wire.unpack(np.zeros(20))      # bad magic
    rec = wire.pack(np.ones(6))
    with pytest.raises(ValueError):
        wire.unpack(rec[:10])          # truncated

"""Batch of independent graph problems across the GPUs of one node.

The path shards by problem: rank r solves problems r, r + W,
r + 2W, ... on its own GPU with no collective on the iteration path; at the end
the learned adjacencies are gathered (BASELINE config 4).  Two communicators
with the same small interface:

  * ``RcclComm``  -- RCCL over xGMI through the C ABI (``gl_comm_*``); bootstrap of
    the 128-byte unique id goes through a file in /tmp (single node), keyed by
    MASTER_PORT and the launcher's pid, so no PyTorch is needed;
  * ``GlooComm``  -- ``torch.distributed`` with the gloo backend, host buffers
    only; used by the CPU tests of the sharding / gather logic (and usable where
    no RCCL is available).

Launch one process per GPU (``python -m torch.distributed.run --nproc-per-node N``
or any launcher that sets RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT).
"""

from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np

from . import _lib


def _host_record(session, edge_tol, problem_id, n_iter, theta, converged, failed=False):
    """One gather record (wire.py: 64-byte header + condensed weights) built on the host."""
    from . import wire

    return wire.pack(session.weights_condensed(edge_tol), problem_id=problem_id, n_iter=n_iter, theta=theta,
                     edge_tol=edge_tol, converged=converged, failed=failed)


class BatchError(RuntimeError):
    """A rank of the batch failed.  Every collective is entered by all ranks or by none (a failing rank sends a
    record flagged `failed`, or aborts the communicator when it cannot even do that), and every wait is bounded, so
    the other ranks see this error instead of waiting for the launcher to kill them."""


class SingleComm:
    """World of one: the same interface, nothing to exchange."""

    rank, world = 0, 1

    def barrier(self):
        pass

    def max_over_ranks(self, x: float) -> float:
        return float(x)

    def allgather_arrays(self, arr: np.ndarray) -> np.ndarray:
        return np.asarray(arr)[None]

    def allgather_condensed(self, session, edge_tol=1e-3) -> np.ndarray:
        return session.weights_condensed(edge_tol)[None]

    def allgather_records(self, session, edge_tol, problem_id, n_iter, theta, converged=False, failed=False) -> np.ndarray:
        return _host_record(session, edge_tol, problem_id, n_iter, theta, converged, failed)[None]

    def abort(self):
        pass

    def close(self):
        pass


class RcclComm:
    """RCCL communicator bound to a ``_lib.Context`` (one rank per GPU)."""

    def __init__(self, ctx: _lib.Context, rank: int, world: int, rendezvous_key: str | None = None,
                 timeout_s: float = 120.0, comm_timeout_s: float = 300.0):
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        self._lib = ctx._lib
        uid = (C.c_char * 128)()
        path = self._rendezvous_path(rendezvous_key)
        if self.rank == 0:
            rc = self._lib.gl_comm_unique_id(uid)
            if rc != _lib.GL_OK:
                raise _lib.GraphLearnError("gl_comm_unique_id failed: RCCL is not available")
            tmp = f"{path}.{os.getpid()}.tmp"
            with open(tmp, "wb") as f:
                f.write(bytes(uid))
            os.replace(tmp, path)  # atomic publish
        else:
            deadline = time.monotonic() + timeout_s
            while True:
                try:
                    with open(path, "rb") as f:
                        data = f.read()
                    if len(data) == 128:
                        break
                except FileNotFoundError:
                    pass
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rank {self.rank}: no RCCL unique id at {path} after {timeout_s}s")
                time.sleep(0.01)
            C.memmove(uid, data, 128)
        self._path = path
        h = C.c_void_p()
        ctx.check(self._lib.gl_comm_create(ctx.handle, self.rank, self.world, uid, C.byref(h)), "comm_create")
        self._h = h
        # every wait for a collective is bounded (gl_comm.hip): a rank that never arrives ends the others with an error
        ctx.check(self._lib.gl_comm_set_timeout(h, float(os.environ.get("GRAPHLEARN_AMD_COMM_TIMEOUT_S", comm_timeout_s))), "comm_set_timeout")
        self.barrier()
        if self.rank == 0:
            try:
                os.unlink(path)
            except OSError:
                pass

    @staticmethod
    def _rendezvous_path(key):
        if key is None:
            key = f"{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{os.getppid()}"
        return os.path.join(os.environ.get("TMPDIR", "/tmp"), f"graphlearn_amd_{key}.rcclid")

    def barrier(self):
        self.ctx.check(self._lib.gl_comm_barrier(self._h), "comm_barrier")

    def max_over_ranks(self, x: float) -> float:
        v = C.c_double(float(x))
        self.ctx.check(self._lib.gl_comm_allreduce_max(self._h, C.byref(v)), "comm_allreduce_max")
        return v.value

    def allgather_condensed(self, session, edge_tol=1e-3) -> np.ndarray:
        """(world, E) array: every rank's condensed, thresholded weights, gathered device-to-device."""
        out = np.empty((self.world, session.n_edges))
        self.ctx.check(self._lib.gl_comm_allgather_weights(self._h, session._h, float(edge_tol), _lib.ptr(out)),
                       "comm_allgather_weights")
        return out

    def allgather_arrays(self, arr: np.ndarray) -> np.ndarray:
        """(world, ...) stack of every rank's small float64 array (same shape everywhere)."""
        a = np.ascontiguousarray(arr, dtype=np.float64)
        out = np.empty((self.world,) + a.shape)
        self.ctx.check(self._lib.gl_comm_allgather_f64(self._h, _lib.ptr(a.reshape(-1)), a.size, _lib.ptr(out)),
                       "comm_allgather_f64")
        return out

    def allgather_records(self, session, edge_tol, problem_id, n_iter, theta, converged=False, failed=False) -> np.ndarray:
        """(world, 8 + E): every rank's self-describing record (wire.py) -- header and weights travel
        together through ONE RCCL all-gather, straight from device memory."""
        out = np.empty((self.world, 8 + session.n_edges))
        flags = (1 if converged else 0) | (2 if failed else 0)
        self.ctx.check(self._lib.gl_comm_allgather_records(self._h, session._h, float(edge_tol), int(problem_id),
                                                           int(n_iter), float(theta), flags,
                                                           _lib.ptr(out)), "comm_allgather_records")
        return out

    def abort(self):
        """This rank cannot go on: nobody must wait for it (ncclCommAbort; later calls fail at once)."""
        if getattr(self, "_h", None):
            self._lib.gl_comm_abort(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gl_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GlooComm:
    """torch.distributed (gloo) communicator over host buffers -- CPU tests / fallback."""

    def __init__(self, rank: int, world: int, init_method: str | None = None, timeout_s: float | None = None):
        import datetime

        import torch.distributed as dist

        self._dist = dist
        self.rank, self.world = int(rank), int(world)
        self._own = not dist.is_initialized()
        # the bound of every collective: a rank that never enters one ends the others with an error after this long
        timeout_s = float(os.environ.get("GRAPHLEARN_AMD_COMM_TIMEOUT_S", 300.0 if timeout_s is None else timeout_s))
        self.timeout_s = timeout_s
        if self._own:
            if init_method is None:
                if "TORCHELASTIC_RUN_ID" in os.environ:  # under torch.distributed.run: use the agent's store
                    init_method = "env://"
                else:
                    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
                    port = os.environ.get("MASTER_PORT", "29500")
                    init_method = f"tcp://{addr}:{port}"
            dist.init_process_group("gloo", init_method=init_method, rank=self.rank, world_size=self.world,
                                    timeout=datetime.timedelta(seconds=timeout_s))

    def barrier(self):
        self._dist.barrier()

    def max_over_ranks(self, x: float) -> float:
        import torch

        t = torch.tensor([float(x)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t[0])

    def allgather_arrays(self, arr: np.ndarray) -> np.ndarray:
        import torch

        a = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64))
        outs = [torch.empty_like(a) for _ in range(self.world)]
        self._dist.all_gather(outs, a)
        return np.stack([o.numpy() for o in outs])

    def allgather_condensed(self, session, edge_tol=1e-3) -> np.ndarray:
        return self.allgather_arrays(session.weights_condensed(edge_tol))

    def allgather_records(self, session, edge_tol, problem_id, n_iter, theta, converged=False, failed=False) -> np.ndarray:
        return self.allgather_arrays(_host_record(session, edge_tol, problem_id, n_iter, theta, converged, failed))

    def abort(self):
        """Nothing to tear down on the host side: the peers' collectives end at the process group's timeout."""

    def close(self):
        if self._own and self._dist.is_initialized():
            self._dist.destroy_process_group()


def make_comm(rank: int, world: int, local_rank: int = 0, ctx: _lib.Context | None = None, backend: str | None = None,
              allow_fallback: bool = True, timeout_s: float = 300.0):
    """RCCL by default; ``GRAPHLEARN_AMD_COMM=gloo`` (or backend="gloo") selects the host-staged
    torch.distributed/gloo communicator.  If RCCL cannot be brought up at all (library missing,
    communicator init error on every rank) the gloo communicator is used instead and says so
    (``comm.kind == "gloo-fallback"``) -- a slower gather, never a silent change of the solver; with
    ``allow_fallback=False`` (bench.py) the RCCL error is raised instead.  ``timeout_s`` bounds every wait for a
    collective (a rank that never arrives ends the others with an error, not with the launcher's kill)."""
    if world <= 1:
        return SingleComm()
    backend = backend or os.environ.get("GRAPHLEARN_AMD_COMM", "rccl")
    if backend == "gloo":
        c = GlooComm(rank, world, timeout_s=timeout_s)
        c.kind = "gloo"
        return c
    try:
        c = RcclComm(ctx or _lib.Context(local_rank), rank, world, comm_timeout_s=timeout_s)
        c.kind = "rccl"
        return c
    except (_lib.GraphLearnError, TimeoutError, OSError) as exc:
        if not allow_fallback:
            raise
        print(f"[graph_learning_amd] rank {rank}: RCCL communicator unavailable ({exc}); falling back to gloo",
              flush=True)
        c = GlooComm(rank, world, timeout_s=timeout_s)
        c.kind = "gloo-fallback"
        return c


def shard(n_items: int, rank: int, world: int) -> list[int]:
    """Items of rank r: r, r + W, r + 2W, ... (independent problems, no exchange)."""
    return list(range(rank, n_items, world))


def fit_batch(problems, fit_one, comm=None):
    """Solve ``problems`` (a sequence) sharded over the ranks of ``comm`` and return, on every
    rank, the list of all results in problem order.

    ``fit_one(problem) -> 1-D float64 array`` (e.g. condensed weights); all results must have the
    same length.  Ranks with fewer problems contribute zero-filled rows that are dropped again.
    """
    comm = comm or SingleComm()
    n = len(problems)
    mine = shard(n, comm.rank, comm.world)
    local = [np.asarray(fit_one(problems[i]), dtype=np.float64) for i in mine]
    rounds = (n + comm.world - 1) // comm.world
    width = comm.max_over_ranks(float(local[0].size if local else 0))
    width = int(width)
    out = [None] * n
    for r in range(rounds):
        row = local[r] if r < len(local) else np.zeros(width)
        if row.size != width:
            raise ValueError("fit_batch: results must all have the same length")
        gathered = comm.allgather_arrays(row) if hasattr(comm, "allgather_arrays") else None
        if gathered is None:
            raise TypeError("fit_batch needs a communicator with allgather_arrays (GlooComm / SingleComm); "
                            "use LogModelBatch for device-resident RCCL gathers")
        for src in range(comm.world):
            idx = r * comm.world + src
            if idx < n:
                out[idx] = gathered[src]
    return out


class LogModelBatch:
    """``LogModel`` over a batch of signal matrices, one problem per GPU at a time.

    Every rank passes the same list ``xs`` (each (n_samples, n_nodes), same n_nodes); rank r fits
    xs[r], xs[r + W], ...; after each round every rank's result is all-gathered as ONE self-describing
    record (``wire``: 64-byte header with problem id, theta, iteration count and flags, then the condensed
    thresholded weights) -- with RCCL straight from device memory -- so ``weights_``, ``theta_``, ``n_iter_``
    and ``converged_`` of EVERY problem are right on every rank, whoever solved it.

    ``session_factory(n_nodes, ctx)`` defaults to the device session (``_lib.LogDegreeSession``); the CPU
    tests of the distribution logic pass an oracle-backed stand-in with the same methods.
    """

    def __init__(self, avg_degree=None, *, maxit=1000, tol=1e-5, step_size=0.5, edge_tol=1e-3, session_factory=None):
        self.avg_degree, self.maxit, self.tol, self.step_size, self.edge_tol = avg_degree, maxit, tol, step_size, edge_tol
        self.session_factory = session_factory

    def fit(self, xs, comm=None, ctx: _lib.Context | None = None):
        from . import wire

        comm = comm or SingleComm()
        n_items = len(xs)
        n_nodes = np.asarray(xs[0]).shape[1]
        n_edges = n_nodes * (n_nodes - 1) // 2
        rounds = (n_items + comm.world - 1) // comm.world
        self.weights_ = np.zeros((n_items, n_edges))
        self.theta_ = np.ones(n_items)
        self.n_iter_ = np.zeros(n_items, dtype=np.int64)
        self.converged_ = np.zeros(n_items, dtype=bool)
        if self.session_factory is not None:
            session = self.session_factory(n_nodes, ctx)
        else:
            session = _lib.LogDegreeSession(n_nodes, ctx or getattr(comm, "ctx", None) or _lib.default_context())
        with session as s:
            for r in range(rounds):
                idx = r * comm.world + comm.rank
                theta, n_it, conv, pid, err = 1.0, 0, False, -1, None
                try:
                    if idx < n_items:
                        pid = idx
                        theta = s.set_signals(xs[idx], self.avg_degree)
                        s.init(1, 1, self.step_size, None)
                        n_it, conv, _ = s.iterate(self.maxit, self.tol)
                    else:  # idle rank in the last round: contribute the zero iterate under problem id -1
                        s.set_distances(np.zeros(n_edges))
                        s.init(1, 1, self.step_size, None)
                except Exception as exc:  # noqa: BLE001 -- whatever it is, the other ranks must not wait for this one
                    err = exc
                # Every rank enters the gather, also one whose problem failed: it sends the zero iterate flagged `failed`,
                # so the others get their answer now instead of a timeout.  A rank that cannot even do that aborts the
                # communicator (its peers' wait is bounded and ends with an error).
                try:
                    if err is not None:
                        theta, n_it, conv = 1.0, 0, False
                        s.set_distances(np.zeros(n_edges))
                        s.init(1, 1, self.step_size, None)
                    gathered = comm.allgather_records(s, self.edge_tol, pid, n_it, theta, conv, failed=err is not None)
                except Exception as exc2:  # noqa: BLE001
                    if hasattr(comm, "abort"):
                        comm.abort()
                    raise BatchError(f"rank {comm.rank}: could not take part in the gather of round {r}: {exc2!r}"
                                     + (f" (after {err!r})" if err is not None else "")) from (err or exc2)
                bad = []
                for src in range(comm.world):
                    j = r * comm.world + src
                    rec = wire.unpack(gathered[src])
                    if rec["failed"]:
                        bad.append((src, rec["problem_id"]))
                        continue
                    if j >= n_items:
                        continue
                    if rec["problem_id"] != j or rec["n_nodes"] != n_nodes:
                        raise RuntimeError(f"gathered record of rank {src} is for problem {rec['problem_id']} "
                                           f"({rec['n_nodes']} nodes), expected problem {j} ({n_nodes} nodes)")
                    self.weights_[j] = rec["weights"]
                    self.theta_[j], self.n_iter_[j], self.converged_[j] = rec["theta"], rec["n_iter"], rec["converged"]
                if bad:  # the same exception on every rank, at the same point: nobody goes on into the next round alone
                    what = ", ".join(f"rank {src} (problem {p})" for src, p in bad)
                    raise BatchError(f"round {r}: {what} failed to solve its problem"
                                     + (f": {err!r}" if err is not None else "")) from err
        return self

    def records(self):
        """The gathered graphs as self-describing records (``graph_learning_amd.wire``), one per problem."""
        from . import wire

        return [wire.pack(self.weights_[j], problem_id=j, n_iter=int(self.n_iter_[j]), theta=float(self.theta_[j]),
                          edge_tol=self.edge_tol, converged=bool(self.converged_[j])) for j in range(len(self.weights_))]

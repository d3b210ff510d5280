"""Payload of the multi-GPU gather / on-disk form of one learned graph.

The reference keeps results as ndarray attributes only; the batch-of-graphs path needs a
self-describing record so that a gathered (or saved) block of bytes can be split again:

    offset  size  field
    0       4     magic  b"GLAM"
    4       4     version (uint32, = 1)
    8       8     n_nodes (int64)
    16      8     n_edges (int64) = n_nodes (n_nodes - 1) / 2
    24      8     problem id / seed (int64)
    32      8     iterations run (int64)
    40      8     theta (float64)
    48      8     edge_tol applied to the weights (float64)
    56      8     flags (uint64): bit 0 = stopped on the tolerance, bit 1 = the sender FAILED to solve this problem
                  (the record still travels, so that no rank waits for it and every rank learns of the failure)
    64      8 E   condensed fp64 weights, upper triangle row by row (scipy ``squareform`` order)

Little-endian, fixed 64-byte header, so a (world, 8 + E) float64 all-gather buffer of such records is
also valid: the header occupies exactly 8 doubles.
"""

from __future__ import annotations

import struct

import numpy as np

MAGIC = b"GLAM"
VERSION = 1
HEADER = struct.Struct("<4sIqqqqddQ")
HEADER_DOUBLES = HEADER.size // 8
assert HEADER.size == 64


def pack(weights_condensed, *, problem_id=0, n_iter=0, theta=1.0, edge_tol=0.0, converged=False, failed=False) -> np.ndarray:
    """One record as a float64 vector of length 8 + E (header bytes reinterpreted as 8 doubles)."""
    w = np.ascontiguousarray(weights_condensed, dtype=np.float64)
    if w.ndim != 1:
        raise ValueError("weights must be the condensed (E,) vector")
    e = w.size
    n = int(round((1 + np.sqrt(1 + 8 * e)) / 2))
    if n * (n - 1) // 2 != e:
        raise ValueError(f"{e} is not a valid number of edges")
    head = HEADER.pack(MAGIC, VERSION, n, e, int(problem_id), int(n_iter), float(theta), float(edge_tol),
                       (1 if converged else 0) | (2 if failed else 0))
    out = np.empty(HEADER_DOUBLES + e)
    out[:HEADER_DOUBLES] = np.frombuffer(head, dtype=np.float64)
    out[HEADER_DOUBLES:] = w
    return out


def unpack(record) -> dict:
    """Inverse of ``pack``; accepts bytes or a float64 vector.  ``weights`` is a view when possible."""
    buf = np.frombuffer(record, dtype=np.float64) if isinstance(record, (bytes, bytearray, memoryview)) \
        else np.ascontiguousarray(record, dtype=np.float64)
    if buf.size < HEADER_DOUBLES:
        raise ValueError("record shorter than its header")
    magic, version, n, e, pid, n_iter, theta, edge_tol, flags = HEADER.unpack(buf[:HEADER_DOUBLES].tobytes())
    if magic != MAGIC:
        raise ValueError("not a graph record (bad magic)")
    if version != VERSION:
        raise ValueError(f"unsupported record version {version}")
    if e != n * (n - 1) // 2 or buf.size < HEADER_DOUBLES + e:
        raise ValueError("inconsistent record size")
    return {"n_nodes": n, "n_edges": e, "problem_id": pid, "n_iter": n_iter, "theta": theta, "edge_tol": edge_tol,
            "converged": bool(flags & 1), "failed": bool(flags & 2), "weights": buf[HEADER_DOUBLES:HEADER_DOUBLES + e]}


def save(path, records) -> None:
    """Concatenated records (same or different sizes) in one file."""
    with open(path, "wb") as f:
        for r in records:
            f.write(np.ascontiguousarray(r, dtype=np.float64).tobytes())


def load(path) -> list[dict]:
    data = np.fromfile(path, dtype=np.float64)
    out, pos = [], 0
    while pos < data.size:
        rec = unpack(data[pos:])
        out.append(rec)
        pos += HEADER_DOUBLES + rec["n_edges"]
    return out

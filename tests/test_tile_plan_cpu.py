"""The tile engine's static work decomposition (graph_learning_amd/csrc/gl_tile_plan.h: host code, no HIP) checked on the
CPU for every lattice size: the header is compiled with plain g++ into tests/harness/tile_plan_harness.cpp.

What a wrong plan would do on the GPU is compute a wrong graph (an edge chunk or a diagonal row pair dealt out twice, or
not at all) or read past a descriptor -- so the invariants are checked here, where no GPU is needed:
every chunk of every bulk tile exactly once; every row pair of every diagonal tile exactly once, by a wave that works on that
node block (or has no chunk); the workgroups' block lists consistent with their waves and short enough for the LDS; every
node block owned; and -- the point of round 4's deal -- no YOUNGER wave of a SIMD carries a fringe group while the chip has
older waves or free CUs to take it (at n = 4096 those waves were the last of the chip to finish, every iteration).
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        pytest.skip("no host C++ compiler")
    exe = str(tmp_path_factory.mktemp("plan") / "tile_plan_harness")
    subprocess.run([cxx, "-O1", "-std=c++17", "-I", os.path.join(ROOT, "graph_learning_amd", "csrc"),
                    os.path.join(ROOT, "tests", "harness", "tile_plan_harness.cpp"), "-o", exe], check=True)
    return exe


def plans(exe, ncu):
    out = subprocess.run([exe, str(ncu)], check=True, capture_output=True, text=True).stdout
    rows = {}
    for line in out.splitlines():
        f = line.split()
        rows[int(f[1])] = {f[k]: int(f[k + 1]) for k in range(2, len(f), 2)}
    return rows


@pytest.mark.parametrize("ncu", [256, 128, 85, 32, 12])      # the chip; a half / third / eighth / twentieth of it (batches of graphs)
def test_plan_invariants_for_every_lattice(harness, ncu):
    rows = plans(harness, ncu)
    assert sorted(rows) == list(range(2, 65))
    planned = 0
    for nb, r in rows.items():
        f = nb * (nb - 1) // 2
        if not r["plan"]:
            continue            # (the lattice does not fit that many CUs: the other engines serve it)
        planned += 1
        assert r["chunks_ok"] and r["fringe_ok"] and r["adjacent_ok"] and r["lists_ok"], (nb, r)
        assert r["NWG"] <= ncu and r["owners"] == nb and r["maxblk"] <= 13, (nb, r)
        assert f * (32 // r["U"]) <= 8 * r["NWG"]
    assert planned > 0
    if ncu == 256:
        assert all(rows[nb]["plan"] for nb in range(3, 65))
        # BASELINE configs[1], n = 4096: 2016 chunk waves; the older half takes 2016 of the 2048 fringe row pairs, two added
        # workgroups the other 32, the younger half none
        r = rows[64]
        assert (r["U"], r["NWG"], r["older"], r["younger"], r["idle"]) == (32, 254, 1008, 0, 16), r
        # chunks of 32 row pairs (n > 2880): never a younger wave with a fringe group
        assert all(rows[nb]["younger"] == 0 for nb in range(46, 65)), {nb: rows[nb]["younger"] for nb in range(46, 65)}

"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every
symbol include/*.h declares, the ctypes table covers the header one to one, and the
product fails loudly (no fallback) when no gfx950 device is present."""

import ctypes
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names += re.findall(r"\b(gl_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as entry

    entry.build()
    from graph_learning_amd import _lib

    return _lib


def test_header_declares_functions():
    names = declared_functions()
    assert "gl_logdeg_iterate" in names and "gl_ctx_create" in names and len(names) > 20


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(built.lib_path())
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"


def test_ctypes_table_matches_header(built):
    assert sorted(built.SIGNATURES) == declared_functions()


def test_header_is_plain_c():
    """The boundary is a C ABI: the header must compile as C, with no torch/HIP types."""
    hdr = os.path.join(ROOT, "include", "graphlearn_amd.h")
    src = '#include "graphlearn_amd.h"\nint main(void){return GL_OK;}\n'
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.dirname(hdr), "-x", "c", "-"],
                       input=src.encode(), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    text = open(hdr).read()
    assert "torch" not in text and "hip/" not in text


def test_version_and_strerror(built):
    lib = built.load()
    assert lib.gl_version() == 100
    assert lib.gl_strerror(0) == b"ok"
    assert lib.gl_strerror(-1) == b"invalid argument"


def test_no_device_fails_loudly(built):
    if built.device_count() > 0:
        pytest.skip("a GPU is visible here")
    from graph_learning_amd import GraphLearnError, LogModel
    import numpy as np

    with pytest.raises(GraphLearnError, match="no CPU fallback"):
        LogModel(avg_degree=2).fit(np.random.default_rng(0).standard_normal((10, 5)))


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under graph_learning_amd/ may reference it."""
    pkg = os.path.join(ROOT, "graph_learning_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
    bench = open(os.path.join(ROOT, "bench.py")).read()
    # bench.py may use the oracle only inside its cpu_baseline leg
    for m in re.finditer(r"^\s*(from|import)\s+oracle\b.*$", bench, flags=re.M):
        before = bench[: m.start()]
        assert before.rfind("def cpu_baseline") > before.rfind("\ndef main"), "oracle import outside cpu_baseline"

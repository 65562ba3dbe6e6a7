"""The C-ABI libraries load on a CPU-only box and export every symbol include/tsgo.h declares."""
import ctypes as C
import os
import re

import pytest

from toyslam_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="tsgo.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tsgo_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_what_the_bindings_expect():
    names = declared_symbols()
    for s in _lib.HOST_SYMBOLS + _lib.DEVICE_SYMBOLS:
        assert s in names, s
    assert set(names) == set(_lib.HOST_SYMBOLS + _lib.DEVICE_SYMBOLS)


def test_host_library_exports_host_symbols():
    lib = C.CDLL(build.build_host())
    for s in _lib.HOST_SYMBOLS:
        assert hasattr(lib, s), s


def test_hip_library_exports_every_symbol():
    if not os.path.exists(build.HIP_SO):
        pytest.skip("libtsgo_hip.so not built yet (run __graft_entry__.build())")
    lib = C.CDLL(build.HIP_SO)
    for s in declared_symbols():
        assert hasattr(lib, s), s


def test_testing_library_exports_the_product_abi_plus_the_testing_entry_points():
    if not os.path.exists(build.HIP_TESTING_SO):
        pytest.skip("libtsgo_hip_testing.so not built yet (run __graft_entry__.build())")
    lib = C.CDLL(build.HIP_TESTING_SO)
    for s in declared_symbols() + declared_symbols("tsgo_testing.h"):
        assert hasattr(lib, s), s
    assert sorted(declared_symbols("tsgo_testing.h")) == sorted(_lib.TESTING_SYMBOLS)


def test_shipped_binaries_contain_no_test_hook_or_research_variable():
    """`strings graph_optimizer | grep TSGO_INJECT` is empty — and so is every other name of host/knobs.h's
    research set, in the server and in both product libraries; the testing twin of the library has them."""
    names = [b"TSGO_INJECT_AMG_FAILURE", b"TSGO_FORCE_HOST_SLOW", b"TSGO_FORCE_PACED", b"TSGO_SYM_DECLINE", b"TSGO_HOST_PRODUCTS", b"TSGO_HIER_MAX_AGE",
             b"TSGO_HIER_SLACK", b"TSGO_AGGC", b"TSGO_AGG_LIST", b"TSGO_AGG_MODE", b"TSGO_SWEEPS_LIST", b"TSGO_PACE_LEAD", b"TSGO_SORT_WINDOW_POSE", b"tsgo_local_group_create"]
    for path in (build.HIP_SO, build.HOST_SO, build.SERVER):
        if not os.path.exists(path):
            pytest.skip("%s not built yet" % path)
        blob = open(path, "rb").read()
        for n in names:
            assert n not in blob, (path, n)
    if os.path.exists(build.HIP_TESTING_SO):
        blob = open(build.HIP_TESTING_SO, "rb").read()
        for n in names:
            assert n in blob, n


def test_device_entry_points_fail_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available() or not os.path.exists(build.HIP_SO):
        pytest.skip("only meaningful on a CPU-only box with the library built")
    lib = _lib.hip_lib()
    h = C.c_void_p()
    rc = lib.tsgo_create(None, C.byref(h))
    assert rc != 0 and not h.value
    assert b"no CPU fallback" in lib.tsgo_last_error()


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it —
    not the package and not tools/ (scripts that need the oracle live under tests/)."""
    bad = []
    for root, _d, files in list(os.walk(os.path.join(ROOT, "toyslam_amd"))) + list(os.walk(os.path.join(ROOT, "tools"))):
        for f in files:
            if f.endswith((".py", ".h", ".cpp", ".hip")):
                s = open(os.path.join(root, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", s, flags=re.M) or "oracle/" in s and f.endswith((".cpp", ".hip", ".h")) and "#include" in s and re.search(r'#include\s+"[^"]*oracle', s):
                    bad.append(f)
    assert not bad, bad

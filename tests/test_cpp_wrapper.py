"""include/tsgo.hpp — the C++ host side above the C ABI, shaped like the reference's GraphCpu / IOptimizer
(remote/graph/GraphCpu.h:15-53, remote/optimizer/IOptimizer.h:10-26).  tests/cpp/wrapper_demo.cpp rebuilds the
golden config-1 graph through AddVertex / AddEdge / FixVertex and optimises it with OptimizerHip."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "wrapper_demo")
    lib_dir = os.path.join(ROOT, "toyslam_amd")
    if not os.path.exists(os.path.join(lib_dir, "libtsgo_hip.so")):
        pytest.fail("toyslam_amd/libtsgo_hip.so is not built (run __graft_entry__.build())")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "wrapper_demo.cpp"), "-o", exe, "-L" + lib_dir, "-ltsgo_hip",
                           "-Wl,-rpath," + lib_dir])
    return exe


def test_wrapper_compiles_links_and_fails_loudly_without_a_device(tmp_path):
    import torch
    exe = _build(tmp_path)
    if torch.cuda.is_available() or torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible: the run itself is covered by the gpu-marked test")
    p = subprocess.run([exe, os.path.join(util.GOLDEN, "c1_request.bin"), "5", str(tmp_path / "o.txt")], capture_output=True, text=True)
    assert p.returncode != 0
    assert "no CPU fallback" in p.stderr


@pytest.mark.gpu
def test_wrapper_runs_config_1_like_the_reference_pipeline(tmp_path):
    exe = _build(tmp_path)
    out = tmp_path / "o.txt"
    p = subprocess.run([exe, os.path.join(util.GOLDEN, "c1_request.bin"), "50", str(out)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert "Plateau: NO MORE OPT" in p.stdout and "Summary() error = " in p.stdout          # OptimizerCpu.h:169,182
    g = util.c1_arrays()
    ref = oracle.optimize(util.to_oracle(g), 50, mode="cpp", solver="chol")
    pos = {}
    chi2 = []
    for line in out.read_text().splitlines():
        t = line.split()
        if t[0] == "v":
            pos[int(t[1])] = [float(t[3]), float(t[4]), float(t[5])]
        elif t[0] == "chi2":
            chi2.append(float(t[2]))
    v = np.array([pos[int(i)] for i in g.v_id])
    assert len(chi2) == ref["iters"]
    np.testing.assert_allclose(chi2, ref["chi2"], rtol=1e-9)
    assert util.max_vertex_diff(v, ref["v_pos"], g.v_type) < 1e-8

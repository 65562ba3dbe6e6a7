"""The big BASELINE configurations on the device (`-m gpu`), checked by something that shares no code with the
product: tests/independent.py (per-edge Jacobians of the dense restatement + numpy), plus the CPU twin where a second
opinion on the whole step is useful.  Config 2 runs its own 50 iterations; configs 3 and 5 are at full size."""
import numpy as np
import pytest

from oracle import oracle
from tests import independent, util
from toyslam_amd import synth
from toyslam_amd.optimizer import HipOptimizer

pytestmark = pytest.mark.gpu


def _check_against_numpy(name, pcg_iter_bound):
    g = synth.make_config(name)
    lin = independent.Linearisation(g)
    o = HipOptimizer(pcg_rel_tol=1e-12)
    try:
        o.set_graph(g)
        diag, grad, chi2 = o.linearize()
        step = o.solve_step()
    finally:
        o.close()
    # OptimizerCpu.h:88-119,132-138: chi^2, b and the diagonal of H — f64, summation order differs only
    assert abs(chi2 - lin.chi2) <= 1e-11 * lin.chi2
    gref = lin.gradient(); dref = lin.diag_blocks()
    np.testing.assert_allclose(grad, gref, rtol=0, atol=1e-9 * np.abs(gref).max())
    np.testing.assert_allclose(diag, dref, rtol=0, atol=1e-9 * np.abs(dref).max())
    # SolverEigen.h:20: H delta = b.  The CPU twin's own step leaves 7e-13 here at 100k poses.
    res = lin.residual_of(step["delta"])
    assert res < 1e-9, res
    assert abs(step["chi2"] - lin.chi2) <= 1e-11 * lin.chi2
    assert 0 < step["cg_iters"] < pcg_iter_bound
    return g, lin, step


def test_c2_linearisation_and_solve_against_the_numpy_checker():
    _check_against_numpy("c2_10k", 120)


def test_c3_linearisation_and_solve_against_the_numpy_checker():
    _check_against_numpy("c3_100k", 150)


def test_c5_1m_poses_linearisation_and_solve_against_the_numpy_checker_and_the_twin():
    """BASELINE config 5 (1M poses / 9.1M edges incl. 121k loop closures) on ONE device: the linearisation and one exact
    step against the numpy checker, the same step against the CPU twin."""
    g, lin, step = _check_against_numpy("c5_1m", 200)
    ref = oracle.sparse_step(util.to_oracle(g), 1e-12, precond="amg")
    assert abs(step["chi2"] - ref["chi2"]) <= 1e-11 * ref["chi2"]
    assert np.abs(step["delta"] - ref["delta"]).max() <= 1e-7 * np.abs(ref["delta"]).max()      # north_star: 1e-6


def test_c5_full_size_properties():
    """Size-independent properties at 1M poses: chi^2 falls at the damped-GN rate, the estimate moves towards the
    truth, the gauge vertex stays put, every solve stays in tens of multigrid iterations with no fallback."""
    g, truth = synth.make_config("c5_1m", with_truth=True)
    o = HipOptimizer(pcg_rel_tol=1e-10)
    try:
        o.set_graph(g)
        r = o.optimize(4)
        v = o.vertices()
    finally:
        o.close()
    assert r["iters"] == 4 and np.all(np.diff(r["chi2"]) < 0)
    assert r["fallbacks"] == 0 and r["cg_iters"].max() < 200, r["cg_iters"]
    ratio = r["chi2"][1:] / r["chi2"][:-1]
    assert np.all(ratio < 0.9) and np.all(ratio > 0.4), ratio
    e0 = np.linalg.norm(g.v_pos[:, :2] - truth[:, :2], axis=1).mean()
    e1 = np.linalg.norm(v[:, :2] - truth[:, :2], axis=1).mean()
    assert e1 < e0
    assert np.abs(v[0] - g.v_pos[0]).max() < 1e-3
    # the state the device holds after 4 iterations is a fixed point of its own read-out: chi^2 of the returned
    # vertices, recomputed by the numpy checker, continues the device's trajectory
    g4 = g.copy(); g4.v_pos[:] = v
    chi_next = independent.Linearisation(g4).chi2
    assert chi_next < r["chi2"][-1] and chi_next > 0.4 * r["chi2"][-1]


def test_c2_fifty_iterations_like_the_config_says():
    """BASELINE config 2: '50 GN iters vs cpu/eigen chi^2'.  The dense cpu/eigen algorithm cannot hold n = 70k (39 GB);
    the twin runs the same rules.  Whole trajectory, stop rule, final vertices; then the final state is re-linearised
    by the numpy checker, which knows nothing of either."""
    g = synth.make_config("c2_10k")
    ref = oracle.sparse_optimize(util.to_oracle(g), 50, pcg_tol=1e-12, precond="jacobi")
    o = HipOptimizer(pcg_rel_tol=1e-12)
    try:
        o.set_graph(g)
        r = o.optimize(50)
        v = o.vertices()
        _, _, chi_dev = o.linearize()
    finally:
        o.close()
    assert (r["iters"], r["stop"]) == (ref["iters"], ref["stop"])
    np.testing.assert_allclose(r["chi2"], ref["chi2"], rtol=1e-8)                  # north_star: 1e-6
    assert util.max_vertex_diff(v, ref["v_pos"], g.v_type) < 1e-7                  # north_star: 1e-6
    gf = g.copy(); gf.v_pos[:] = v
    chi_np = independent.Linearisation(gf).chi2
    assert abs(chi_dev - chi_np) <= 1e-9 * chi_np


def test_c3_twelve_iterations_final_chi2_and_poses():
    """BASELINE config 3 at full size, the bar north_star states: final chi^2 (relative) and pose deltas (absolute) within 1e-6 —
    twelve Gauss-Newton iterations at the bench's tolerance (1e-10) against the tightly converged twin (1e-12), and the device's
    final state re-linearised by the numpy checker, which shares nothing with either."""
    g = synth.make_config("c3_100k")
    ref = oracle.sparse_optimize(util.to_oracle(g), 12, pcg_tol=1e-12, precond="amg")
    o = HipOptimizer(pcg_rel_tol=1e-10)
    try:
        o.set_graph(g)
        r = o.optimize(12)
        v = o.vertices()
        _, _, chi_dev = o.linearize()
    finally:
        o.close()
    assert (r["iters"], r["stop"]) == (ref["iters"], ref["stop"]) and r["fallbacks"] == 0
    np.testing.assert_allclose(r["chi2"], ref["chi2"], rtol=1e-6)
    assert abs(r["chi2"][-1] - ref["chi2"][-1]) <= 1e-8 * ref["chi2"][-1]           # measured: ~1e-11
    assert util.max_vertex_diff(v, ref["v_pos"], g.v_type) < 1e-6                  # measured: ~1e-9
    gf = g.copy(); gf.v_pos[:] = v
    chi_np = independent.Linearisation(gf).chi2
    assert abs(chi_dev - chi_np) <= 1e-9 * chi_np

"""Analytic SE(2) Jacobians for ODOM edges — an EXTENSION (the reference's README lists it as further
development and its Jacobians are the constants -I / +I in every implementation, EdgeSe2.h:35-37).  Nothing in the reference
pins it, so it is pinned here by what it must be: the derivative of the reference's own residual under the reference's own
vertex update (finite differences), and then dense restatement -> twin -> device as for everything else.  Default off."""
import numpy as np
import pytest

from oracle import oracle
from tests import edge_cases, util
from toyslam_amd import synth


@pytest.fixture
def analytic():
    oracle.set_odom_jacobian("analytic")
    yield
    oracle.set_odom_jacobian("constant")


def test_oracle_jacobians_are_the_derivatives_of_the_reference_residual(analytic):
    g = synth.make(30, 4, loop_closures=3, seed=2)
    od = np.where(g.e_type == 0)[0]
    g.e_meas[od[1]] = g.e_meas[od[1]] * np.array([1.1, 1, 1, 1, 0.9, 1, 1, 1, 1])      # a measurement that is not a rigid transform
    _, A, B = oracle.edge_eval(util.to_oracle(g))
    idx = {int(v): k for k, v in enumerate(g.v_id)}
    h, worst = 1e-6, 0.0
    for ei in od[:10]:
        for vi, J in ((idx[int(g.e_ids[ei, 0])], A[ei].reshape(3, 3)), (idx[int(g.e_ids[ei, 1])], B[ei].reshape(3, 3))):
            for c in range(3):                                   # x, y (world frame), theta: all additive (VertexSe2.h:16-27)
                gp, gm = g.copy(), g.copy()
                gp.v_pos[vi, c] += h; gm.v_pos[vi, c] -= h
                d = oracle.edge_eval(util.to_oracle(gp))[0][ei] - oracle.edge_eval(util.to_oracle(gm))[0][ei]
                d[2] = (d[2] + np.pi) % (2 * np.pi) - np.pi
                worst = max(worst, np.abs(d / (2 * h) - J[:, c]).max())
    assert worst < 1e-7, worst
    # LM edges are untouched by the switch
    oracle.set_odom_jacobian("constant")
    _, A0, B0 = oracle.edge_eval(util.to_oracle(g))
    lm = g.e_type == 1
    np.testing.assert_array_equal(A[lm], A0[lm]); np.testing.assert_array_equal(B[lm], B0[lm])
    assert np.abs(A0[~lm] - np.tile(-np.eye(3).reshape(-1), (int((~lm).sum()), 1))).max() == 0


@pytest.mark.parametrize("precond", ["amg", "jacobi"])
def test_twin_with_analytic_jacobians_matches_the_dense_restatement(analytic, precond):
    for g, n in ((util.c1_arrays(), 8), (edge_cases.pose_graph_without_landmarks(), 12), (synth.make(200, 6, loop_closures=8, seed=5), 6)):
        rd = oracle.optimize(util.to_oracle(g), n, mode="cpp", solver="chol")
        rt = oracle.sparse_optimize(util.to_oracle(g), n, pcg_tol=1e-13, precond=precond)
        assert (rt["iters"], rt["stop"]) == (rd["iters"], rd["stop"])
        np.testing.assert_allclose(rt["chi2"], rd["chi2"], rtol=1e-10)
        assert util.max_vertex_diff(rt["v_pos"], rd["v_pos"], g.v_type) < 1e-9


def test_a_pose_graph_with_loop_closures_diverges_under_the_constants_and_converges_under_the_analytic_jacobians():
    g = edge_cases.pose_graph_without_landmarks()
    ref = oracle.optimize(util.to_oracle(g), 30, mode="cpp", solver="chol")
    assert ref["stop"] == "worse" and ref["chi2"][-1] > ref["chi2"][0]            # the reference's own behaviour: "Error is getting worse"
    oracle.set_odom_jacobian("analytic")
    try:
        r = oracle.optimize(util.to_oracle(g), 30, mode="cpp", solver="chol")
    finally:
        oracle.set_odom_jacobian("constant")
    assert np.all(np.diff(r["chi2"]) < 0) and r["chi2"][-1] < 1e-3 * r["chi2"][0]

"""Virtual landmark measurements (edge type 2, include/tsgo.h) on the device (`-m gpu`): the general pose-pose slots of tsgo_math.h
(k_lin_pose<.., 1>, k_schur_pose<.., 1>, k_schur_blocks) against the dense restatement, the twin and the numpy checker."""
import numpy as np
import pytest

from oracle import oracle
from tests import independent, util
from tests.test_gpu_sharded_inprocess import _merge_landmarks, _run_sharded
from toyslam_amd import synth
from toyslam_amd.optimizer import HipOptimizer

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precond", ["amg", "jacobi"])
@pytest.mark.parametrize("jac", ["constant", "analytic"])
def test_device_with_virtual_landmarks_matches_the_dense_restatement(precond, jac):
    oracle.set_odom_jacobian(jac)
    try:
        for g, n, solver in ((util.with_virtual_landmarks(util.c1_arrays(), 0.4, seed=3), 8, "chol"),
                             (util.with_virtual_landmarks(synth.make(150, 6, loop_closures=4, seed=5), 0.7, seed=4, keep_lm=False), 5, "qr")):
            ref = oracle.optimize(util.to_oracle(g), n, mode="cpp", solver=solver)
            H, b, err, off = oracle.linearize(util.to_oracle(g))
            o = HipOptimizer(pcg_rel_tol=1e-12, preconditioner=precond, odom_jacobian=jac)
            try:
                o.set_graph(g)
                diag, grad, chi2 = o.linearize()
                assert abs(chi2 - err) <= 1e-11 * err
                for v in range(len(g.v_id)):
                    d = 3 if g.v_type[v] == 0 else 2
                    np.testing.assert_allclose(grad[v, :d], b[off[v]:off[v] + d], rtol=0, atol=1e-9 * np.abs(b).max())
                    np.testing.assert_allclose(diag[v].reshape(3, 3)[:d, :d], H[off[v]:off[v] + d, off[v]:off[v] + d], rtol=0, atol=1e-9 * np.abs(H).max())
                r = o.optimize(n)
                v = o.vertices()
            finally:
                o.close()
            assert (r["iters"], r["stop"]) == (ref["iters"], ref["stop"])
            np.testing.assert_allclose(r["chi2"], ref["chi2"], rtol=1e-9)
            assert util.max_vertex_diff(v, ref["v_pos"], g.v_type) < 1e-8
    finally:
        oracle.set_odom_jacobian("constant")


def test_ten_thousand_poses_with_virtual_landmarks_against_the_numpy_checker_and_the_twin():
    g = util.with_virtual_landmarks(synth.make_config("c2_10k"), 0.25, seed=7)
    assert (g.e_type == 2).sum() > 3000
    lin = independent.Linearisation(g)
    ref = oracle.sparse_optimize(util.to_oracle(g), 6, pcg_tol=1e-12, precond="amg")
    o = HipOptimizer(pcg_rel_tol=1e-12)
    try:
        o.set_graph(g)
        diag, grad, chi2 = o.linearize()
        step = o.solve_step()
        r = o.optimize(6)
        v = o.vertices()
    finally:
        o.close()
    assert abs(chi2 - lin.chi2) <= 1e-11 * lin.chi2
    gref = lin.gradient(); dref = lin.diag_blocks()
    np.testing.assert_allclose(grad, gref, rtol=0, atol=1e-9 * np.abs(gref).max())
    np.testing.assert_allclose(diag, dref, rtol=0, atol=1e-9 * np.abs(dref).max())
    assert lin.residual_of(step["delta"]) < 1e-9
    assert r["fallbacks"] == 0 and r["cg_iters"].max() < 120, r["cg_iters"]
    np.testing.assert_allclose(r["chi2"], ref["chi2"], rtol=1e-9)
    assert util.max_vertex_diff(v, ref["v_pos"], g.v_type) < 1e-8


def test_virtual_landmarks_through_the_sharded_device_path_and_in_f32():
    g = util.with_virtual_landmarks(synth.make(6000, 10, loop_closures=40, seed=13), 0.3, seed=5)
    single = HipOptimizer(pcg_rel_tol=1e-12)
    try:
        single.set_graph(g); rs = single.optimize(4); vs = single.vertices()
    finally:
        single.close()
    outs = _run_sharded(g, 2, 4, pcg_rel_tol=1e-12)
    v = _merge_landmarks(g, outs)
    for r, _ in outs:
        np.testing.assert_allclose(r["chi2"], rs["chi2"], rtol=1e-10)
        assert r["stop"] == rs["stop"]
    np.testing.assert_array_equal(outs[0][0]["chi2"], outs[1][0]["chi2"])
    assert util.max_vertex_diff(v, vs, g.v_type) < 1e-9
    o = HipOptimizer(precision=32, pcg_rel_tol=1e-5)
    try:
        o.set_graph(g); r32 = o.optimize(4)
    finally:
        o.close()
    np.testing.assert_allclose(r32["chi2"], rs["chi2"], rtol=2e-3)


def test_python_graph_optimizer_takes_the_virtual_landmark_edge_class():
    """toyslam_amd.optimizer.GraphOptimizer(graph).optimize(n) (the call shape of python/optimizer/graph_optimizer.py:11-20) on an
    OptGraph that holds an EdgeVirtualLandmark2d, against the dense restatement of the same flattened graph."""
    from toyslam_amd.graph import EdgeLandmark2d, EdgeOdometry2d, EdgeVirtualLandmark2d, GraphArrays, OptGraph, Vertex2d, VertexPose2d
    from toyslam_amd.optimizer import GraphOptimizer
    rng = np.random.default_rng(3)

    def T(x, y, t):
        c, s = np.cos(t), np.sin(t)
        return np.array([[c, -s, x], [s, c, y], [0, 0, 1.0]])
    og = OptGraph()
    truth = [(0.0, 0.0, 0.0), (1.0, 0.1, 0.2), (2.0, 0.5, 0.5), (2.8, 1.2, 0.9)]
    lms = [(1.5, 2.0), (3.0, -0.5), (2.5, 2.5)]
    for i, (x, y, t) in enumerate(truth):
        og.add_vertex(i, VertexPose2d(T(x + 0.1 * rng.normal(), y + 0.1 * rng.normal(), t + 0.05 * rng.normal())), fixed=(i == 0))
    for j, (lx, ly) in enumerate(lms[:2]):
        og.add_vertex(10 + j, Vertex2d([lx + 0.2 * rng.normal(), ly + 0.2 * rng.normal()]))

    def obs(i, l):
        x, y, t = truth[i]; dx, dy = l[0] - x, l[1] - y
        return np.array([np.hypot(dx, dy), np.arctan2(dy, dx) - t])
    for i in range(3):
        og.add_edge(EdgeOdometry2d(i, i + 1, np.linalg.inv(T(*truth[i])) @ T(*truth[i + 1]), np.diag([4.0, 4.0, 65.0])))
    for i in range(4):
        for j in range(2):
            og.add_edge(EdgeLandmark2d(i, 10 + j, obs(i, lms[j]), np.diag([44.0, 44.0])))
    for a, b in ((0, 2), (1, 3), (0, 3)):                      # the third landmark has no vertex: three pose pairs that both saw it
        og.add_edge(EdgeVirtualLandmark2d(a, b, obs(a, lms[2]), obs(b, lms[2]), np.diag([44.0, 44.0])))
    arr = GraphArrays.from_optgraph(og)
    ref = oracle.optimize(util.to_oracle(arr), 25, mode="cpp", solver="chol")
    r = GraphOptimizer(og, pcg_rel_tol=1e-12).optimize(25)
    assert (r["iters"], r["stop"]) == (ref["iters"], ref["stop"])
    np.testing.assert_allclose(r["chi2"], ref["chi2"], rtol=1e-9, atol=1e-12)
    after = GraphArrays.from_optgraph(og)
    assert util.max_vertex_diff(after.v_pos, ref["v_pos"], arr.v_type) < 1e-8
    assert r["chi2"][-1] < 1e-3 * r["chi2"][0]                 # exact measurements: the estimate goes to the truth

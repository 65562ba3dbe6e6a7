"""Virtual landmark measurements — edge type 2, the last item of SURVEY 8f rank 4 ("More edge/vertex types the README lists as
future work: 2d, 3d, BA, Virtual Meas.", README.md:53; the sketch the reference keeps commented out, python/optimizer/edges2d.py:83-121).
The wire format cannot carry the type (remote/serialization/DeserializeGraph.h:93-95 throws), so it exists behind the C ABI only, and
nothing in the reference pins it: it is pinned by what it must be — the derivative of its residual under the reference's own vertex
update (finite differences of the dense restatement), then dense restatement -> twin -> device as for everything else."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle
from tests import independent, util
from toyslam_amd import _lib, synth


def test_residual_is_zero_at_the_truth_and_jacobians_are_its_derivatives():
    g, truth = synth.make(40, 5, loop_closures=2, seed=3, with_truth=True)
    gv = util.with_virtual_landmarks(g, 0.8, seed=1)
    vl = np.where(gv.e_type == 2)[0]
    assert len(vl) >= 10
    e, A, B = oracle.edge_eval(util.to_oracle(gv))
    assert np.all(e[vl, 2] == 0) and np.all(A[vl, 6:] == 0) and np.all(B[vl, 6:] == 0)
    # at the ground truth both poses put the point in the same place, up to the sensor noise of the two observations
    gt = gv.copy(); gt.v_pos[:] = truth
    et = oracle.edge_eval(util.to_oracle(gt))[0][vl]
    assert np.abs(et[:, :2]).max() < 1.0 and np.abs(e[vl, :2]).max() > np.abs(et[:, :2]).max()
    idx = {int(v): k for k, v in enumerate(gv.v_id)}
    h, worst = 1e-6, 0.0
    for ei in vl[:12]:
        for vi, J in ((idx[int(gv.e_ids[ei, 0])], A[ei, :6].reshape(2, 3)), (idx[int(gv.e_ids[ei, 1])], B[ei, :6].reshape(2, 3))):
            for c in range(3):                                   # x, y (world frame), theta: all additive (VertexSe2.h:16-27)
                gp, gm = gv.copy(), gv.copy()
                gp.v_pos[vi, c] += h; gm.v_pos[vi, c] -= h
                d = oracle.edge_eval(util.to_oracle(gp))[0][ei] - oracle.edge_eval(util.to_oracle(gm))[0][ei]
                worst = max(worst, np.abs(d[:2] / (2 * h) - J[:, c]).max())
    assert worst < 1e-7, worst
    # the other two edge types are untouched
    e0, A0, B0 = oracle.edge_eval(util.to_oracle(g))
    n = len(g.e_type)
    np.testing.assert_array_equal(e[:n], e0); np.testing.assert_array_equal(A[:n], A0); np.testing.assert_array_equal(B[:n], B0)


def test_dense_linearisation_with_virtual_landmarks_equals_the_numpy_checker():
    """tests/independent.py forms chi^2, gradient, diagonal blocks and H @ delta from the per-edge Jacobians alone."""
    gv = util.with_virtual_landmarks(synth.make(60, 5, loop_closures=3, seed=5), 0.5, seed=2)
    H, b, err, off = oracle.linearize(util.to_oracle(gv))
    lin = independent.Linearisation(gv)
    assert abs(lin.chi2 - err) <= 1e-12 * err
    grad = lin.gradient(); diag = lin.diag_blocks()
    for v in range(len(gv.v_id)):
        d = 3 if gv.v_type[v] == 0 else 2
        np.testing.assert_allclose(b[off[v]:off[v] + d], grad[v, :d], rtol=0, atol=1e-9 * np.abs(b).max())
        np.testing.assert_allclose(H[off[v]:off[v] + d, off[v]:off[v] + d], diag[v].reshape(3, 3)[:d, :d], rtol=0, atol=1e-9 * np.abs(H).max())
    delta = np.linalg.solve(H, b)
    dv = np.zeros((len(gv.v_id), 3))
    for v in range(len(gv.v_id)):
        d = 3 if gv.v_type[v] == 0 else 2
        dv[v, :d] = delta[off[v]:off[v] + d]
    assert lin.residual_of(dv) < 1e-10


@pytest.mark.parametrize("precond", ["amg", "jacobi"])
@pytest.mark.parametrize("jac", ["constant", "analytic"])
def test_twin_with_virtual_landmarks_matches_the_dense_restatement(precond, jac):
    """Pose-pose slots in general form (tsgo_math.h: eight numbers per slot) on the CPU twin: graphs that mix all three edge
    types, graphs whose landmarks are mostly replaced by virtual measurements, with the reference's constant ODOM Jacobians and
    with the analytic ones."""
    oracle.set_odom_jacobian(jac)
    try:
        # (the second graph leaves landmarks without edges: singular rows, which the reference's rank-revealing QR sets to zero)
        cases = [(util.with_virtual_landmarks(util.c1_arrays(), 0.4, seed=3), 8, "chol"),
                 (util.with_virtual_landmarks(synth.make(150, 6, loop_closures=4, seed=5), 0.7, seed=4, keep_lm=False), 5, "qr")]
        for g, n, solver in cases:
            assert (g.e_type == 2).sum() > 20
            rd = oracle.optimize(util.to_oracle(g), n, mode="cpp", solver=solver)
            rt = oracle.sparse_optimize(util.to_oracle(g), n, pcg_tol=1e-13, precond=precond)
            assert (rt["iters"], rt["stop"]) == (rd["iters"], rd["stop"])
            np.testing.assert_allclose(rt["chi2"], rd["chi2"], rtol=1e-10)
            assert util.max_vertex_diff(rt["v_pos"], rd["v_pos"], g.v_type) < 1e-9
    finally:
        oracle.set_odom_jacobian("constant")


def test_the_wire_codec_refuses_the_type_the_reference_cannot_read():
    gv = util.with_virtual_landmarks(synth.make(30, 4, seed=1), 0.9, seed=1)
    lib = _lib.host_lib()
    cg = gv.c_struct()
    n = lib.tsgo_wire_encode_request(C.byref(cg), None, 0)
    assert n < 0 and b"ODOM (0) and LM (1) edges only" in lib.tsgo_last_error()


def test_layout_accepts_virtual_landmarks_as_pose_pose_slots():
    g = synth.make(500, 6, seed=2)
    gv = util.with_virtual_landmarks(g, 0.5, seed=2)
    nv = int((gv.e_type == 2).sum())
    lib = _lib.host_lib()
    a, b = _lib.tsgo_layout_info(), _lib.tsgo_layout_info()
    for gg, info in ((g, a), (gv, b)):
        cg = gg.c_struct()
        _lib.check(lib, lib.tsgo_layout_probe(C.byref(cg), 0, 1, 0, 0, C.byref(info)), "tsgo_layout_probe")
    assert b.n_odom_slots == a.n_odom_slots + 2 * nv and b.n_lm_edges_local == a.n_lm_edges_local
    bad = gv.copy(); bad.e_ids[np.where(bad.e_type == 2)[0][0], 1] = gv.v_id[gv.v_type == 1][0]      # a virtual measurement must join two poses
    cg = bad.c_struct()
    assert lib.tsgo_layout_probe(C.byref(cg), 0, 1, 0, 0, C.byref(a)) != 0 and b"must join two Se2 vertices" in lib.tsgo_last_error()


def test_python_graph_model_carries_the_reference_sketchs_edge_class():
    """toyslam_amd.graph.EdgeVirtualLandmark2d has the constructor of the class the reference keeps commented out
    (python/optimizer/edges2d.py:83-89); GraphArrays flattens it to edge type 2; the wire encoder refuses it with the library's reason."""
    from toyslam_amd import remote
    from toyslam_amd.graph import EdgeLandmark2d, EdgeOdometry2d, EdgeVirtualLandmark2d, GraphArrays, OptGraph, Vertex2d, VertexPose2d

    def pose(x, y, t):
        c, s = np.cos(t), np.sin(t)
        return VertexPose2d(np.array([[c, -s, x], [s, c, y], [0, 0, 1.0]]))
    og = OptGraph()
    og.add_vertex(0, pose(0, 0, 0.1), fixed=True); og.add_vertex(1, pose(1, 0.2, 0.3)); og.add_vertex(2, Vertex2d([2.0, 1.0]))
    og.add_edge(EdgeOdometry2d(0, 1, np.array([[1, 0, 1.0], [0, 1, 0.1], [0, 0, 1.0]]), np.diag([4.0, 4.0, 65.0])))
    og.add_edge(EdgeLandmark2d(0, 2, np.array([2.2, 0.4]), np.diag([44.0, 44.0])))
    og.add_edge(EdgeVirtualLandmark2d(0, 1, np.array([2.2, 0.4]), np.array([1.3, 0.5]), np.diag([30.0, 20.0])))
    e = og.get_edges()[-1]
    assert e.get_type() == 2 and (e.get_id(0), e.get_id(1)) == (0, 1)
    arr = GraphArrays.from_optgraph(og)
    assert list(arr.e_type) == [0, 1, 2]
    np.testing.assert_array_equal(arr.e_meas[2], [2.2, 0.4, 1.3, 0.5, 0, 0, 0, 0, 0])
    np.testing.assert_array_equal(arr.e_inf[2], [30.0, 20.0, 0.0])
    r = oracle.optimize(util.to_oracle(arr), 5, mode="cpp", solver="qr")          # the dense restatement takes the flattened graph
    assert r["iters"] >= 1 and np.isfinite(r["chi2"]).all()
    with pytest.raises(RuntimeError, match="ODOM .0. and LM .1. edges only"):
        remote.graph_to_bytes(og)

#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference's Python modules.

Runs only in the authoring container (needs /root/reference, read-only).  Nothing from the
reference is copied: its modules are imported and driven; only inputs and outputs (data) are
written.  The GPU box never runs this file; it consumes the committed fixtures.

What is replayed (reference file:line):
  * the config-1 simulation of python/slam_main.py:19-147 (np.random.seed(0), ROBOT_STEPS=150,
    same RNG draw order) WITHOUT importing slam_main (it is a UI script, slam_main.py:69-97,274);
  * construct_optimizer_graph, python/slam_main.py:157-187;
  * graph_to_bytes, python/remote/graph_to_bytes.py:32-67  -> c1_request.bin;
  * EdgeLandmark2d/EdgeOdometry2d.calc_error, python/optimizer/edges2d.py:21-53,65-78;
  * GraphOptimizer.calculate_H_b / optimize, python/optimizer/graph_optimizer.py:20-155;
  * VertexPose2d/Vertex2d.update, python/optimizer/vertices.py:28-33,45-46.

Outputs (all < 1 MB):
  c1_request.bin        wire request, 106 672 B
  c1_graph.npz          SoA of the OptGraph, f64 as built by the sim
  c1_lin_f64.npz        per-edge (e, A, B), H (COO), b, err at linearisation 0 of the f64 graph
  c1_lin_wire.npz       the same for the graph a server actually receives (values rounded to f32)
  c1_pyopt.npz          chi2 per iteration + final vertices of GraphOptimizer.optimize(10, lr=.2)
                        started from the wire (f32-rounded) graph
  tiny_*.npz            three hand-checkable graphs, same content as c1_lin_f64
  update_check.npz      vertex state before/after one update() with a known delta
"""
import contextlib
import io
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
REF = "/root/reference/python"
sys.path.insert(0, REF)

import numpy as np  # noqa: E402

from environment import load_env  # noqa: E402
from lidar_sensor import calc_lidar_measurements  # noqa: E402
from optimizer.edges2d import EdgeLandmark2d, EdgeOdometry2d  # noqa: E402
from optimizer.graph_optimizer import GraphOptimizer  # noqa: E402
from optimizer.opt_graph import OptGraph  # noqa: E402
from optimizer.vertices import Vertex2d, VertexPose2d  # noqa: E402
from remote.graph_to_bytes import graph_to_bytes  # noqa: E402
from slam.graph2d import Graph2d  # noqa: E402
from slam.slam_helper import add_to_graph, motion_model  # noqa: E402
from tools import angle_to_mat_2d, mat_to_angle_2d  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


# --------------------------------------------------------------------------- config-1 sim replay
def build_c1():
    np.random.seed(0)
    ROBOT_STEPS = 150
    lidar_fov = np.deg2rad(120)
    lidar_ray_step = np.deg2rad(6)
    lidar_std = .15
    pos_std = 0.5
    ori_std = np.deg2rad(7.1)
    LIDAR_NOISE = np.identity(2)
    LIDAR_NOISE[0, 0] = lidar_std ** 2
    LIDAR_NOISE[1, 1] = lidar_std ** 2
    ODOMETRY_NOISE = np.identity(3)
    ODOMETRY_NOISE[:2, :2] *= pos_std ** 2
    ODOMETRY_NOISE[2, 2] *= ori_std ** 2
    LIDAR_INF = np.linalg.inv(LIDAR_NOISE) * 1
    ODOM_INF = np.linalg.inv(ODOMETRY_NOISE) * 1

    environment, radius = load_env()
    state_mat = np.eye(3, 3, dtype=float)
    state_mat[0, 2] = 5
    state_mat[1, 2] = 15
    state_mat_gt = np.copy(state_mat)

    graph = Graph2d()
    lms, ids = calc_lidar_measurements(state_mat, environment, radius, lidar_fov, lidar_ray_step)
    assert lms is not None
    pos_id = add_to_graph(graph, state_mat, lms, ids, LIDAR_NOISE, True)

    def motion(pid):
        table = [(10, 3., 2.0), (20, 6., 0.9), (40, -6., 0.9), (60, 5., 0.8)]
        deg, step = 3., 0.7
        for lim, d, s in table:
            if pid < lim:
                deg, step = d, s
                break
        m = np.eye(3, 3)
        m[:2, :2] = angle_to_mat_2d(np.deg2rad(deg))
        m[:2, 2] = np.array([step, 0.0])
        return m

    guard = 0
    while graph.get_size() < ROBOT_STEPS:
        guard += 1
        assert guard < 10000
        tr = motion(pos_id)
        state_mat_gt = motion_model(state_mat_gt, tr)
        lms, ids = calc_lidar_measurements(state_mat_gt, environment, radius, lidar_fov, lidar_ray_step)
        if lms is None or ids is None:
            continue
        RT = np.copy(tr)
        RT[0, 2] = RT[0, 2] + np.random.normal(0, ODOMETRY_NOISE[0, 0])
        RT[1, 2] = RT[1, 2] + np.random.normal(0, ODOMETRY_NOISE[1, 1])
        RT[:2, :2] = angle_to_mat_2d(mat_to_angle_2d(tr[:2, :2]) + np.random.normal(0, ODOMETRY_NOISE[2, 2]))
        RT[2, 2] = 1
        graph.get_pose(pos_id).set_odometry(RT)
        state_mat = motion_model(state_mat, RT)
        pos_id = add_to_graph(graph, state_mat, lms, ids, LIDAR_NOISE, False)

    # construct_optimizer_graph (slam_main.py:157-187)
    g = OptGraph()
    positions = graph.get_positions()
    for i in range(len(positions)):
        p = positions[i]
        g.add_vertex(p.id, VertexPose2d(p.position), p.is_fixed)
    for i in range(len(positions)):
        p = positions[i]
        if p.odometry is not None:
            g.add_edge(EdgeOdometry2d(p.id, p.id + 1, p.odometry, ODOM_INF))
    lm_glob = len(positions)
    lm_map = {}
    for i in range(len(positions)):
        p = positions[i]
        for lm_id in p.landmark_measurements:
            if lm_id not in lm_map:
                lm_map[lm_id] = lm_glob
                lm_glob += 1
            g.add_edge(EdgeLandmark2d(p.id, lm_map[lm_id], p.landmark_measurements[lm_id], LIDAR_INF))
    landmarks = graph.get_landmarks()
    for lm_id in landmarks:
        if lm_id not in lm_map:
            continue
        g.add_vertex(lm_map[lm_id], Vertex2d(landmarks[lm_id]), False)
    return g


# --------------------------------------------------------------------------- helpers
def graph_soa(g):
    """Plain arrays describing an OptGraph (f64 as held by the reference objects)."""
    vid, vtype, vpos = [], [], []
    for k, v in g.vertices.items():
        vid.append(k)
        vtype.append(v.get_type())
        if v.get_type() == 0:
            vpos.append([v.position[0, 2], v.position[1, 2], mat_to_angle_2d(v.position[:2, :2])])
        else:
            vpos.append([v.position[0], v.position[1], 0.0])
    etype, eid, emeas, einf = [], [], [], []
    for e in g.edges:
        etype.append(e.get_type())
        eid.append([e.id_1, e.id_2])
        m = np.zeros(9)
        w = np.zeros(3)
        if e.get_type() == 0:
            m[:] = np.asarray(e.measurement, dtype=np.float64).reshape(-1)
            w[:] = np.diag(e.information)
        else:
            m[:2] = e.measurement
            w[:2] = np.diag(e.information)
        emeas.append(m)
        einf.append(w)
    return dict(v_id=np.array(vid, np.uint32), v_type=np.array(vtype, np.uint32),
                v_pos=np.array(vpos, np.float64), e_type=np.array(etype, np.uint32),
                e_ids=np.array(eid, np.uint32), e_meas=np.array(emeas, np.float64),
                e_inf=np.array(einf, np.float64),
                fixed=np.array(sorted(g.fixed_vertices), np.uint32))


def graph_from_soa(s, as_wire):
    """Rebuild an OptGraph from SoA with the reference classes.  as_wire=True rounds every
    float to f32 first (what python/remote/graph_to_bytes.py:4-7 puts on the wire) and rebuilds
    pose matrices from (x, y, theta) exactly as a receiver must."""
    def r(x):
        return np.float64(np.float32(x)) if as_wire else np.float64(x)
    g = OptGraph()
    for k, t, p in zip(s["v_id"], s["v_type"], s["v_pos"]):
        if t == 0:
            th = r(p[2])
            m = np.array([[np.cos(th), -np.sin(th), r(p[0])],
                          [np.sin(th), np.cos(th), r(p[1])],
                          [0, 0, 1]], dtype=np.float64)
            g.add_vertex(int(k), VertexPose2d(m))
        else:
            g.add_vertex(int(k), Vertex2d(np.array([r(p[0]), r(p[1])], dtype=np.float64)))
    for t, ids, m, w in zip(s["e_type"], s["e_ids"], s["e_meas"], s["e_inf"]):
        if t == 0:
            mm = np.array([r(x) for x in m], dtype=np.float64).reshape(3, 3)
            g.add_edge(EdgeOdometry2d(int(ids[0]), int(ids[1]), mm, np.diag([r(x) for x in w])))
        else:
            g.add_edge(EdgeLandmark2d(int(ids[0]), int(ids[1]), np.array([r(m[0]), r(m[1])]),
                                      np.diag([r(w[0]), r(w[1])])))
    for k in s["fixed"]:
        g.fix_vertex(int(k))
    return g


def linearisation(g):
    """Per-edge (e, A, B) and the reference's own H, b, err (graph_optimizer.py:94-155)."""
    E = len(g.edges)
    e_out = np.zeros((E, 3))
    A_out = np.zeros((E, 9))
    B_out = np.zeros((E, 9))
    for i, ed in enumerate(g.edges):
        e, A, B = ed.calc_error(g)
        e_out[i, :len(e)] = e
        A_out[i, :A.size] = np.asarray(A).reshape(-1)
        B_out[i, :B.size] = np.asarray(B).reshape(-1)
    opt = GraphOptimizer(g)
    err = quiet(opt.calculate_H_b)
    H = opt.H
    nz = np.nonzero(H)
    order = np.array([opt.vertex_ids_map[k] for k in g.vertices], np.int64)
    return dict(edge_e=e_out, edge_A=A_out, edge_B=B_out,
                H_row=nz[0].astype(np.int32), H_col=nz[1].astype(np.int32), H_val=H[nz],
                H_sum=np.float64(H.sum()), H_trace=np.float64(np.trace(H)),
                b=opt.b.copy(), err=np.float64(err), index_of_vertex=order,
                n=np.int64(H.shape[0]))


def tiny_graphs():
    """Three hand-checkable graphs (also exercise Huber on/off and duplicate fixed ids)."""
    out = {}
    # (1) two poses, one landmark seen from both, one odometry edge
    g = OptGraph()
    def pose(x, y, th):
        return np.array([[np.cos(th), -np.sin(th), x], [np.sin(th), np.cos(th), y], [0, 0, 1.0]])
    g.add_vertex(0, VertexPose2d(pose(0.0, 0.0, 0.0)), True)
    g.add_vertex(1, VertexPose2d(pose(1.1, 0.1, 0.05)))
    g.add_vertex(2, Vertex2d(np.array([2.0, 1.0])))
    g.add_edge(EdgeOdometry2d(0, 1, pose(1.0, 0.0, 0.0), np.diag([4.0, 4.0, 65.0])))
    g.add_edge(EdgeLandmark2d(0, 2, np.array([2.2, 0.45]), np.diag([44.0, 44.0])))
    g.add_edge(EdgeLandmark2d(1, 2, np.array([1.3, 0.80]), np.diag([44.0, 44.0])))
    out["tiny_a"] = g
    # (2) pure odometry chain with a loop closure, small residuals (no Huber)
    g = OptGraph()
    for i in range(5):
        a = 2 * np.pi * i / 5
        g.add_vertex(10 + i, VertexPose2d(pose(np.cos(a) + 0.01 * i, np.sin(a) - 0.02 * i, a + np.pi / 2 + 0.01)),
                     i == 0)
    for i in range(5):
        a0, a1 = 2 * np.pi * i / 5, 2 * np.pi * (i + 1) / 5
        T0 = pose(np.cos(a0), np.sin(a0), a0 + np.pi / 2)
        T1 = pose(np.cos(a1), np.sin(a1), a1 + np.pi / 2)
        g.add_edge(EdgeOdometry2d(10 + i, 10 + (i + 1) % 5, np.linalg.inv(T0) @ T1, np.diag([4.0, 4.0, 65.0])))
    out["tiny_b"] = g
    # (3) one pose, three landmarks, one far off (Huber active), ids sparse / unordered
    g = OptGraph()
    g.add_vertex(7, VertexPose2d(pose(-3.0, 2.0, -2.5)), True)
    g.add_vertex(1000, Vertex2d(np.array([-4.0, 0.5])))
    g.add_vertex(3, Vertex2d(np.array([-6.0, 3.0])))
    g.add_vertex(42, Vertex2d(np.array([5.0, 5.0])))
    g.add_edge(EdgeLandmark2d(7, 3, np.array([3.1, 0.4]), np.diag([44.0, 30.0])))
    g.add_edge(EdgeLandmark2d(7, 1000, np.array([1.9, -1.0]), np.diag([44.0, 44.0])))
    g.add_edge(EdgeLandmark2d(7, 42, np.array([2.0, 2.0]), np.diag([10.0, 44.0])))
    out["tiny_c"] = g
    return out


def main():
    g = quiet(build_c1)
    s = graph_soa(g)
    P = int((s["v_type"] == 0).sum())
    L = int((s["v_type"] == 1).sum())
    Eo = int((s["e_type"] == 0).sum())
    El = int((s["e_type"] == 1).sum())
    print("C1: P=%d L=%d Eo=%d El=%d fixed=%s" % (P, L, Eo, El, s["fixed"]))
    req = quiet(graph_to_bytes, g)
    print("request bytes:", len(req))
    assert (P, L, Eo, El, len(req)) == (150, 342, 149, 1974, 106672)
    with open(os.path.join(OUT, "c1_request.bin"), "wb") as f:
        f.write(req)
    np.savez_compressed(os.path.join(OUT, "c1_graph.npz"), **s)

    lin = linearisation(g)
    print("err f64 graph:", repr(float(lin["err"])), "sumH", float(lin["H_sum"]), "sum b", lin["b"].sum())
    assert abs(float(lin["err"]) - 114586.1496325) < 1e-6
    np.savez_compressed(os.path.join(OUT, "c1_lin_f64.npz"), **lin)

    gw = graph_from_soa(s, as_wire=True)
    linw = linearisation(gw)
    print("err wire graph:", repr(float(linw["err"])))
    np.savez_compressed(os.path.join(OUT, "c1_lin_wire.npz"), **linw)

    # the reference optimizer's own 10-iteration trajectory from the wire graph
    gw = graph_from_soa(s, as_wire=True)
    opt = GraphOptimizer(gw)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        opt.optimize(10, 0.2)
    chi = [float(l.split()[1]) for l in buf.getvalue().splitlines() if l.startswith("err:")]
    print("python optimizer chi2:", chi)
    fin = graph_soa(gw)
    np.savez_compressed(os.path.join(OUT, "c1_pyopt.npz"), chi2=np.array(chi),
                        v_id=fin["v_id"], v_type=fin["v_type"], v_pos=fin["v_pos"])

    for name, tg in tiny_graphs().items():
        ts = graph_soa(tg)
        tl = linearisation(tg)
        ts.update(tl)
        ts["request"] = np.frombuffer(quiet(graph_to_bytes, tg), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **ts)
        print(name, "err", float(tl["err"]), "n", int(tl["n"]))

    # vertex update (vertices.py:28-33,45-46)
    vp = VertexPose2d(np.array([[np.cos(3.0), -np.sin(3.0), 1.5], [np.sin(3.0), np.cos(3.0), -2.0], [0, 0, 1.0]]))
    d = np.array([0.3, -0.2, 0.4])
    before = [vp.position[0, 2], vp.position[1, 2], mat_to_angle_2d(vp.position[:2, :2])]
    vp.update(d)
    after = [vp.position[0, 2], vp.position[1, 2], mat_to_angle_2d(vp.position[:2, :2])]
    vl = Vertex2d(np.array([1.0, 2.0]))
    vl.update(np.array([0.5, -0.25]))
    np.savez_compressed(os.path.join(OUT, "update_check.npz"), pose_before=np.array(before), delta=d,
                        pose_after=np.array(after), pose_mat_after=vp.position, lm_after=vl.position)
    print("done")


if __name__ == "__main__":
    main()

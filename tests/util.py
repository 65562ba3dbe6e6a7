"""Shared helpers for the tests (the oracle is imported HERE and in tests only)."""
import os

import numpy as np

from oracle import oracle
from toyslam_amd.graph import GraphArrays

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def to_oracle(g):
    return oracle.Graph(g.v_id, g.v_type, g.v_pos, g.e_type, g.e_ids, g.e_meas, g.e_inf, g.fixed)


def to_arrays(g):
    return GraphArrays(g.v_id, g.v_type, g.v_pos, g.e_type, g.e_ids, g.e_meas, g.e_inf, g.fixed)


def c1_arrays(as_wire=True):
    z = load("c1_graph.npz")
    g = GraphArrays(z["v_id"], z["v_type"], z["v_pos"], z["e_type"], z["e_ids"], z["e_meas"], z["e_inf"], z["fixed"])
    return g.rounded_to_wire() if as_wire else g


def tiny_arrays(name):
    z = load(name + ".npz")
    return GraphArrays(z["v_id"], z["v_type"], z["v_pos"], z["e_type"], z["e_ids"], z["e_meas"], z["e_inf"], z["fixed"])


def dense_solution(g):
    """H delta = b by the dense oracle + LAPACK; returns per-vertex (n,3) delta, chi2, diag blocks, grad."""
    o = to_oracle(g)
    H, b, err, idx = oracle.linearize(o)
    x = np.linalg.solve(H, b)
    dims = np.where(o.v_type == 0, 3, 2)
    d = np.zeros((len(o.v_id), 3)); gr = np.zeros((len(o.v_id), 3)); dg = np.zeros((len(o.v_id), 9))
    for i in range(len(o.v_id)):
        k, m = idx[i], dims[i]
        d[i, :m] = x[k:k + m]; gr[i, :m] = b[k:k + m]
        blk = np.zeros((3, 3)); blk[:m, :m] = H[k:k + m, k:k + m]
        dg[i] = blk.reshape(-1)
    return d, err, dg, gr


def angle_diff(a, b):
    return (a - b + np.pi) % (2 * np.pi) - np.pi


def max_vertex_diff(v1, v2, v_type):
    d = np.abs(v1 - v2)
    d[:, 2] = np.where(v_type == 0, np.abs(angle_diff(v1[:, 2], v2[:, 2])), 0.0)
    return float(d.max())


def first_poses(g, n_poses):
    """The graph a SLAM front-end held when it had seen the first `n_poses` poses of g: those poses (ids 0..n_poses-1 in the
    synthetic graphs), the landmarks they observe and the edges among them, with g's own vertex ids (python/slam_main.py:157-187
    rebuilds and resends exactly such a growing graph)."""
    pose = g.v_type == 0
    keep_pose_id = set(int(i) for i in g.v_id[pose][:n_poses])
    e1_in = np.isin(g.e_ids[:, 0], list(keep_pose_id))
    lm_edge = (g.e_type == 1) & e1_in
    keep_lm_id = np.unique(g.e_ids[lm_edge, 1])
    keep_v = (pose & np.isin(g.v_id, list(keep_pose_id))) | ((g.v_type == 1) & np.isin(g.v_id, keep_lm_id))
    kept_ids = g.v_id[keep_v]
    keep_e = np.isin(g.e_ids[:, 0], kept_ids) & np.isin(g.e_ids[:, 1], kept_ids)
    return GraphArrays(g.v_id[keep_v], g.v_type[keep_v], g.v_pos[keep_v], g.e_type[keep_e], g.e_ids[keep_e], g.e_meas[keep_e], g.e_inf[keep_e],
                       g.fixed[np.isin(g.fixed, kept_ids)])


def with_virtual_landmarks(g, fraction=0.3, seed=0, keep_lm=True):
    """A copy of `g` in which, for a `fraction` of the landmarks seen from at least two poses, one pair of its LM observations
    (pose a, pose b) becomes ONE virtual landmark measurement (edge type 2, include/tsgo.h) between a and b:
    meas = (range_a, bearing_a, range_b, bearing_b), information = the first observation's.  keep_lm = False also drops the two
    LM edges it was made from (a landmark may then be left without edges: allowed)."""
    import numpy as np
    from toyslam_amd.graph import GraphArrays
    rng = np.random.default_rng(seed)
    by_lm = {}
    for k in np.where(g.e_type == 1)[0]:
        by_lm.setdefault(int(g.e_ids[k, 1]), []).append(int(k))
    e_type, e_ids, e_meas, e_inf = [g.e_type], [g.e_ids], [g.e_meas], [g.e_inf]
    drop = np.zeros(len(g.e_type), bool)
    nt, ni, nm, nf = [], [], [], []
    for lm, ks in by_lm.items():
        if len(ks) < 2 or rng.random() > fraction:
            continue
        a, b = rng.choice(ks, size=2, replace=False)
        if g.e_ids[a, 0] == g.e_ids[b, 0]:
            continue
        m = np.zeros(9); m[0:2] = g.e_meas[a, 0:2]; m[2:4] = g.e_meas[b, 0:2]
        nt.append(2); ni.append([g.e_ids[a, 0], g.e_ids[b, 0]]); nm.append(m); nf.append([g.e_inf[a, 0], g.e_inf[a, 1], 0.0])
        if not keep_lm:
            drop[a] = drop[b] = True
    if not nt:
        return g.copy()
    keep = ~drop
    return GraphArrays(g.v_id, g.v_type, g.v_pos,
                       np.concatenate([g.e_type[keep], np.array(nt, np.uint32)]), np.concatenate([g.e_ids[keep], np.array(ni, np.uint32)]),
                       np.concatenate([g.e_meas[keep], np.array(nm)]), np.concatenate([g.e_inf[keep], np.array(nf)]), g.fixed)

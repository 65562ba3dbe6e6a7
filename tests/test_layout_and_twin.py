"""Host layout builder + shard planner (product code, csrc/host/problem.cpp) and the sparse CPU twin
(oracle/oracle_sparse.cpp) against the dense restatement."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle
from tests import util
from toyslam_amd import _lib, synth


def probe(g, rank=0, world=1, gp=0, gl=0):
    lib = _lib.host_lib()
    info = _lib.tsgo_layout_info()
    cg = g.c_struct()
    _lib.check(lib, lib.tsgo_layout_probe(C.byref(cg), rank, world, gp, gl, C.byref(info)), "tsgo_layout_probe")
    return info


def test_layout_counts_c1():
    g = util.c1_arrays()
    i = probe(g)
    assert (i.n_pose, i.n_lm_local, i.n_lm_total, i.n_lm_edges_local) == (150, 342, 342, 1974)
    assert i.n_odom_slots == 2 * 149                          # every ODOM edge is listed at both ends
    assert i.rows_by_pose * 64 >= 1974 and i.rows_by_lm * 64 >= 1974
    assert i.lanes_per_pose in (1, 2, 4, 8) and i.lanes_per_lm in (1, 2, 4, 8)


def test_padding_stays_moderate_on_synthetic_graphs():
    g = synth.make(5000, 10, seed=1)
    i = probe(g)
    n_lm_edges = int((g.e_type == 1).sum())
    assert i.rows_by_pose * 64 < 1.2 * n_lm_edges
    assert i.rows_by_lm * 64 < 1.6 * n_lm_edges              # degree-sorted windows keep landmark padding low


@pytest.mark.parametrize("world", [2, 3, 8])
def test_shards_partition_landmarks_edges_and_poses(world):
    g = synth.make(3000, 10, seed=2)
    infos = [probe(g, r, world) for r in range(world)]
    assert infos[0].lm_first == 0 and infos[-1].lm_last == g.n_landmarks
    assert infos[0].pose_first == 0 and infos[-1].pose_last == g.n_poses
    for a, b in zip(infos, infos[1:]):
        assert a.lm_last == b.lm_first and a.pose_last == b.pose_first
    assert sum(i.n_lm_edges_local for i in infos) == int((g.e_type == 1).sum())
    assert sum(i.n_odom_slots for i in infos) == 2 * int((g.e_type == 0).sum())
    share = np.array([i.n_lm_edges_local for i in infos], float)
    assert share.max() / share.mean() < 1.1                   # balanced by edge count


def test_every_shard_chooses_the_same_lanes_per_pose():
    """The ranks all-reduce [3P | one partial per workgroup of the pose table]: the table's shape — lanes per pose — must be the
    same on every rank.  On this graph the two shards' own mean degrees are 5.998 and 6.001, either side of the 2-lane
    threshold (found by tests/research/soak_sharded_gpu.py as "ranks disagree on the buffer size")."""
    g = synth.make(2077, 12, loop_closures=12, seed=741807)
    for world in (2, 3, 5):
        infos = [probe(g, rank, world) for rank in range(world)]
        assert len({i.lanes_per_pose for i in infos}) == 1 and len({i.lanes_per_lm for i in infos}) == 1
        assert len({i.n_pose for i in infos}) == 1


def test_layout_rejects_inconsistent_graphs():
    g = util.tiny_arrays("tiny_a")
    bad = g.copy(); bad.v_id[1] = bad.v_id[0]
    with pytest.raises(RuntimeError, match="duplicate"):
        probe(bad)
    bad = g.copy(); bad.fixed = np.array([77], np.uint32)
    with pytest.raises(RuntimeError, match="fixed vertex"):
        probe(bad)
    bad = g.copy(); bad.e_meas[0] = 0                           # singular ODOM measurement
    with pytest.raises(RuntimeError, match="singular"):
        probe(bad)


def test_twin_one_step_equals_dense_solve_c1():
    g = util.c1_arrays()
    d_ref, err, _, _ = util.dense_solution(g)
    r = oracle.sparse_step(util.to_oracle(g), 1e-13)
    assert abs(r["chi2"] - err) < 1e-12 * err
    assert np.abs(r["delta"] - d_ref).max() < 1e-10 * np.abs(d_ref).max()


@pytest.mark.parametrize("name", ["tiny_a", "tiny_b", "tiny_c"])
def test_twin_full_run_equals_dense_on_tiny_graphs(name):
    g = util.tiny_arrays(name)
    ref = oracle.optimize(util.to_oracle(g), 20, mode="cpp", solver="qr")
    r = oracle.sparse_optimize(util.to_oracle(g), 20, pcg_tol=1e-14)
    assert r["stop"] == ref["stop"] and r["iters"] == ref["iters"]
    np.testing.assert_allclose(r["chi2"], ref["chi2"], rtol=1e-10, atol=1e-13)
    assert util.max_vertex_diff(r["v_pos"], ref["v_pos"], g.v_type) < 1e-9


def test_twin_full_run_equals_dense_c1():
    g = util.c1_arrays()
    ref = oracle.optimize(util.to_oracle(g), 50, mode="cpp", solver="chol")
    r = oracle.sparse_optimize(util.to_oracle(g), 50, pcg_tol=1e-13)
    assert r["stop"] == ref["stop"] == "plateau" and r["iters"] == ref["iters"]
    np.testing.assert_allclose(r["chi2"], ref["chi2"], rtol=1e-10)
    assert util.max_vertex_diff(r["v_pos"], ref["v_pos"], g.v_type) < 1e-9


def test_twin_equals_dense_on_a_synthetic_graph_with_loop_closures():
    g = synth.make(250, 8, loop_closures=20, seed=4)
    d_ref, err, _, _ = util.dense_solution(g)
    r = oracle.sparse_step(util.to_oracle(g), 1e-13)
    assert abs(r["chi2"] - err) < 1e-12 * err
    assert np.abs(r["delta"] - d_ref).max() < 1e-9 * np.abs(d_ref).max()


def test_twin_against_scipy_sparse_direct_solve_at_2k_poses():
    """Independent check at a size the dense path cannot reach: assemble H from the dense oracle's
    per-edge (e, A, B) in scipy.sparse and solve directly."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    g = synth.make(2000, 10, seed=6)
    o = util.to_oracle(g)
    e, A, B = oracle.edge_eval(o)
    dims = np.where(o.v_type == 0, 3, 2); off = np.concatenate([[0], np.cumsum(dims)[:-1]])
    pos = {int(k): i for i, k in enumerate(o.v_id)}
    n = int(dims.sum()); rows, cols, vals = [], [], []; b = np.zeros(n); chi = 0.0
    for k in range(len(o.e_type)):
        v1, v2 = pos[int(o.e_ids[k, 0])], pos[int(o.e_ids[k, 1])]
        m = 3 if o.e_type[k] == 0 else 2
        d1, d2 = dims[v1], dims[v2]
        Ak = A[k, :m * d1].reshape(m, d1); Bk = B[k, :m * d2].reshape(m, d2); ek = e[k, :m]; w = o.e_inf[k, :m]
        c2 = float((ek * ek * w).sum())
        hw = 1.0 if c2 <= 2.25 else 1.5 / np.sqrt(c2)
        chi += c2 if c2 <= 2.25 else 2 * np.sqrt(c2) * 1.5 - 2.25
        W = np.diag(w * hw)
        for (i0, X), (j0, Y) in [((off[v1], Ak), (off[v1], Ak)), ((off[v2], Bk), (off[v2], Bk)),
                                 ((off[v1], Ak), (off[v2], Bk)), ((off[v2], Bk), (off[v1], Ak))]:
            blk = X.T @ W @ Y
            r_, c_ = np.meshgrid(np.arange(blk.shape[0]) + i0, np.arange(blk.shape[1]) + j0, indexing="ij")
            rows.append(r_.ravel()); cols.append(c_.ravel()); vals.append(blk.ravel())
        b[off[v1]:off[v1] + d1] -= Ak.T @ W @ ek; b[off[v2]:off[v2] + d2] -= Bk.T @ W @ ek
    for f in o.fixed:
        i0 = off[pos[int(f)]]
        for k in range(dims[pos[int(f)]]):
            rows.append([i0 + k]); cols.append([i0 + k]); vals.append([1e6])
    H = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    x = spl.spsolve(H, b)
    r = oracle.sparse_step(o, 1e-13)
    d = np.concatenate([r["delta"][i, :dims[i]] for i in range(len(dims))])
    assert abs(r["chi2"] - chi) < 1e-10 * chi
    assert np.abs(d - x).max() < 1e-8 * np.abs(x).max()


@pytest.mark.parametrize("n_poses", [40, 300, 3000])
def test_twin_multigrid_preconditioner_gives_the_same_solution_in_far_fewer_iterations(n_poses):
    g = synth.make(n_poses, 10, loop_closures=10, seed=8)
    o = util.to_oracle(g)
    a = oracle.sparse_step(o, 1e-12, precond="jacobi")
    b = oracle.sparse_step(o, 1e-12, precond="amg")
    assert abs(a["chi2"] - b["chi2"]) <= 1e-13 * a["chi2"]
    assert np.abs(a["delta"] - b["delta"]).max() <= 1e-8 * np.abs(a["delta"]).max()
    if n_poses >= 300:
        assert b["cg_iters"] * 5 < a["cg_iters"]


def test_twin_multigrid_full_run_c1_matches_dense():
    g = util.c1_arrays()
    ref = oracle.optimize(util.to_oracle(g), 50, mode="cpp", solver="chol")
    r = oracle.sparse_optimize(util.to_oracle(g), 50, pcg_tol=1e-13, precond="amg")
    assert r["stop"] == ref["stop"] and r["iters"] == ref["iters"]
    np.testing.assert_allclose(r["chi2"], ref["chi2"], rtol=1e-10)
    assert util.max_vertex_diff(r["v_pos"], ref["v_pos"], g.v_type) < 1e-9
    assert r["cg_iters"].max() < 30


def _amg_info(g):
    lib = _lib.host_lib(); info = _lib.tsgo_amg_info(); cg = g.c_struct()
    _lib.check(lib, lib.tsgo_amg_probe(C.byref(cg), C.byref(info)), "tsgo_amg_probe")
    return info


@pytest.mark.parametrize("shape", ["landmarks", "odometry_only", "no_odometry", "two_trajectories"])
def test_matched_aggregates_have_bounded_sizes_and_coarsen_every_level(shape):
    """host/amg.cpp aggregate_by_matching: every level shrinks by about its target (4, 4, 8 ...), no aggregate grows
    past 1.5x the target plus the small groups it absorbed, on chains, landmark-only graphs and disconnected pieces."""
    from toyslam_amd.graph import GraphArrays
    g = synth.make(3000, 10, loop_closures=30, seed=4)
    if shape == "odometry_only":
        keep = g.e_type == 0
        pose = g.v_type == 0
        g = GraphArrays(g.v_id[pose], g.v_type[pose], g.v_pos[pose], g.e_type[keep], g.e_ids[keep], g.e_meas[keep], g.e_inf[keep], g.fixed)
    elif shape == "no_odometry":
        keep = g.e_type == 1
        g = GraphArrays(g.v_id, g.v_type, g.v_pos, g.e_type[keep], g.e_ids[keep], g.e_meas[keep], g.e_inf[keep], g.fixed)
    elif shape == "two_trajectories":      # cut the odometry chain in the middle and drop the landmarks both halves see
        P = int((g.v_type == 0).sum()); half = P // 2
        pose_of = {int(v): i for i, v in enumerate(g.v_id[g.v_type == 0])}
        side = {}
        for t, (a, b) in zip(g.e_type, g.e_ids):
            if t == 1:
                side.setdefault(int(b), set()).add(pose_of[int(a)] < half)
        keep = np.array([(t == 0 and not (pose_of[int(a)] < half) != (pose_of[int(b)] < half)) or (t == 1 and len(side[int(b)]) == 1)
                         for t, (a, b) in zip(g.e_type, g.e_ids)])
        second_anchor = g.v_id[g.v_type == 0][half]      # every connected piece needs its own gauge
        g = GraphArrays(g.v_id, g.v_type, g.v_pos, g.e_type[keep], g.e_ids[keep], g.e_meas[keep], g.e_inf[keep],
                        np.concatenate([g.fixed, [second_anchor]]).astype(g.fixed.dtype))
    info = _amg_info(g)
    targets = [4, 4, 8, 8, 8, 8]
    assert info.n_levels >= 3
    for l in range(info.n_levels - 1):
        assert 1 <= info.agg_min[l] and info.agg_max[l] <= 3 * targets[l], (l, info.agg_min[l], info.agg_max[l])
        assert info.rows[l + 1] * 2 <= info.rows[l], "level %d does not coarsen: %d -> %d" % (l, info.rows[l], info.rows[l + 1])
    assert info.rows[info.n_levels - 1] <= 28
    # and the preconditioner built on them solves the system
    o = util.to_oracle(g)
    a = oracle.sparse_step(o, 1e-12, precond="jacobi"); b = oracle.sparse_step(o, 1e-12, precond="amg")
    assert np.abs(a["delta"] - b["delta"]).max() <= 1e-7 * max(np.abs(a["delta"]).max(), 1e-30)
    assert b["cg_iters"] <= 80


def test_hub_landmark_keeps_the_multigrid_lists_bounded_and_the_answer_exact():
    """One landmark observed from every pose (d = 400 -> 160 000 pose pairs) must not blow up the
    preconditioner's gather lists, and the solve stays exact."""
    g = synth.make(400, 6, seed=9)
    P = g.n_poses
    hub = g.v_id[g.v_type == 1][0]
    extra = P
    g2 = util.to_arrays(g)
    import numpy as np
    e_type = np.concatenate([g.e_type, np.ones(extra, np.uint32)])
    e_ids = np.concatenate([g.e_ids, np.stack([np.arange(P, dtype=np.uint32), np.full(P, hub, np.uint32)], 1)])
    meas = np.zeros((extra, 9)); meas[:, 0] = 5.0; meas[:, 1] = np.linspace(-1, 1, extra)
    inf = np.tile([1.0, 1.0, 0.0], (extra, 1))
    from toyslam_amd.graph import GraphArrays
    g2 = GraphArrays(g.v_id, g.v_type, g.v_pos, e_type, e_ids, np.concatenate([g.e_meas, meas]), np.concatenate([g.e_inf, inf]), g.fixed)
    lib = _lib.host_lib(); info = _lib.tsgo_amg_info(); cg = g2.c_struct()
    _lib.check(lib, lib.tsgo_amg_probe(C.byref(cg), C.byref(info)), "tsgo_amg_probe")
    assert info.schur_contribs < 40 * len(e_type)
    d_ref, err, _, _ = util.dense_solution(g2)
    r = oracle.sparse_step(util.to_oracle(g2), 1e-12, precond="amg")
    assert np.abs(r["delta"] - d_ref).max() <= 1e-8 * np.abs(d_ref).max()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_multigrid_lists_partition_the_whole_graphs_lists(world):
    """Edge-sharded runs replicate the hierarchy and split only the level-0 contribution lists (host/amg.cpp:
    build_amg_sharded): same levels on every rank, and every landmark-pair / odometry term is summed by exactly one rank."""
    lib = _lib.host_lib()
    g = synth.make(3000, 10, loop_closures=40, seed=4)
    cg = g.c_struct()
    whole = _lib.tsgo_amg_info(); od_whole = C.c_int64()
    _lib.check(lib, lib.tsgo_amg_probe_shard(C.byref(cg), 0, 1, C.byref(whole), C.byref(od_whole)), "tsgo_amg_probe_shard")
    pairs = odoms = 0
    for rank in range(world):
        info = _lib.tsgo_amg_info(); od = C.c_int64()
        _lib.check(lib, lib.tsgo_amg_probe_shard(C.byref(cg), rank, world, C.byref(info), C.byref(od)), "tsgo_amg_probe_shard")
        assert info.n_levels == whole.n_levels
        assert list(info.rows) == list(whole.rows) and list(info.blocks) == list(whole.blocks) and list(info.p_blocks) == list(whole.p_blocks)
        pairs += info.schur_contribs; odoms += od.value
    assert pairs == whole.schur_contribs and odoms == od_whole.value and pairs > 0 and odoms > 0


def test_eight_shards_of_config_3_all_reduce_buffers_of_equal_length():
    """BASELINE config 4 at its size (100k poses / 1M LM edges, 8 ranks).  What the ranks all-reduce: [18 P + one chi^2 partial per
    workgroup of the pose table] after the linearisation, [3 P + the same partials] after every Schur product, [9 x level-0
    blocks] after k_schur_blocks.  Their lengths follow from P, the lanes per pose (pose-table shape) and the level-0 pattern:
    all three must be the same on every rank (RCCL hangs otherwise), the landmark / pose ranges must tile the graph, and the
    level-0 contribution lists must partition the unsharded ones."""
    lib = _lib.host_lib()
    g = synth.make_config("c3_100k")
    cg = g.c_struct()
    world = 8
    infos = [probe(g, r, world) for r in range(world)]
    assert len({(i.lanes_per_pose, i.lanes_per_lm, i.n_pose) for i in infos}) == 1
    assert infos[0].lm_first == 0 and infos[-1].lm_last == g.n_landmarks and infos[0].pose_first == 0 and infos[-1].pose_last == g.n_poses
    for a, b in zip(infos, infos[1:]):
        assert a.lm_last == b.lm_first and a.pose_last == b.pose_first
    assert sum(i.n_lm_edges_local for i in infos) == int((g.e_type == 1).sum())
    share = np.array([i.n_lm_edges_local for i in infos], float)
    assert share.max() / share.mean() < 1.02
    whole = _lib.tsgo_amg_info(); od_whole = C.c_int64()
    _lib.check(lib, lib.tsgo_amg_probe_shard(C.byref(cg), 0, 1, C.byref(whole), C.byref(od_whole)), "tsgo_amg_probe_shard")
    pairs = odoms = 0
    for rank in range(world):
        info = _lib.tsgo_amg_info(); od = C.c_int64()
        _lib.check(lib, lib.tsgo_amg_probe_shard(C.byref(cg), rank, world, C.byref(info), C.byref(od)), "tsgo_amg_probe_shard")
        assert info.n_levels == whole.n_levels and list(info.rows) == list(whole.rows)
        assert list(info.blocks) == list(whole.blocks) and list(info.p_blocks) == list(whole.p_blocks)     # blocks[0] x 9 = the 60 MB all-reduce
        pairs += info.schur_contribs; odoms += od.value
    assert pairs == whole.schur_contribs and odoms == od_whole.value


def test_eight_shards_of_config_5_agree_on_the_pose_table_shape():
    """The same shape rule at 1 M poses / 10.1 M edges (BASELINE config 5): first, a middle and the last of eight ranks."""
    g = synth.make_config("c5_1m")
    infos = [probe(g, r, 8) for r in (0, 3, 7)]
    assert len({(i.lanes_per_pose, i.lanes_per_lm, i.n_pose) for i in infos}) == 1
    assert infos[0].lm_first == 0 and infos[-1].lm_last == g.n_landmarks and infos[-1].pose_last == g.n_poses
    share = np.array([i.n_lm_edges_local for i in infos], float)
    assert share.max() / share.min() < 1.02


def test_twin_python_rules_reproduce_the_reference_python_optimizer():
    """rules="python": the loop of python/optimizer/graph_optimizer.py:20-92 (lambda * I damping, step lr, b zeroed at fixed
    vertices).  Pinned by the 10-iteration trajectory the reference's own GraphOptimizer.optimize(10, lr=.2) produced
    (tests/golden/c1_pyopt.npz) — through the Schur complement and multigrid PCG instead of scipy.linalg.solve."""
    z = util.load("c1_pyopt.npz")
    g = util.c1_arrays()
    for precond in ("amg", "jacobi"):
        r = oracle.sparse_optimize(util.to_oracle(g), 10, pcg_tol=1e-13, precond=precond, rules="python", lr=0.2)
        np.testing.assert_allclose(r["chi2"], z["chi2"], rtol=1e-10)
        assert util.max_vertex_diff(r["v_pos"], z["v_pos"], g.v_type) < 1e-8
    # lr = 1 (the Python signature's default) against the dense restatement of the same loop
    rd = oracle.optimize(util.to_oracle(g), 6, mode="python", solver="chol", lr=1.0)
    r1 = oracle.sparse_optimize(util.to_oracle(g), 6, pcg_tol=1e-13, precond="amg", rules="python", lr=1.0)
    assert r1["iters"] == rd["iters"] and r1["stop"] == rd["stop"]
    np.testing.assert_allclose(r1["chi2"], rd["chi2"], rtol=1e-9)


def test_layout_and_patterns_do_not_depend_on_the_host_thread_count():
    """The slot fill (host/problem.cpp) and the pattern builder (host/amg.cpp) split their work over host threads; the k-th
    edge of a vertex in input order must still take the vertex's k-th slot and every list keep its order, or summation
    orders on the device — and with them the bits of every result — would depend on the machine.  tsgo_amg_info.checksum
    hashes every table, numbering, pattern and gather list: 330 k edges (several fill chunks), 1 / 2 / 5 / 16 threads, whole
    graph and one shard of three."""
    import os
    g = synth.make(30000, 10, loop_closures=200, seed=9)
    lib = _lib.host_lib(); cg = g.c_struct()
    saved = os.environ.get("TSGO_HOST_THREADS")
    sums = {}
    try:
        for threads in ("1", "2", "5", "16"):
            os.environ["TSGO_HOST_THREADS"] = threads
            for rank, world in ((0, 1), (1, 3)):
                info = _lib.tsgo_amg_info(); od = C.c_int64()
                _lib.check(lib, lib.tsgo_amg_probe_shard(C.byref(cg), rank, world, C.byref(info), C.byref(od)), "tsgo_amg_probe_shard")
                sums.setdefault((rank, world), set()).add(info.checksum)
    finally:
        if saved is None:
            os.environ.pop("TSGO_HOST_THREADS", None)
        else:
            os.environ["TSGO_HOST_THREADS"] = saved
    assert all(len(v) == 1 for v in sums.values()), sums
    assert sums[(0, 1)] != sums[(1, 3)]

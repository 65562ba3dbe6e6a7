"""The edge-sharded DEVICE path with 2 and 3 ranks on the one GPU a box has (`-m gpu`).  RCCL refuses two ranks on one device,
so the handles of this process (one thread each) are joined by the library's in-process all-reduce group
(tsgo_comm_init_local: same places, same buffers as the RCCL calls, through host memory).  What this covers that a one-rank
communicator cannot: per-shard slot tables, which rank applies a pose's diagonal block / gauge / damping, rank 0 alone
contributing the level-0 diagonal, per-rank level-0 contribution lists, partial products and partial dots."""
import threading

import numpy as np
import pytest

from oracle import oracle
from tests import util
from toyslam_amd import synth
from toyslam_amd.optimizer import HipOptimizer, free_local_group, local_group

pytestmark = pytest.mark.gpu


def _run_sharded(g, world, iterations, **kw):
    group = local_group(world)
    out, errs = [None] * world, []

    def rank_main(rank):
        try:
            o = HipOptimizer(rank=rank, world=world, testing=True, **kw)
            try:
                o.comm_init_local(group)
                o.set_graph(g)
                r = o.optimize(iterations)
                out[rank] = (r, o.vertices())
            finally:
                o.close()
        except Exception as e:                       # noqa: BLE001 - reported by the caller
            errs.append((rank, repr(e)))

    th = [threading.Thread(target=rank_main, args=(k,), daemon=True) for k in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs, errs
    assert not any(t.is_alive() for t in th), "a rank is stuck in the in-process all-reduce (another one left early?)"
    free_local_group(group)
    return out


def _merge_landmarks(g, outs):
    """Landmarks are shard-local: every rank returns its own; poses are replicated."""
    v = outs[0][1].copy()
    moved = np.zeros(len(g.v_id), bool)
    for _, vr in outs:
        m = np.any(vr != g.v_pos, axis=1) & (g.v_type == 1)
        assert not np.any(moved & m), "a landmark was updated by two ranks"
        v[m] = vr[m]; moved |= m
    return v


@pytest.mark.parametrize("world,precond", [(2, "amg"), (3, "amg"), (2, "jacobi")])
def test_sharded_device_run_matches_the_single_handle_run(world, precond):
    g = synth.make(6000, 10, loop_closures=40, seed=13)
    g.fixed = np.array([0, int(g.v_id[4000]), int(g.v_id[-5])], np.uint32)       # a fixed pose in another shard's range, a fixed landmark
    single = HipOptimizer(pcg_rel_tol=1e-12, preconditioner=precond)
    try:
        single.set_graph(g); rs = single.optimize(5); vs = single.vertices()
    finally:
        single.close()
    outs = _run_sharded(g, world, 5, pcg_rel_tol=1e-12, preconditioner=precond)
    for r, _ in outs:
        np.testing.assert_allclose(r["chi2"], rs["chi2"], rtol=1e-10)
        np.testing.assert_array_equal(r["chi2"], outs[0][0]["chi2"])            # the ranks agree bit for bit ...
        np.testing.assert_array_equal(r["cg_iters"], outs[0][0]["cg_iters"])    # ... and take the same decisions
        assert r["fallbacks"] == 0
        assert abs(r["delta_norm"] - rs["delta_norm"]) <= 1e-9 * rs["delta_norm"]
        if precond == "amg":
            assert r["cg_iters"].max() < 60 and np.all(np.abs(r["cg_iters"] - rs["cg_iters"]) <= 3), (r["cg_iters"], rs["cg_iters"])
    v = _merge_landmarks(g, outs)
    assert util.max_vertex_diff(v, vs, g.v_type) < 1e-8
    ref = oracle.sparse_optimize(util.to_oracle(g), 5, pcg_tol=1e-12, precond="amg")
    assert util.max_vertex_diff(v, ref["v_pos"], g.v_type) < 1e-7


def test_sharded_device_run_with_python_rules_and_analytic_jacobians():
    g = synth.make(4000, 8, loop_closures=60, seed=17)
    oracle.set_odom_jacobian("analytic")
    try:
        ref = oracle.sparse_optimize(util.to_oracle(g), 4, pcg_tol=1e-12, precond="amg", rules="python", lr=0.6)
    finally:
        oracle.set_odom_jacobian("constant")
    outs = _run_sharded(g, 2, 4, pcg_rel_tol=1e-12, rules="python", lr=0.6, odom_jacobian="analytic")
    for r, _ in outs:
        np.testing.assert_allclose(r["chi2"], ref["chi2"], rtol=1e-9)
    assert util.max_vertex_diff(_merge_landmarks(g, outs), ref["v_pos"], g.v_type) < 1e-7


def test_bench_probe_sequence_of_a_sharded_run_does_not_deadlock():
    """bench.py --gpus N (N > 1) runs, on EVERY rank, timed steps, the kernel probes (each linearises, i.e. all-reduces), the
    coarse-level probe, the whole-iteration and setup probes, a second set_graph (structure reuse) and the 50-iteration
    convergence run.  The same sequence with three in-process ranks: every rank must come back, with the same numbers where
    they are replicated."""
    g = synth.make(5000, 10, seed=19)
    world = 3
    group = local_group(world)
    res, errs = [None] * world, []

    def rank_main(rank):
        try:
            o = HipOptimizer(rank=rank, world=world, pcg_rel_tol=1e-10, testing=True)
            try:
                o.comm_init_local(group)
                assert o.comm_selftest() == world                                   # bench.py's order: self-test and all-reduce timing before the graph is set
                ar_us = [o.comm_time_allreduce(n, reps=3) for n in (3 * 5000 + 40, 18 * 5000 + 40, 1 << 20)]
                assert min(ar_us) > 0, ar_us
                o.set_graph(g)
                chi = [o.optimize(1)["chi2"][0] for _ in range(3)]
                probes = [o.time_kernel(w, reps=5)[0] for w in (0, 1, 2, 3, 4)]
                levels = o.level_sweep_times(reps=5)
                it_us = o.time_kernel(5, reps=3)[0]; setup_us = o.time_kernel(6, reps=2)[0]
                o.set_graph(g)
                r = o.optimize(8)
                res[rank] = (chi, r["chi2"], r["structure_reused"], len(levels), it_us > 0 and setup_us > 0 and min(probes) > 0)
            finally:
                o.close()
        except Exception as e:                       # noqa: BLE001
            errs.append((rank, repr(e)))

    th = [threading.Thread(target=rank_main, args=(k,), daemon=True) for k in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs, errs
    assert not any(t.is_alive() for t in th), "deadlock: a rank did not come back"
    free_local_group(group)
    for k in range(1, world):
        assert res[k][0] == res[0][0] and np.array_equal(res[k][1], res[0][1])
    assert all(r[2] and r[3] >= 1 and r[4] for r in res)


@pytest.mark.parametrize("case", ["pose_graph_without_landmarks", "near_optimum", "fixed_landmark_and_duplicate_fixed_ids", "self_loop_and_duplicate_edges", "isolated_vertices"])
def test_sharded_stop_rules_and_degenerate_shards(case):
    """Every way out of the loop, with two ranks: 'worse' (the landmark-norm reduction is then owed after the loop), 'converged'
    (the reduction that decides it runs only when the pose part is already small), the cap; shards that own no landmark at
    all; a fixed landmark; duplicate edges.  Against the dense cpu/eigen restatement."""
    from tests import edge_cases
    g = getattr(edge_cases, case)()
    ref = oracle.optimize(util.to_oracle(g), 30, mode="cpp", solver="qr")        # rank-revealing, like the reference (isolated vertices make H singular)
    outs = _run_sharded(g, 2, 30, pcg_rel_tol=1e-12)
    for r, _ in outs:
        assert (r["iters"], r["stop"]) == (ref["iters"], ref["stop"]), (case, r["stop"], r["iters"], ref["stop"], ref["iters"])
        np.testing.assert_allclose(r["chi2"], ref["chi2"], rtol=1e-8)
        assert abs(r["delta_norm"] - ref["delta_norm"]) <= 1e-7 * max(1.0, ref["delta_norm"])
    scale = max(1.0, float(np.abs(ref["v_pos"]).max()))
    assert util.max_vertex_diff(_merge_landmarks(g, outs), ref["v_pos"], g.v_type) < 1e-7 * scale


def test_shards_whose_own_mean_degrees_straddle_a_lane_threshold():
    """Two shards with 5.998 and 6.001 LM edges per pose (tests/test_layout_and_twin.py::test_every_shard_chooses_the_same_lanes_per_pose):
    with per-shard lane choices the ranks' all-reduce buffers had different lengths."""
    g = synth.make(2077, 12, loop_closures=12, seed=741807)
    g.fixed = np.array([0, 3754], np.uint32)
    kw = dict(pcg_rel_tol=1e-12, odom_jacobian="analytic")
    single = HipOptimizer(**kw)
    try:
        single.set_graph(g); rs = single.optimize(6); vs = single.vertices()
    finally:
        single.close()
    outs = _run_sharded(g, 2, 6, **kw)
    for r, _ in outs:
        np.testing.assert_allclose(r["chi2"], rs["chi2"], rtol=1e-9)
        np.testing.assert_array_equal(r["cg_iters"], outs[0][0]["cg_iters"])
    assert util.max_vertex_diff(_merge_landmarks(g, outs), vs, g.v_type) < 1e-8


def test_one_all_reduce_per_iteration_variant_on_the_device():
    """What `bench.py --gpus N` runs (tsgo_config.cycle_level0 = 1): in-cycle products on the replicated explicit level-0 matrix.
    Three ranks against a single handle with the switch, and against the default form."""
    g = synth.make(6000, 10, loop_closures=40, seed=13)
    g.fixed = np.array([0, int(g.v_id[4000]), int(g.v_id[-5])], np.uint32)
    plain = HipOptimizer(pcg_rel_tol=1e-12)
    try:
        plain.set_graph(g); rp = plain.optimize(5); vp = plain.vertices()
    finally:
        plain.close()
    single = HipOptimizer(pcg_rel_tol=1e-12, cycle_level0="explicit")
    try:
        single.set_graph(g); rs = single.optimize(5); vs = single.vertices()
    finally:
        single.close()
    assert not np.array_equal(rs["cg_iters"], rp["cg_iters"])                    # the switch did change the preconditioner ...
    np.testing.assert_allclose(rs["chi2"], rp["chi2"], rtol=1e-10)               # ... and not the answer
    assert util.max_vertex_diff(vs, vp, g.v_type) < 1e-8
    outs = _run_sharded(g, 3, 5, pcg_rel_tol=1e-12, cycle_level0="explicit")
    for r, _ in outs:
        np.testing.assert_allclose(r["chi2"], rs["chi2"], rtol=1e-10)
        np.testing.assert_array_equal(r["chi2"], outs[0][0]["chi2"])
        np.testing.assert_array_equal(r["cg_iters"], outs[0][0]["cg_iters"])
        assert r["fallbacks"] == 0 and np.all(np.abs(r["cg_iters"] - rs["cg_iters"]) <= 3), (r["cg_iters"], rs["cg_iters"])
    assert util.max_vertex_diff(_merge_landmarks(g, outs), vs, g.v_type) < 1e-8


_C3 = {}


def _c3_single(cycle):
    """Config 3 (100k poses / 199k landmarks / 1.1M edges) through ONE handle: the reference every sharded run below is held to."""
    if cycle not in _C3:
        if "g" not in _C3:
            _C3["g"] = synth.make_config("c3_100k")
        o = HipOptimizer(pcg_rel_tol=1e-12, cycle_level0=cycle)
        try:
            o.set_graph(_C3["g"]); r = o.optimize(6); v = o.vertices()
        finally:
            o.close()
        _C3[cycle] = (r, v)
    return _C3["g"], _C3[cycle][0], _C3[cycle][1]


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("cycle", ["implicit", "explicit"])
def test_config_4_at_its_size_through_the_sharded_device_path(world, cycle):
    """BASELINE config 4: the 100k-pose / 1M-LM-edge graph edge-sharded over 2 and over 8 ranks — here all on one GPU through the
    in-process all-reduce group: 100k-pose shard tables, the 14.4 MB pose-partial and 2.4 MB product all-reduces, the 60 MB
    level-0 all-reduce, eight ranks deciding alike — in both forms of the cycle (`--implicit-cycle` and the explicit-level-0 one
    bench.py --gpus N runs).  Against the single-handle run of the same form (PCG tolerance 1e-12, so that what is compared is the
    sharded arithmetic and not two solves' tolerances): same stop, PCG counts within 2, chi^2 to 1e-10, vertices to 1e-9; ranks
    bit-identical."""
    g, rs, vs = _c3_single(cycle)
    outs = _run_sharded(g, world, 6, pcg_rel_tol=1e-12, cycle_level0=cycle)
    for r, _ in outs:
        assert (r["iters"], r["stop"]) == (rs["iters"], rs["stop"])
        np.testing.assert_allclose(r["chi2"], rs["chi2"], rtol=1e-10)
        np.testing.assert_array_equal(r["chi2"], outs[0][0]["chi2"])
        np.testing.assert_array_equal(r["cg_iters"], outs[0][0]["cg_iters"])
        assert r["fallbacks"] == 0
        assert np.all(np.abs(r["cg_iters"] - rs["cg_iters"]) <= 2), (r["cg_iters"], rs["cg_iters"])
        assert abs(r["delta_norm"] - rs["delta_norm"]) <= 1e-9 * rs["delta_norm"]
    assert util.max_vertex_diff(_merge_landmarks(g, outs), vs, g.v_type) < 1e-9

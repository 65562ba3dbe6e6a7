"""The edge-sharded algorithm across CPU processes (gloo): product shard planner (+ the product's sharded multigrid
pattern builder, host/amg.cpp: build_amg_sharded) + the CPU twin with an all-reduce hook in exactly the places where
the HIP path calls RCCL (tsgo_hip.hip: after lin_pose, after every schur_pose — three per multigrid-preconditioned
PCG iteration —, after the level-0 blocks of a hierarchy build, after the landmark update).  Compared with the
unsharded run, with both preconditioners."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _passive_openmp(monkeypatch):
    """Ranks waiting in a gloo all-reduce must not spin (libgomp's default wait policy) on the CPUs the other ranks need: the spawned
    workers inherit this."""
    monkeypatch.setenv("OMP_WAIT_POLICY", "PASSIVE")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _graph(n_poses, vlm):
    from tests import util
    from toyslam_amd import synth
    g = synth.make(n_poses, 10, loop_closures=30, seed=11)
    return util.with_virtual_landmarks(g, 0.4, seed=3) if vlm else g      # round 4: virtual landmark measurements (edge type 2) ride in the ODOM rows a shard owns


def _worker(rank, world, port, out_dir, n_poses, precond, rules, explicit_cycle=False, vlm=False):
    sys.path.insert(0, ROOT)
    torch.set_num_threads(2 if world <= 3 else 1)      # also sizes the twin's OpenMP loops (same libgomp); eight ranks share this box's eight CPUs
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    from tests import util
    from toyslam_amd import synth
    oracle.set_cycle_level0("explicit" if explicit_cycle else "implicit")       # the twin's side of tsgo_config.cycle_level0
    g = _graph(n_poses, vlm)
    calls = [0]

    def allreduce(buf):
        calls[0] += 1
        dist.all_reduce(torch.from_numpy(buf), op=dist.ReduceOp.SUM)

    r = oracle.sparse_optimize(util.to_oracle(g), 3, pcg_tol=1e-12, rank=rank, world=world, allreduce=allreduce, precond=precond, rules=rules, lr=0.5)
    # landmarks are shard-local: owned ones moved, the others still hold their input value
    moved = np.any(r["v_pos"] != g.v_pos, axis=1) & (g.v_type == 1)
    lm = np.where(moved[:, None], r["v_pos"], 0.0)
    t = torch.from_numpy(lm.copy()); dist.all_reduce(t)
    cnt = torch.from_numpy(moved.astype(np.float64)); dist.all_reduce(cnt)
    v = np.where((g.v_type == 1)[:, None], t.numpy(), r["v_pos"])
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), chi2=r["chi2"], v=v, cg=r["cg_iters"], calls=calls[0],
             owners=cnt.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,precond,n_poses,rules,vlm", [(2, "jacobi", 300, "cpp", False), (2, "amg", 1500, "cpp", False), (3, "amg", 700, "cpp", False), (2, "amg", 600, "python", False),
                                                             (8, "amg", 2400, "cpp", False), (2, "amg", 900, "cpp", True)])
def test_sharded_run_across_processes_matches_single_process(tmp_path, world, precond, n_poses, rules, vlm):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), n_poses, precond, rules, False, vlm), nprocs=world, join=True)
    from oracle import oracle
    from tests import util
    g = _graph(n_poses, vlm)
    ref = oracle.sparse_optimize(util.to_oracle(g), 3, pcg_tol=1e-12, precond=precond, rules=rules, lr=0.5)
    outs = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for o in outs:
        np.testing.assert_allclose(o["chi2"], ref["chi2"], rtol=1e-11)
        assert util.max_vertex_diff(o["v"], ref["v_pos"], g.v_type) < 1e-9
        assert o["calls"] > 10                       # the hook really carried the reduction
        assert np.all(o["owners"][g.v_type == 1] <= 1)   # no landmark is owned twice
        if precond == "amg":                         # multigrid iteration counts, not block-Jacobi ones, and the same
            assert o["cg"].max() < 60, o["cg"]       # work as the unsharded multigrid run (+-2: the level-0 sums differ in order)
            assert np.all(np.abs(o["cg"] - ref["cg_iters"]) <= 2), (o["cg"], ref["cg_iters"])
    for o in outs[1:]:
        np.testing.assert_array_equal(outs[0]["chi2"], o["chi2"])      # ranks agree bit for bit
        np.testing.assert_array_equal(outs[0]["cg"], o["cg"])          # ... and take the same decisions (or RCCL would hang)


def test_one_all_reduce_per_iteration_variant_across_processes(tmp_path):
    """What `bench.py --gpus N` runs (tsgo_config.cycle_level0 = 1): the two products inside the multigrid cycle read the replicated
    explicit level-0 matrix, so only the PCG's own product is all-reduced.  Two processes against one, both with the switch
    , and against the default form's answer; the hook is called far less."""
    one = tmp_path / "one"; two = tmp_path / "two"; plain = tmp_path / "plain"
    for d in (one, two, plain):
        d.mkdir()
    mp.spawn(_worker, args=(1, _free_port(), str(one), 1500, "amg", "cpp", True), nprocs=1, join=True)
    mp.spawn(_worker, args=(2, _free_port(), str(two), 1500, "amg", "cpp", True), nprocs=2, join=True)
    mp.spawn(_worker, args=(2, _free_port(), str(plain), 1500, "amg", "cpp", False), nprocs=2, join=True)
    ref = np.load(os.path.join(str(one), "rank0.npz"))
    outs = [np.load(os.path.join(str(two), "rank%d.npz" % r)) for r in range(2)]
    base = np.load(os.path.join(str(plain), "rank0.npz"))
    from toyslam_amd import synth
    from tests import util
    g = synth.make(1500, 10, loop_closures=30, seed=11)
    for o in outs:
        np.testing.assert_allclose(o["chi2"], ref["chi2"], rtol=1e-11)
        assert util.max_vertex_diff(o["v"], ref["v"], g.v_type) < 1e-9
        assert np.all(np.abs(o["cg"] - ref["cg"]) <= 2), (o["cg"], ref["cg"])
        np.testing.assert_allclose(o["chi2"], base["chi2"], rtol=1e-9)           # the same answer as the default form
        assert o["cg"].max() < 80
    np.testing.assert_array_equal(outs[0]["chi2"], outs[1]["chi2"])
    np.testing.assert_array_equal(outs[0]["cg"], outs[1]["cg"])
    # per PCG iteration one all-reduce instead of three
    per_it = float(outs[0]["calls"]) / float(outs[0]["cg"].sum()); per_it_plain = float(base["calls"]) / float(base["cg"].sum())
    assert per_it < 1.6 and per_it_plain > 2.8, (per_it, per_it_plain)

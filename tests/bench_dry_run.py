"""`bench.py --gpus N --dry-run-cpu`: a rehearsal of bench.py's own N-rank launch path on a box without GPUs.

The parent starts `torch.distributed.run` exactly as for a measured run; every rank lands here instead of on the device path:
process group over gloo, the product's shard planner + the CPU twin (oracle/oracle_sparse.cpp) with an all-reduce hook where the HIP
path calls RCCL, W warm-up + K timed steps between barriers, max over ranks, ONE JSON line from rank 0 — marked `dry_run`: the
number is the CPU twin's, it is NOT a measurement of the product and never `value` of a bench line the driver records.
Test infrastructure (it imports oracle/): lives under tests/, exercised by tests/test_bench_launch.py.
"""
import json
import os
import time

import numpy as np


def run(args, rank, world, watch):
    import datetime

    import torch
    import torch.distributed as dist
    from oracle import oracle
    from tests import util
    from toyslam_amd import synth

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.set_num_threads(2)
    oracle.set_threads(2)
    import sys
    with watch.phase("torch.distributed process group (gloo) + first barrier", args.phase_timeout):
        sys.stdout.flush()
        saved = os.dup(1); os.dup2(2, 1)          # gloo prints a connection banner on STDOUT; this script's stdout is ONE JSON line
        try:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=args.phase_timeout))
            dist.barrier()
        finally:
            sys.stdout.flush(); os.dup2(saved, 1); os.close(saved)
    stall = os.environ.get("TSGO_DRY_RUN_STALL_RANK")        # test hook of the watchdog: this rank never reaches the first all-reduce
    n_poses = 1500
    g = synth.make(n_poses, 10, loop_closures=0, seed=0)
    calls = [0]

    def allreduce(buf):
        calls[0] += 1
        dist.all_reduce(torch.from_numpy(buf), op=dist.ReduceOp.SUM)

    with watch.phase("first all-reduce on the data path", args.phase_timeout):
        if stall is not None and int(stall) == rank:
            time.sleep(10 * args.phase_timeout)
        probe = np.array([rank + 1.0])
        allreduce(probe)
        assert probe[0] == world * (world + 1) / 2, probe
    o = util.to_oracle(g)
    steps, warm = max(1, args.steps), max(0, args.warmup)
    with watch.phase("warm-up + timed steps (CPU twin)", 10 * args.phase_timeout):
        dist.barrier()
        t0 = time.perf_counter()
        r = oracle.sparse_optimize(o, warm + steps, pcg_tol=args.pcg_tol, rank=rank, world=world, allreduce=allreduce, precond=args.precond)
        dist.barrier()
        dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
    n = torch.tensor([float(calls[0])], dtype=torch.float64); dist.all_reduce(n, op=dist.ReduceOp.MAX)
    with watch.phase("final barrier", args.phase_timeout):
        dist.barrier()
        dist.destroy_process_group()
    watch.note("done")
    if rank == 0:
        n_it = int(r["iters"])
        print(json.dumps({
            "dry_run": "launch-path rehearsal on CPU: gloo + the CPU twin; NOT a measurement of the product",
            "metric": "edges/sec per GN iter", "value": None, "unit": "edges/s", "n_gpus": 0, "ranks": world, "steps": steps, "warmup": warm,
            "twin_edges_per_s": len(g.e_type) * n_it / float(t.item()), "twin_ms_per_step": 1e3 * float(t.item()) / n_it,
            "config": {"workload": "dry run: %d poses / %d edges, edge-sharded x%d over gloo" % (n_poses, len(g.e_type), world)},
            "all_reduce_calls": int(n.item()), "pcg_iters_per_gn_iter": float(np.mean(r["cg_iters"])), "chi2_first_last": [float(r["chi2"][0]), float(r["chi2"][-1])]}))

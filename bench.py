#!/usr/bin/env python3
"""bench.py — edges/s per Gauss-Newton iteration on the BASELINE.json workload.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3_100k|c2_10k|c5_1m] [--precision 64]
  N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE full Gauss-Newton iteration on the device-resident graph: linearise every edge
(residual, Jacobian, Huber, block accumulation), solve H delta = b by implicit-Schur PCG to the
configured tolerance, update every vertex.  Inputs are resident in HBM before the timed region
(tsgo_set_graph is outside it).  value = (ODOM + LM edges) * K / wall time of K steps, max over ranks.

N > 1, default: request-parallel — every rank optimises its OWN graph of the named configuration (seed =
rank), the way a multi-GPU graph_optimizer serves independent connections; no data-path collective,
"scaling": "weak".  `--shard` instead splits ONE graph by edge set across the ranks (landmark ranges
per rank, pose state replicated, RCCL all-reduce of the pose partials once per GN iteration and of the
Schur product once per PCG iteration; "scaling": "strong"; DESIGN.md section 5 says why that mode is
latency-bound at this size).

Extra objects on the JSON line: `roofline` (dominant kernel, algorithmic bytes / hipEvent time measured
here) and `cpu_baseline` (the CPU twin of the same math on the host cores; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
PMC_TRAFFIC = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")   # rocprofv3 --pmc digest of this same command


def pmc_traffic(kernel, workload, precision):
    """Memory-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE x2 +
    WRITE_SIZE, MI355X_MICROARCH.md HBM section), or None when no digest matches this configuration."""
    try:
        d = json.load(open(PMC_TRAFFIC))
        if d.get("workload") != workload or d.get("precision") != precision:
            return None
        # the instance tsgo_time_kernel launches: f64 slot planes (LOW = 0), product mode (MODE = 0)
        ending = {"k_schur_lm": ", 0, 0>", "k_schur_pose": ", 0>"}.get(kernel, ">")
        for k, v in d["kernels"].items():
            if k.startswith(kernel + "<") and k.endswith(ending):
                return v["hbm_bytes_corrected"]
    except (OSError, ValueError, KeyError):
        pass
    return None
KERNELS = {0: "k_schur_lm", 1: "k_schur_pose", 2: "k_cg_update", 3: "k_lin_lm", 4: "k_lin_pose"}


def cpu_baseline(g, threads):
    """One full GN iteration (iteration 0) of the SAME graph by the CPU twin (oracle/oracle_sparse.cpp):
    same layout, same Schur PCG, same tolerance.  kind = "port": the reference's own dense algorithm
    cannot run this size (O(n^2) memory)."""
    from oracle import oracle
    from tests import util
    oracle.set_threads(threads)
    o = util.to_oracle(g)
    t0 = time.time()
    n_it = max(1, ARGS.steps)
    r = oracle.sparse_optimize(o, n_it, pcg_tol=ARGS.pcg_tol, precond=ARGS.precond)
    dt = time.time() - t0
    host = dt - r["seconds_linearize"] - r["seconds_solve"]        # layout + multigrid patterns, built once
    per_iter = (r["seconds_linearize"] + r["seconds_solve"]) / r["iters"]
    return {"value": len(g.e_type) / per_iter, "unit": "edges/s per GN iter", "cores": threads, "kind": "port",
            "sample": "the first %d GN iterations of the same %s graph by the CPU twin of the same algorithm (%s-preconditioned "
                      "Schur PCG, tol %g, %.1f PCG iterations per GN iteration): %.1f s of CPU work + %.1f s one-time host setup (excluded, as on the GPU side)"
                      % (r["iters"], ARGS.workload, ARGS.precond, ARGS.pcg_tol, float(np.mean(r["cg_iters"])), dt - host, host),
            "pcg_iters_per_gn_iter": float(np.mean(r["cg_iters"])), "seconds_per_gn_iter": per_iter}


def main():
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != ARGS.gpus:
        if world == 1 and ARGS.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with that many processes" % ARGS.gpus)
    from toyslam_amd import synth
    from toyslam_amd.optimizer import HipOptimizer

    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    shard = ARGS.shard and world > 1
    g = synth.make_config(ARGS.workload, seed=0 if shard else rank)
    n_edges = len(g.e_type)
    opt = HipOptimizer(device=local_rank, precision=ARGS.precision, pcg_rel_tol=ARGS.pcg_tol,
                       rank=rank if shard else 0, world=world if shard else 1,
                       use_graphs=not ARGS.no_graphs, lanes_per_pose=ARGS.lanes_pose, lanes_per_lm=ARGS.lanes_lm,
                       preconditioner=ARGS.precond)
    if shard:
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid = torch.tensor(list(opt.comm_unique_id()), dtype=torch.uint8, device="cuda")
        dist.broadcast(uid, 0)
        opt.comm_init(bytes(uid.cpu().tolist()))
    opt.set_graph(g)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    chi2, cg = [], []
    ms_setup = None
    for _ in range(ARGS.warmup):
        r = opt.optimize(1); chi2 += list(r["chi2"]); cg += list(r["cg_iters"])
    barrier()
    t0 = time.perf_counter()
    ms_lin = ms_solve = ms_upd = 0.0
    for _ in range(ARGS.steps):
        r = opt.optimize(1); chi2 += list(r["chi2"]); cg += list(r["cg_iters"])
        ms_lin += r["ms_linearize"]; ms_solve += r["ms_solve"]; ms_upd += r["ms_update"]; ms_setup = r["ms_setup"]
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if not shard:           # request-parallel: the job processed every rank's edges
            e = torch.tensor([float(n_edges)], dtype=torch.float64, device="cuda")
            dist.all_reduce(e, op=dist.ReduceOp.SUM)
            n_edges = int(e.item())
    timed_cg = cg[ARGS.warmup:]

    out = None
    if rank == 0:
        # dominant kernel: the one with the largest share of a GN iteration
        n_cg = float(np.mean(timed_cg)) if timed_cg else 0.0
        shares = {}
        amg = ARGS.precond == "amg" and not shard
        for which in (0, 1, 2, 3, 4):
            us, nbytes = opt.time_kernel(which, reps=200)
            launches = {0: 3.0 if amg else 1.0, 1: 3.0 if amg else 1.0, 2: 0.0 if amg else 1.0}     # per PCG iteration
            per_iter = us * (n_cg * launches[which] if which < 3 else 1.0)
            shares[which] = (per_iter, us, nbytes)
        dom = max(shares, key=lambda k: shares[k][0])
        _, us, nbytes = shares[dom]
        achieved = nbytes / (us * 1e-6) / 1e9
        out = {
            "metric": "edges/sec per GN iter", "value": n_edges * ARGS.steps / dt, "unit": "edges/s",
            "n_gpus": world, "steps": ARGS.steps, "warmup": ARGS.warmup, "ms_per_step": 1e3 * dt / ARGS.steps,
            "higher_is_better": True, "scaling": "strong" if shard else "weak", "vs_baseline": None,
            "dtype": "f64" if ARGS.precision == 64 else "f32", "data": "synthetic",
            "config": {"workload": "%s: %d poses / %d landmarks / %d ODOM + %d LM edges, seeded synthetic 2-D SLAM graph"
                                   % (ARGS.workload, g.n_poses, g.n_landmarks, int((g.e_type == 0).sum()), int((g.e_type == 1).sum())),
                       "solver": "implicit-Schur PCG (Chronopoulos-Gear), %s, rel tol %g"
                                 % ("smoothed-aggregation multigrid V(1,1) preconditioner" if (ARGS.precond == "amg" and not shard) else "block-Jacobi on the Schur diagonal", ARGS.pcg_tol),
                       "parallelism": ("one graph edge-sharded x%d, RCCL all-reduce per PCG iteration" % world) if shard else
                                      ("request-parallel: %d independent graphs, one per GPU, no collective" % world if world > 1 else "single GPU"),
                       "hipgraph": not ARGS.no_graphs},
            "gn_iters_per_s": ARGS.steps / dt,
            "pcg_iters_per_gn_iter": n_cg,
            "chi2_first_last": [chi2[0], chi2[-1]],
            "ms_per_step_device": {"linearize": ms_lin / ARGS.steps, "solve": ms_solve / ARGS.steps, "update": ms_upd / ARGS.steps},
            "ms_setup_once_per_graph": ms_setup,
            "roofline": {"bound": "hbm", "kernel": KERNELS[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(KERNELS[dom], ARGS.workload, ARGS.precision), "us_per_launch": us, "algorithmic_bytes_per_launch": nbytes,
                         "all_kernels_us": {KERNELS[k]: shares[k][1] for k in shares},
                         "us_per_pcg_iteration": opt.time_kernel(5, reps=20)[0],
                         "us_multigrid_numeric_setup": opt.time_kernel(6, reps=5)[0] if amg else None},
        }
        if not shard and not ARGS.no_conv:      # request-parallel ranks run the same kind of graph: rank 0's own run stands for all
            # the second half of BASELINE.json's metric: Gauss-Newton iterations until the reference's plateau rule
            # |chi2_k - chi2_{k-1}| < 1e-3 fires (OptimizerCpu.h:167-171), capped at 50; a fresh run, outside the timed region
            opt.set_graph(g)
            rc = opt.optimize(50)
            out["iters_to_chi2_tol"] = {"iterations": int(rc["iters"]), "stop": rc["stop"], "cap": 50, "chi2_first": float(rc["chi2"][0]),
                                        "chi2_last": float(rc["chi2"][-1]), "pcg_iters_total": int(rc["cg_total"]),
                                        "seconds": rc["ms_total"] / 1e3, "pcg_fallbacks": int(rc["fallbacks"])}
        if world == 1 and not ARGS.no_cpu:
            out["cpu_baseline"] = cpu_baseline(g, ARGS.cpu_threads or min(16, len(os.sched_getaffinity(0))))
    opt.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3_100k")
    ap.add_argument("--precision", type=int, default=64)
    ap.add_argument("--pcg-tol", dest="pcg_tol", type=float, default=1e-10)
    ap.add_argument("--precond", default="amg", choices=["amg", "jacobi"])
    ap.add_argument("--shard", action="store_true", help="N > 1: split ONE graph by edge set (strong scaling)")
    ap.add_argument("--no-graphs", dest="no_graphs", action="store_true")
    ap.add_argument("--no-cpu", dest="no_cpu", action="store_true")
    ap.add_argument("--no-conv", dest="no_conv", action="store_true", help="skip the 50-iteration convergence run")
    ap.add_argument("--cpu-threads", dest="cpu_threads", type=int, default=0)
    ap.add_argument("--lanes-pose", dest="lanes_pose", type=int, default=0)
    ap.add_argument("--lanes-lm", dest="lanes_lm", type=int, default=0)
    ARGS = ap.parse_args()
    main()

#!/usr/bin/env python3
"""bench.py — edges/s per Gauss-Newton iteration on the BASELINE.json workload.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3_100k|c2_10k|c5_1m] [--precision 64]
  N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE full Gauss-Newton iteration on the device-resident graph: linearise every edge
(residual, Jacobian, Huber, block accumulation), solve H delta = b by implicit-Schur PCG to the
configured tolerance, update every vertex.  Inputs are resident in HBM before the timed region
(tsgo_set_graph is outside it).  value = (ODOM + LM edges) * K / wall time of K steps, max over ranks.

N = 1: BASELINE.json config 3 (100k poses / 1M LM edges on one MI355X).
N > 1 (default): BASELINE.json config 4 — the SAME one graph, edge-sharded across the N ranks ("scaling": "strong"):
every rank owns a contiguous landmark range with all its LM edges plus the ODOM rows / gauge terms of a contiguous pose
range; pose state, PCG vectors and the multigrid hierarchy are replicated.  RCCL all-reduces: the pose partials once per
GN iteration, the PCG's Schur product once per iteration (the two products inside the multigrid cycle read the replicated
explicit level-0 matrix; --implicit-cycle shards and all-reduces them too), the level-0 blocks once per hierarchy build.  DESIGN.md section 5 gives the expected 1 -> 8 curve and why this size is latency-bound.
`--request-parallel` instead gives every rank its OWN graph of the named configuration (what a multi-GPU
graph_optimizer does with independent connections; no data-path collective, "scaling": "weak").

Extra objects on the JSON line: `roofline` (the kernel with the largest time share: algorithmic bytes / hipEvent time
measured here; plus SURVEY 8d's byte model for the whole step and for one PCG iteration) and `cpu_baseline` (the CPU twin
of the same math on the host cores, and the reference's dense algorithm at config 1; rank 0, N = 1 only).
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
PMC_TRAFFIC = os.path.join("profiles", "r03_pmc_traffic.json")   # rocprofv3 --pmc digest of this same command (tools/pmc_traffic.py)
ROCPROF_STATS = os.path.join("profiles", "r03_rocprofv3_kernel_stats.csv")   # rocprofv3 --kernel-trace --stats of this same command
KERNELS = {0: "k_schur_lm", 1: "k_schur_pose", 2: "k_cg_update", 3: "k_lin_lm", 4: "k_lin_pose"}


def rocprof_average_us(symbol):
    """Average duration of `symbol` (kernel name without arguments, as rocprofv3 prints it) in the COMMITTED --stats CSV."""
    import csv
    try:
        for row in csv.DictReader(open(os.path.join(ROOT, ROCPROF_STATS))):
            name = row["Name"]
            name = name[len("void "):] if name.startswith("void ") else name
            name = name[len("tsgo::"):] if name.startswith("tsgo::") else name
            if name.split("(")[0] == symbol:
                return float(row["AverageNs"]) / 1e3
    except (OSError, ValueError, KeyError):
        pass
    return None


def pmc_traffic(kernel, workload, precision):
    """Memory-side bytes per launch of `kernel` from the COMMITTED rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
    MI355X_MICROARCH.md HBM section) — not measured in this run: PMC needs rocprofv3 around the process."""
    try:
        d = json.load(open(os.path.join(ROOT, PMC_TRAFFIC)))
        if d.get("workload") != workload or d.get("precision") != precision:
            return None
        k = d["kernels"].get(kernel, {})
        return k.get("hbm_bytes_mean", k.get("hbm_bytes_corrected"))
    except (OSError, ValueError, KeyError):
        return None


def survey_bytes(P, L, Eo, El, s):
    """algorithmic bytes of one linearisation, one PCG iteration, one update (ids are 4 B, scalars s B).
    f32 (s = 4) gives B_lin = 68 Eo + 48 El + 96 P + 48 L, B_cg = 36 Eo + 24 El + 192 P + 112 L, B_upd = 3 (12 P + 8 L)."""
    b_lin = Eo * (8 + 6 * s + 9 * s) + El * (8 + 4 * s + 6 * s) + P * (3 + 9 + 3 + 9) * s + L * (2 + 4 + 2 + 4) * s
    b_cg = Eo * 9 * s + El * 6 * s + P * (9 + 9 + 30) * s + L * (4 + 4 + 20) * s
    b_upd = 3 * (3 * s * P + 2 * s * L)
    return b_lin, b_cg, b_upd


def cpu_baseline(g, threads):
    """The SAME graph by the CPU twin (oracle/oracle_sparse.cpp): same layout, same Schur PCG, same tolerance.  kind = "port":
    the reference's own dense algorithm cannot run this size (O(n^2) memory); what it does at the one
    configuration it can run (config 1) is timed beside it: reference_dense_c1."""
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
    out = {"value": len(g.e_type) / per_iter, "unit": "edges/s per GN iter", "cores": threads, "kind": "port",
           "sample": "the first %d GN iterations of the same %s graph by the CPU twin of the same algorithm (%s-preconditioned "
                     "Schur PCG, tol %g, %.1f PCG iterations per GN iteration): %.1f s of CPU work + %.1f s one-time host setup (excluded, as on the GPU side)"
                     % (r["iters"], ARGS.workload, ARGS.precond, ARGS.pcg_tol, float(np.mean(r["cg_iters"])), dt - host, host),
           "pcg_iters_per_gn_iter": float(np.mean(r["cg_iters"])), "seconds_per_gn_iter": per_iter}
    # the reference's algorithm itself (dense H, column-pivoted Householder QR, float like remote/app/main.cpp:40, ONE thread:
    # the reference's thread pool is disabled, OptimizerCpu.h:78,125-130) at the only BASELINE configuration it can hold
    oracle.set_threads(1)
    c1 = util.to_oracle(util.c1_arrays())
    t0 = time.time()
    rd = oracle.optimize(c1, 3, mode="cpp", solver="qr", precision="f32")
    dd = time.time() - t0
    out["reference_dense_c1"] = {"kind": "reference algorithm, restated (oracle/oracle_dense.cpp); Eigen itself is not in the image",
                                 "workload": "config 1: 150 poses / 342 landmarks / 2123 edges, n = 1134", "dtype": "f32", "solver": "dense column-pivoted QR",
                                 "cores": 1, "gn_iterations": int(rd["iters"]), "ms_per_gn_iter": 1e3 * dd / rd["iters"],
                                 "edges_per_s": len(c1.e_type) * rd["iters"] / dd,
                                 "note": "dense H is n^2: 39 GB at config 2, 3.9 TB at config 3 — this algorithm cannot run the benchmarked size"}
    oracle.set_threads(threads)
    return out


class stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when a communicator is created; this script's stdout is ONE JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


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
        with stdout_to_stderr():
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            dist.barrier()                           # the communicator (and its banner) is created lazily: now

    shard = world > 1 and not ARGS.request_parallel
    # Sharded runs take the two products INSIDE the multigrid cycle from the replicated explicit level-0 matrix
    # (tsgo_config.cycle_level0 = 1): one all-reduce per PCG iteration instead of three for 6 % more iterations — ahead as soon as an
    # all-reduce costs more than 19 us (profiles/r02e_explicit_level0_in_cycle.txt).  --implicit-cycle keeps the single-device form.
    explicit_cycle = (shard and not ARGS.implicit_cycle) or ARGS.explicit_cycle
    g = synth.make_config(ARGS.workload, seed=0 if shard else rank)
    n_edges = len(g.e_type)
    opt = HipOptimizer(device=local_rank, precision=ARGS.precision, pcg_rel_tol=ARGS.pcg_tol,
                       rank=rank if shard else 0, world=world if shard else 1,
                       use_graphs=(False if ARGS.no_graphs else (True if ARGS.graphs else "auto")), lanes_per_pose=ARGS.lanes_pose, lanes_per_lm=ARGS.lanes_lm,
                       preconditioner=ARGS.precond, warm_start=ARGS.warm_start, cycle_level0="explicit" if explicit_cycle else "implicit", cycle_storage=ARGS.cycle_storage)
    if ARGS.force_collective and world == 1:          # research: the sharded code path (eager launches + RCCL calls) with a one-rank communicator
        with stdout_to_stderr():
            opt.comm_init(opt.comm_unique_id())
    if shard:
        with stdout_to_stderr():
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
    graph_replay = False
    for _ in range(ARGS.steps):
        r = opt.optimize(1); chi2 += list(r["chi2"]); cg += list(r["cg_iters"])
        ms_lin += r["ms_linearize"]; ms_solve += r["ms_solve"]; ms_upd += r["ms_update"]; ms_setup = r["ms_setup"]; graph_replay = graph_replay or r["graph_replay"]
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
    # Probes after the timed region.  In a sharded run every probe that linearises or iterates contains collectives, so
    # EVERY rank runs them (rank 0 alone would wait for the others forever); request-parallel ranks are independent.
    if rank == 0 or shard:
        n_cg = float(np.mean(timed_cg)) if timed_cg else 0.0
        amg = ARGS.precond == "amg"
        # In situ: one PCG iteration launched eagerly with a hipEvent before every kernel, on the engine's stream, averaged over 20
        # iterations (tsgo_profile_iteration) — every kernel in the cache state the solve leaves it in.  An event between two
        # kernels costs a little itself: the sum of the marks against the same iteration timed without them gives that cost per
        # launch, which is taken off every entry ("us_in_situ"; "us_in_situ_raw" is the interval as recorded).
        prof = opt.profile_iteration(reps=20)
        us_pcg = opt.time_kernel(5, reps=20)[0]
        overhead = max(0.0, (sum(e["us"] for e in prof) - us_pcg) / max(1, len(prof)))
        shares = {}
        for e in prof:
            k = shares.setdefault(e["name"], {"launches_per_iteration": 0, "us_raw": 0.0, "bytes": 0.0, "where": []})
            k["launches_per_iteration"] += 1; k["us_raw"] += e["us"]; k["bytes"] += e["bytes"]; k["where"].append(e["where"])
        for name, k in shares.items():
            n = k["launches_per_iteration"]
            k["us_in_situ_raw"] = k.pop("us_raw") / n
            k["us_in_situ"] = max(k["us_in_situ_raw"] - overhead, 1e-3)
            k["algorithmic_bytes_per_launch"] = k.pop("bytes") / n
            k["launches_per_step"] = n_cg * n
            k["us_per_step"] = k["us_in_situ"] * n_cg * n
            k["us_rocprof_stats"] = rocprof_average_us(name)
            k["achieved_GBps"] = k["algorithmic_bytes_per_launch"] / (k["us_in_situ"] * 1e-6) / 1e9
            k["frac_of_hbm_peak"] = k["achieved_GBps"] / HBM_PEAK_GBS
        # back-to-back figures of the table kernels (200 launches of one kernel in a row: warm L2 / MALL), for comparison
        tname = "double" if ARGS.precision == 64 else "float"
        for which in (0, 1, 2, 3, 4):
            us, nbytes = opt.time_kernel(which, reps=200)
            # the instantiation tsgo_time_kernel launches: f64 / f32 slot planes in T, product mode (k_schur_lm<T, G, 0, 0>, k_schur_pose<T, G, 0, OJ>)
            pat = {0: r"k_schur_lm<%s, \d+, 0, 0>$", 1: r"k_schur_pose<%s, \d+, 0, \d>$"}.get(which, KERNELS[which] + r"<%s[,>]")
            hit = [n for n in shares if re.match(pat % tname, n)]
            if hit:
                shares[hit[0]]["us_back_to_back"] = us
            elif which >= 3:      # the two linearisation kernels run once per step, outside the PCG iteration
                shares[KERNELS[which]] = {"launches_per_iteration": 0, "us_back_to_back": us, "us_in_situ": us, "algorithmic_bytes_per_launch": nbytes,
                                          "launches_per_step": 1.0, "us_per_step": us, "us_rocprof_stats": None, "achieved_GBps": nbytes / (us * 1e-6) / 1e9,
                                          "frac_of_hbm_peak": nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "where": ["linearisation (timed back to back)"]}
        # What bounds a launch at this size: a floor of dependent round trips (arguments -> row bounds -> indices -> gathered operand ->
        # store: the shortest kernels of the iteration, < 0.5 MB, take 3.5-4 us) plus the bytes the memory side really moves (PMC
        # where a committed digest has the symbol, else the algorithmic figure) at the rate an MI355X streams in practice
        # (MI355X_MICROARCH.md: 6.29 TB/s copy; 6.0 used).  us_latency_model / us_in_situ near 1 = the kernel is on that line.
        FLOOR_US, STREAM_GBS = 3.7, 6000.0
        for name, k in shares.items():
            if not k["launches_per_iteration"]:
                continue
            moved = pmc_traffic(name, ARGS.workload, ARGS.precision) or k["algorithmic_bytes_per_launch"]
            k["us_latency_model"] = FLOOR_US + moved / (STREAM_GBS * 1e9) * 1e6
        model_iter = sum(k["us_latency_model"] * k["launches_per_iteration"] for k in shares.values() if k.get("us_latency_model"))
        us_setup = opt.time_kernel(6, reps=5)[0] if amg else None
        conv = None
        if not ARGS.no_conv:
            # the second half of BASELINE.json's metric: Gauss-Newton iterations until the reference's plateau rule
            # |chi2_k - chi2_{k-1}| < 1e-3 fires (OptimizerCpu.h:167-171), capped at 50; a fresh run, outside the timed region
            opt.set_graph(g)
            rc = opt.optimize(50)
            def conv_of(rc):
                return {"iterations": int(rc["iters"]), "stop": rc["stop"], "cap": 50, "chi2_first": float(rc["chi2"][0]),
                        "chi2_last": float(rc["chi2"][-1]), "pcg_iters_total": int(rc["cg_total"]),
                        "seconds": rc["ms_total"] / 1e3, "pcg_fallbacks": int(rc["fallbacks"])}
            conv = conv_of(rc)
            if conv["stop"] == "cap":
                conv["note"] = "the plateau rule |chi2_k - chi2_(k-1)| < 1e-3 does not fire within the cap at this size (chi2 is still falling by more than 1e-3 per 0.2-damped step): the figure is the cap; see other_configs for the configurations where it does fire"
            if world == 1 and ARGS.workload == "c3_100k":
                # ... and where the rule does fire: config 1 (the reference's own graph; plateau at iteration 41 under cpu/eigen too,
                # tests/golden/c1_cpprules_ref.npz) and config 2
                from toyslam_amd.graph import GraphArrays
                other = {}
                z = np.load(os.path.join(ROOT, "tests", "golden", "c1_graph.npz"))
                c1 = GraphArrays(z["v_id"], z["v_type"], z["v_pos"], z["e_type"], z["e_ids"], z["e_meas"], z["e_inf"], z["fixed"]).rounded_to_wire()
                for name, gg in (("c1_reference_sim", c1), ("c2_10k", synth.make_config("c2_10k"))):
                    opt.set_graph(gg)
                    other[name] = conv_of(opt.optimize(50))
                conv["other_configs"] = other
                opt.set_graph(g)
    if rank == 0:
        dom = max(shares, key=lambda k: shares[k]["us_per_step"])
        us, nbytes = shares[dom]["us_in_situ"], shares[dom]["algorithmic_bytes_per_launch"]
        achieved = nbytes / (us * 1e-6) / 1e9
        s = 8 if ARGS.precision == 64 else 4
        Eo, El = int((g.e_type == 0).sum()), int((g.e_type == 1).sum())
        b_lin, b_cg, b_upd = survey_bytes(g.n_poses, g.n_landmarks, Eo, El, s)
        b_gn = b_lin + n_cg * b_cg + b_upd
        ms_step = 1e3 * dt / ARGS.steps
        traffic = pmc_traffic(dom, ARGS.workload, ARGS.precision)
        out = {
            "metric": "edges/sec per GN iter", "value": n_edges * ARGS.steps / dt, "unit": "edges/s",
            "n_gpus": world, "steps": ARGS.steps, "warmup": ARGS.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong" if shard or world == 1 else "weak", "vs_baseline": None,
            "dtype": "f64" if ARGS.precision == 64 else "f32", "data": "synthetic",
            "config": {"workload": "%s: %d poses / %d landmarks / %d ODOM + %d LM edges, seeded synthetic 2-D SLAM graph"
                                   % (ARGS.workload, g.n_poses, g.n_landmarks, Eo, El),
                       "solver": "implicit-Schur PCG (Chronopoulos-Gear), %s, rel tol %g"
                                 % ("smoothed-aggregation multigrid V(1,1) preconditioner" if amg else "block-Jacobi on the Schur diagonal", ARGS.pcg_tol),
                       "parallelism": ("BASELINE config 4: one graph edge-sharded x%d (landmark ranges), replicated multigrid hierarchy, %s" % (world, "in-cycle products on the replicated explicit level-0 matrix: one RCCL all-reduce per PCG iteration" if explicit_cycle else "RCCL all-reduce per Schur product (three per PCG iteration)")) if shard else
                                      ("request-parallel: %d independent graphs, one per GPU, no collective" % world if world > 1 else
                                       ("single GPU, collective code path forced (one-rank RCCL communicator)" if ARGS.force_collective else "single GPU")),
                       "hipgraph": graph_replay, "launch_mode": "tsgo_config.use_graphs = %d (%s)" % (0 if ARGS.no_graphs else (1 if ARGS.graphs else 2), "hipGraph replay" if graph_replay else "eager launches")},
            "gn_iters_per_s": ARGS.steps / dt,
            "pcg_iters_per_gn_iter": n_cg,
            "chi2_first_last": [chi2[0], chi2[-1]],
            "ms_per_step_device": {"linearize": ms_lin / ARGS.steps, "solve": ms_solve / ARGS.steps, "update": ms_upd / ARGS.steps},
            "ms_setup_once_per_graph": ms_setup,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": (PMC_TRAFFIC + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE passes of this same command, committed; NOT measured in this run)") if traffic is not None else None,
                         "us_per_launch": us, "algorithmic_bytes_per_launch": nbytes,
                         "us_per_launch_rocprof_stats": shares[dom].get("us_rocprof_stats"), "rocprof_stats_source": ROCPROF_STATS,
                         "kernel_chosen_by": "largest in-situ time per step, grouped by kernel symbol as rocprofv3 groups them (all levels / call sites of one instantiation together)",
                         "timing": "hipEvent before every launch of 20 eagerly launched PCG iterations on the engine's stream (tsgo_profile_iteration), minus the per-launch cost of the events themselves (%.2f us: sum of the marks vs the same iteration without them); algorithmic bytes = average over the symbol's launches in one iteration" % overhead,
                         "event_overhead_us_per_launch": overhead,
                         "kernels": shares,
                         "iteration_in_launch_order": [{"kernel": e["name"], "where": e["where"], "us_in_situ_raw": e["us"], "algorithmic_bytes": e["bytes"]} for e in prof],
                         # SURVEY 8d's own byte model, with the measured PCG iteration count: what a perfect implementation of the
                         # reference's sparse-equivalent algorithm would have to move, against what this run took
                         "step": {"algorithmic_bytes": b_gn, "B_lin": b_lin, "B_cg": b_cg, "B_upd": b_upd, "N_cg": n_cg, "ms": ms_step,
                                  "achieved": b_gn / (ms_step * 1e-3) / 1e9, "frac": b_gn / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
                         "pcg_iteration": {"algorithmic_bytes": b_cg, "us": us_pcg, "achieved": b_cg / (us_pcg * 1e-6) / 1e9,
                                           "frac": b_cg / (us_pcg * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                           "note": "the byte model prices ONE block-SpMV + vector passes; a multigrid-preconditioned iteration runs three level-0 products and the coarse levels"},
                         "us_per_pcg_iteration": us_pcg,
                         "latency_model": {"us_per_pcg_iteration": model_iter, "floor_us_per_launch": 3.7, "stream_GBps": 6000.0,
                                           "launches_per_pcg_iteration": len(prof),
                                           "note": "sum over the iteration's launches of (3.7 us of dependent round trips + memory-side bytes / 6 TB/s); kernels[*].us_latency_model per symbol"},
                         "ms_solve_outside_iterations": ms_solve / ARGS.steps - n_cg * us_pcg * 1e-3,
                         "us_multigrid_numeric_setup": us_setup},
        }
        if conv is not None:
            out["iters_to_chi2_tol"] = conv
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
    ap.add_argument("--request-parallel", dest="request_parallel", action="store_true",
                    help="N > 1: every rank optimises its own graph (weak scaling) instead of sharding ONE graph (BASELINE config 4)")
    ap.add_argument("--shard", action="store_true", help="accepted for compatibility: sharding is the default for N > 1")
    ap.add_argument("--implicit-cycle", dest="implicit_cycle", action="store_true",
                    help="N > 1: keep the implicit (sharded, all-reduced) Schur products inside the multigrid cycle: three all-reduces per PCG iteration")
    ap.add_argument("--explicit-cycle", dest="explicit_cycle", action="store_true",
                    help="N = 1: tsgo_config.cycle_level0 = 1 (what N > 1 runs by default)")
    ap.add_argument("--warm-start", dest="warm_start", type=int, default=None, help="tsgo_config.warm_start (research: 3 = third-order extrapolation)")
    ap.add_argument("--cycle-storage", dest="cycle_storage", type=int, default=16, choices=[16, 32],
                    help="tsgo_config.cycle_storage: the V-cycle's copies of the hierarchy as packed half floats (default) or f32")
    ap.add_argument("--force-collective", dest="force_collective", action="store_true",
                    help="N = 1: give the engine a one-rank RCCL communicator so that it takes the sharded code path (eager launches, all-reduce calls)")
    ap.add_argument("--no-graphs", dest="no_graphs", action="store_true", help="tsgo_config.use_graphs = 0: eager launches always")
    ap.add_argument("--graphs", dest="graphs", action="store_true", help="tsgo_config.use_graphs = 1: hipGraph replay always (default 2: eager while the host keeps ahead)")
    ap.add_argument("--no-cpu", dest="no_cpu", action="store_true")
    ap.add_argument("--no-conv", dest="no_conv", action="store_true", help="skip the 50-iteration convergence run")
    ap.add_argument("--cpu-threads", dest="cpu_threads", type=int, default=0)
    ap.add_argument("--lanes-pose", dest="lanes_pose", type=int, default=0)
    ap.add_argument("--lanes-lm", dest="lanes_lm", type=int, default=0)
    ARGS = ap.parse_args()
    main()

#!/usr/bin/env python3
"""bench.py — edges/s per Gauss-Newton iteration on the BASELINE.json workload.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3_100k|c2_10k|c5_1m] [--precision 64]
  N > 1, either way: `python bench.py --gpus N ...` starts its own N ranks (a child `python -m torch.distributed.run --nnodes=1
  --nproc-per-node N --master-addr 127.0.0.1 ...` of this script, started BEFORE this process makes any GPU call; the parent
  never touches the GPU), or the same command line under an external `torch.distributed.run` (RANK / WORLD_SIZE in the
  environment): then this process IS a rank.  Every rank runs a watchdog over the phases that can hang on a broken fabric
  (process group, RCCL communicator, first all-reduce): a phase that does not end within --phase-timeout seconds names itself
  on stderr and the rank exits 3; the parent kills the whole process group after --launch-timeout and exits non-zero.

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
# torch is imported by the ranks only (run_rank): the launching parent of a multi-GPU run must not initialise the GPU

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
PMC_TRAFFIC = os.path.join("profiles", "r04_pmc_traffic.json")   # rocprofv3 --pmc digest of this same command (tools/pmc_traffic.py)
ROCPROF_STATS = os.path.join("profiles", "r04_rocprofv3_kernel_stats.csv")   # rocprofv3 --kernel-trace --stats of this same command
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


def host_cpu():
    """nproc (this process's affinity mask) and the CPU model string, for the cpu_baseline object."""
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"nproc": len(os.sched_getaffinity(0)), "logical_cpus": os.cpu_count(), "model": model}


def cpu_baseline(g, threads):
    """SURVEY 8d's CPU figures, timed on this host beside the GPU line (rank 0, N = 1 only; ~20-30 s in all):
    (ii) the SAME graph by the CPU twin (oracle/oracle_sparse.cpp: same layout, same Schur PCG, same tolerance) at all cores
    (`value`) and at ONE thread — kind = "port": the reference's own dense algorithm cannot run this size (O(n^2) memory);
    (i) that algorithm itself (dense H, column-pivoted Householder QR, ONE thread: the reference's thread pool is disabled,
    OptimizerCpu.h:78,125-130) at the only BASELINE configuration it can hold, config 1, in float (remote/app/main.cpp:40) and in
    double, split into linearise / dense solve / update, with what the 50-iteration request of README.md:15 then costs."""
    from oracle import oracle
    from tests import util
    o = util.to_oracle(g)
    n_edges = len(g.e_type)

    def twin(n_threads, n_it):
        oracle.set_threads(n_threads)
        t0 = time.time()
        r = oracle.sparse_optimize(o, n_it, pcg_tol=ARGS.pcg_tol, precond=ARGS.precond)
        dt = time.time() - t0
        work = r["seconds_linearize"] + r["seconds_solve"]
        return {"cores": n_threads, "gn_iterations": int(r["iters"]), "seconds_per_gn_iter": work / r["iters"], "edges_per_s": n_edges * r["iters"] / work,
                "pcg_iters_per_gn_iter": float(np.mean(r["cg_iters"])), "seconds_linearize_per_iter": r["seconds_linearize"] / r["iters"],
                "seconds_solve_per_iter": r["seconds_solve"] / r["iters"], "seconds_host_setup_once": dt - work}
    big = g.n_poses > 300_000
    full = twin(threads, 2 if big else min(10, max(2, ARGS.steps)))
    one = twin(1, 1 if big else 2)
    out = {"value": full["edges_per_s"], "unit": "edges/s per GN iter", "cores": threads, "kind": "port", "host": host_cpu(),
           "sample": "the first %d GN iterations of the same %s graph by the CPU twin of the same algorithm (%s-preconditioned Schur PCG, tol %g, "
                     "%.1f PCG iterations per GN iteration) on %d threads: %.1f s of CPU work + %.1f s one-time host setup (excluded, as on the GPU side); "
                     "the same twin on ONE thread for %d iteration(s): one_thread"
                     % (full["gn_iterations"], ARGS.workload, ARGS.precond, ARGS.pcg_tol, full["pcg_iters_per_gn_iter"], threads,
                        full["seconds_per_gn_iter"] * full["gn_iterations"], full["seconds_host_setup_once"], one["gn_iterations"]),
           "pcg_iters_per_gn_iter": full["pcg_iters_per_gn_iter"], "seconds_per_gn_iter": full["seconds_per_gn_iter"],
           "all_cores": full, "one_thread": one}
    # (i) the reference's algorithm at config 1: 3 iterations per scalar type, phases timed inside the restatement
    # (oracle_last_phase_seconds) — 50 QR iterations would take 25 s per type, and every iteration is the same dense work.
    oracle.set_threads(1)
    c1g = util.c1_arrays()
    c1 = util.to_oracle(c1g)
    dense = {"kind": "reference algorithm, restated (oracle/oracle_dense.cpp); Eigen itself is not in the image",
             "workload": "config 1: 150 poses / 342 landmarks / 2123 edges, n = 1134", "solver": "dense column-pivoted QR", "cores": 1,
             "note": "dense H is n^2: 39 GB at config 2, 3.9 TB at config 3 — this algorithm cannot run the benchmarked size"}
    for prec in ("f32", "f64"):
        t0 = time.time(); rd = oracle.optimize(c1, 3, mode="cpp", solver="qr", precision=prec); per = (time.time() - t0) / rd["iters"]
        ph = oracle.last_phase_seconds() / rd["iters"]
        dense[prec] = {"gn_iterations_timed": int(rd["iters"]), "ms_per_gn_iter": 1e3 * per, "ms_linearize": 1e3 * ph[0], "ms_dense_solve": 1e3 * ph[1],
                       "ms_update": 1e3 * ph[2], "edges_per_s": len(c1g.e_type) / per,
                       "seconds_for_the_50_iteration_request": 41 * per,
                       "seconds_note": "the cpu/eigen rules stop this graph on the plateau rule at iteration 41 (tests/golden/c1_cpprules_ref.npz): 41 x ms_per_gn_iter; 3 iterations timed (the dense solve is the same work every iteration)"}
    dense["dtype"] = "f32"; dense["gn_iterations"] = dense["f32"]["gn_iterations_timed"]
    dense["ms_per_gn_iter"] = dense["f32"]["ms_per_gn_iter"]; dense["edges_per_s"] = dense["f32"]["edges_per_s"]
    out["reference_dense_c1"] = dense
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


# ---- multi-GPU launch: this script starts its own ranks, and nothing in it can hang for good -----------------------------------
PHASE_DIR_ENV = "TSGO_BENCH_PHASE_DIR"


class Watch:
    """A rank's watchdog.  `with watch.phase("rccl communicator")` notes the phase in a file the launching parent reads and, for
    the phases that can hang on a broken fabric (limit given), arms a timer thread: when the phase has not ended after `limit`
    seconds the thread names it on stderr and ends the process with os._exit(3) — the main thread may be blocked inside
    ncclCommInitRank or a collective (both release the GIL), and a process that has touched the GPU is never re-exec'd."""

    def __init__(self, rank):
        self.rank = rank
        d = os.environ.get(PHASE_DIR_ENV)
        self.path = os.path.join(d, "phase.%d" % rank) if d else None
        self.note("start")

    def note(self, name):
        self.current = name
        if self.path:
            try:
                with open(self.path + ".tmp", "w") as f:
                    f.write("%s\n%f\n" % (name, time.time()))
                os.replace(self.path + ".tmp", self.path)
            except OSError:
                pass

    def phase(self, name, limit=None):
        import contextlib
        import threading

        @contextlib.contextmanager
        def cm():
            self.note(name)
            timer = None
            if limit:
                def fire():
                    sys.stderr.write("[bench] rank %d: phase '%s' has not ended after %d s — giving up (exit 3)\n" % (self.rank, name, limit))
                    sys.stderr.flush()
                    self.note("STUCK in " + name)
                    os._exit(3)
                timer = threading.Timer(limit, fire)
                timer.daemon = True
                timer.start()
            try:
                yield
            finally:
                if timer:
                    timer.cancel()
        return cm()


def launch_ranks():
    """`python bench.py --gpus N` with no launcher around it: start N ranks as a child `torch.distributed.run` of this same command
    line, wait for them under a deadline, pass their output through.  This process makes no GPU call (it does not even import
    torch), so nothing here needs an exec of a process that has initialised the GPU."""
    import shutil
    import signal
    import socket
    import subprocess
    import tempfile
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    phase_dir = tempfile.mkdtemp(prefix="tsgo_bench_")
    env = dict(os.environ)
    env[PHASE_DIR_ENV] = phase_dir
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "2" if ARGS.dry_run_cpu else "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ARGS.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    t0 = time.time()
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)       # its own process group: one kill reaches every rank

    def phases():
        out = {}
        for r in range(ARGS.gpus):
            try:
                name, stamp = open(os.path.join(phase_dir, "phase.%d" % r)).read().split("\n")[:2]
                out[r] = (name, time.time() - float(stamp))
            except (OSError, ValueError):
                out[r] = ("not started", time.time() - t0)
        return out

    rc = None
    try:
        while True:
            try:
                rc = proc.wait(timeout=1.0)
                break
            except subprocess.TimeoutExpired:
                pass
            if time.time() - t0 > ARGS.launch_timeout:
                ph = phases()
                sys.stderr.write("[bench] %d ranks still running after %d s — killing the process group.  Phases: %s\n"
                                 % (ARGS.gpus, ARGS.launch_timeout, ", ".join("rank %d: '%s' for %.0f s" % (r, n, a) for r, (n, a) in sorted(ph.items()))))
                for sig in (signal.SIGTERM, signal.SIGKILL):
                    try:
                        os.killpg(proc.pid, sig)
                    except ProcessLookupError:
                        break
                    try:
                        proc.wait(timeout=10)
                        break
                    except subprocess.TimeoutExpired:
                        continue
                rc = 4
                break
    finally:
        if rc not in (0, None):
            ph = phases()
            sys.stderr.write("[bench] the %d-rank run ended with exit code %s after %.0f s; last phase per rank: %s\n"
                             % (ARGS.gpus, rc, time.time() - t0, ", ".join("rank %d '%s'" % (r, n) for r, (n, _a) in sorted(ph.items()))))
        shutil.rmtree(phase_dir, ignore_errors=True)
    sys.exit(rc if rc is not None else 5)


def timed_steps(opt, barrier, n_edges):
    """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides.  One step = one full Gauss-Newton
    iteration (tsgo_optimize(1)): linearise, solve, update."""
    chi2, cg = [], []
    for _ in range(ARGS.warmup):
        r = opt.optimize(1); chi2 += list(r["chi2"]); cg += list(r["cg_iters"])
    barrier()
    t0 = time.perf_counter()
    ms = {"linearize": 0.0, "solve": 0.0, "update": 0.0}
    ms_setup, graph_replay = None, False
    for _ in range(ARGS.steps):
        r = opt.optimize(1); chi2 += list(r["chi2"]); cg += list(r["cg_iters"])
        ms["linearize"] += r["ms_linearize"]; ms["solve"] += r["ms_solve"]; ms["update"] += r["ms_update"]; ms_setup = r["ms_setup"]; graph_replay = graph_replay or r["graph_replay"]
    barrier()
    dt = time.perf_counter() - t0
    return {"dt": dt, "chi2": chi2, "cg": cg, "ms": {k: v / ARGS.steps for k, v in ms.items()}, "ms_setup": ms_setup, "graph_replay": graph_replay}


def main():
    if ARGS.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks()                                # never returns
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != ARGS.gpus:
        raise SystemExit("--gpus %d, but the launcher started %d ranks" % (ARGS.gpus, world))
    watch = Watch(rank)
    if ARGS.dry_run_cpu:                              # the same spawn path over gloo + the CPU twin (tests/bench_dry_run.py); not a measurement
        from tests import bench_dry_run
        return bench_dry_run.run(ARGS, rank, world, watch)
    import torch
    n_visible = torch.cuda.device_count()             # (counting devices does not initialise the GPU)
    if n_visible < world:
        raise SystemExit("bench.py --gpus %d: %d GPUs needed, %d visible" % (world, world, n_visible))
    from toyslam_amd import synth
    from toyslam_amd.optimizer import HipOptimizer

    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with watch.phase("torch.distributed process group (RCCL) + first barrier", ARGS.phase_timeout), stdout_to_stderr():
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=ARGS.phase_timeout))
            dist.barrier()                           # the communicator (and its banner) is created lazily: now

    shard = world > 1 and not ARGS.request_parallel
    # Sharded runs take the two products INSIDE the multigrid cycle from the replicated explicit level-0 matrix
    # (tsgo_config.cycle_level0 = 1): one all-reduce per PCG iteration instead of three for 6 % more iterations — ahead as soon as an
    # all-reduce costs more than 19 us (profiles/r02e_explicit_level0_in_cycle.txt).  --implicit-cycle keeps the single-device form.
    with watch.phase("synthetic graph"):
        g = synth.make_config(ARGS.workload, seed=0 if shard else rank)
    # ... and above 500k poses the three products are worth more sharded than two all-reduces cost (24 MB each at a million poses: DESIGN.md section 5)
    explicit_cycle = (shard and not ARGS.implicit_cycle and g.n_poses <= 500000) or ARGS.explicit_cycle
    n_edges = len(g.e_type)

    def make_opt(rank_, world_, explicit_):
        return HipOptimizer(device=local_rank, precision=ARGS.precision, pcg_rel_tol=ARGS.pcg_tol, rank=rank_, world=world_,
                            use_graphs=(False if ARGS.no_graphs else (True if ARGS.graphs else "auto")), lanes_per_pose=ARGS.lanes_pose, lanes_per_lm=ARGS.lanes_lm,
                            preconditioner=ARGS.precond, warm_start=ARGS.warm_start, cycle_level0="explicit" if explicit_ else "implicit", cycle_storage=ARGS.cycle_storage)
    opt = make_opt(rank if shard else 0, world if shard else 1, explicit_cycle)
    rccl_ranks = 1
    if ARGS.force_collective and world == 1:          # research: the sharded code path (eager launches + RCCL calls) with a one-rank communicator
        with stdout_to_stderr():
            opt.comm_init(opt.comm_unique_id())
            rccl_ranks = opt.comm_selftest()
    if shard:
        with stdout_to_stderr():
            uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                uid = torch.tensor(list(opt.comm_unique_id()), dtype=torch.uint8, device="cuda")
            with watch.phase("broadcast of the RCCL unique id", ARGS.phase_timeout):
                dist.broadcast(uid, 0)
                uid_bytes = bytes(uid.cpu().tolist())
            with watch.phase("engine's RCCL communicator (ncclCommInitRank)", ARGS.phase_timeout):
                opt.comm_init(uid_bytes)
            with watch.phase("first all-reduce on the engine's communicator", ARGS.phase_timeout):
                rccl_ranks = opt.comm_selftest()      # ncclCommCount, and rank + 1 summed over the ranks comes back right
    with watch.phase("tsgo_set_graph"):
        opt.set_graph(g)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    allreduce_us = None
    if shard or (ARGS.force_collective and world == 1):
        # what each of the solver's collectives costs on THIS fabric (every rank calls alike): after a sharded product, after a linearisation,
        # after the level-0 blocks of a hierarchy build (that one is f32 in the solver; timed here in the handle's precision at the same BYTES)
        word = 8 if ARGS.precision == 64 else 4
        sizes = {"product_3P": 3 * g.n_poses + 400, "linearisation_18P": 18 * g.n_poses + 400, "level0_blocks_60MB": int(60e6) // word}
        with watch.phase("timing the all-reduces of the three buffer sizes", ARGS.phase_timeout):
            allreduce_us = {k: {"bytes": n * word, "us": opt.comm_time_allreduce(n, reps=(50 if n * word < 8e6 else 10))} for k, n in sizes.items()}
    if shard:                                         # the first iteration carries every kind of collective the run will issue
        with watch.phase("first sharded Gauss-Newton iteration (pose-partial, product and level-0 all-reduces)", ARGS.phase_timeout):
            opt.optimize(1)
            opt.set_graph(g)                          # (same structure: values refilled, solver state reset — the timed run starts where an unprobed one would)
    with watch.phase("warm-up + timed steps", 3 * ARGS.phase_timeout if world > 1 else None):
        run = timed_steps(opt, barrier, n_edges)
    dt, chi2, cg = run["dt"], run["chi2"], run["cg"]
    ms_setup, graph_replay = run["ms_setup"], run["graph_replay"]
    ms_lin, ms_solve, ms_upd = run["ms"]["linearize"] * ARGS.steps, run["ms"]["solve"] * ARGS.steps, run["ms"]["update"] * ARGS.steps
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if not shard:           # request-parallel: the job processed every rank's edges
            e = torch.tensor([float(n_edges)], dtype=torch.float64, device="cuda")
            dist.all_reduce(e, op=dist.ReduceOp.SUM)
            n_edges = int(e.item())
    timed_cg = cg[ARGS.warmup:]

    # The other multi-GPU mode beside it, same K / W: every rank its OWN graph of the configuration on its own handle, no data-path
    # collective (what a multi-GPU graph_optimizer does with independent connections): "request_parallel" in the line.
    req_par = None
    if shard and not ARGS.no_request_parallel:
        with watch.phase("request-parallel leg"):
            g_own = synth.make_config(ARGS.workload, seed=rank)
            opt_own = make_opt(0, 1, False)
            opt_own.set_graph(g_own)
            run_own = timed_steps(opt_own, barrier, len(g_own.e_type))
            opt_own.close()
            t = torch.tensor([run_own["dt"]], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e = torch.tensor([float(len(g_own.e_type))], dtype=torch.float64, device="cuda"); dist.all_reduce(e, op=dist.ReduceOp.SUM)
            req_par = {"value": float(e.item()) * ARGS.steps / float(t.item()), "unit": "edges/s", "ms_per_step": 1e3 * float(t.item()) / ARGS.steps, "scaling": "weak",
                       "pcg_iters_per_gn_iter": float(np.mean(run_own["cg"][ARGS.warmup:])),
                       "parallelism": "request-parallel: %d independent graphs of the same configuration (seed = rank), one per GPU, no collective" % world}
            del g_own
    probes_watch = watch.phase("probes after the timed region (every rank of a sharded run: they contain collectives)", 5 * ARGS.phase_timeout if world > 1 else None)
    probes_watch.__enter__()

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
            "rccl_ranks": rccl_ranks,
            "allreduce_us": allreduce_us,
            "multi_gpu_design": (("A: landmark-range edge shards, replicated multigrid hierarchy and PCG vectors (DESIGN.md section 5); value = this one graph's edges x K / max-over-ranks time" if shard else
                                  "request-parallel: one graph per GPU, no collective") if world > 1 else None),
            "request_parallel": req_par,
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
    probes_watch.__exit__(None, None, None)
    opt.close()
    if world > 1:
        with watch.phase("final barrier", ARGS.phase_timeout):
            dist.barrier()
            dist.destroy_process_group()
    watch.note("done")
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
    ap.add_argument("--no-request-parallel", dest="no_request_parallel", action="store_true",
                    help="N > 1, sharded: skip the request-parallel leg that is measured beside the sharded one")
    ap.add_argument("--phase-timeout", dest="phase_timeout", type=int, default=180,
                    help="seconds a rank's watchdog allows a phase that can hang on a broken fabric (process group, communicator, first all-reduce)")
    ap.add_argument("--launch-timeout", dest="launch_timeout", type=int, default=1500,
                    help="N > 1 without a launcher: seconds before the parent kills the ranks it started")
    ap.add_argument("--dry-run-cpu", dest="dry_run_cpu", action="store_true",
                    help="rehearsal of the N-rank launch path without GPUs: gloo + the CPU twin (tests/bench_dry_run.py); prints a line marked dry_run, not a measurement")
    ap.add_argument("--no-cpu", dest="no_cpu", action="store_true")
    ap.add_argument("--no-conv", dest="no_conv", action="store_true", help="skip the 50-iteration convergence run")
    ap.add_argument("--cpu-threads", dest="cpu_threads", type=int, default=0)
    ap.add_argument("--lanes-pose", dest="lanes_pose", type=int, default=0)
    ap.add_argument("--lanes-lm", dest="lanes_lm", type=int, default=0)
    ARGS = ap.parse_args()
    main()

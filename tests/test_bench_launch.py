"""`bench.py --gpus N` started like the N = 1 line (no launcher around it) starts its own ranks, fails loudly without GPUs, and cannot
hang: every rank's watchdog names the phase it is stuck in, and the parent kills the process group at its deadline.  The same
spawn path is rehearsed end to end over gloo + the CPU twin (tests/bench_dry_run.py)."""
import json
import os
import subprocess
import sys
import time

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def bench(*args, env=None, timeout=240):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e["OMP_WAIT_POLICY"] = "PASSIVE"
    e.update(env or {})
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH] + list(args), cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    return p.returncode, p.stdout, p.stderr, time.time() - t0


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="needs a box with fewer than 2 GPUs")
def test_two_gpus_asked_for_on_a_box_without_them_fails_in_seconds_and_says_why():
    rc, out, err, dt = bench("--gpus", "2", "--steps", "2", "--warmup", "1")
    assert rc != 0
    assert "2 GPUs needed, %d visible" % torch.cuda.device_count() in err, err[-2000:]
    assert out.strip() == ""                      # no JSON line from a run that did not happen
    assert dt < 90, dt


def test_dry_run_of_the_two_rank_launch_path_prints_one_json_line():
    rc, out, err, _ = bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run-cpu")
    assert rc == 0, err[-3000:]
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["ranks"] == 2 and d["value"] is None and "dry_run" in d
    assert d["all_reduce_calls"] > 10 and d["chi2_first_last"][1] < d["chi2_first_last"][0]


def test_a_rank_that_never_reaches_the_first_all_reduce_is_named_and_the_run_ends_non_zero():
    rc, out, err, dt = bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run-cpu", "--phase-timeout", "5", env={"TSGO_DRY_RUN_STALL_RANK": "1"})
    assert rc != 0
    assert "phase 'first all-reduce on the data path' has not ended after 5 s" in err, err[-3000:]
    assert "STUCK in first all-reduce" in err
    assert out.strip() == ""
    assert dt < 120, dt


def test_the_parent_kills_its_ranks_at_the_launch_deadline():
    # the ranks' own watchdogs are set far out: the parent's deadline is what ends this run
    rc, out, err, dt = bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run-cpu", "--phase-timeout", "600", "--launch-timeout", "12",
                             env={"TSGO_DRY_RUN_STALL_RANK": "0"})
    assert rc == 4, (rc, err[-3000:])
    assert "killing the process group" in err and "first all-reduce on the data path" in err
    assert dt < 120, dt
    # nothing of the run is left behind
    time.sleep(1.0)
    left = subprocess.run(["pgrep", "-f", "bench.py --gpus 2 --steps 2 --warmup 1 --dry-run-cpu --phase-timeout 600"], capture_output=True, text=True).stdout.split()
    assert not left, left


@pytest.mark.gpu
def test_the_bench_line_keeps_its_contract_on_one_gpu():
    """One small run of bench.py as the driver starts it (no launcher, N = 1), and one with the collective code path forced (a one-rank
    RCCL communicator, tsgo_comm_selftest): ONE JSON line on stdout with the fields the driver and the judge read, `value` consistent
    with the timed region, `roofline` and `cpu_baseline` objects present and complete."""
    rc, out, err, _ = bench("--gpus", "1", "--steps", "3", "--warmup", "1", "--workload", "c2_10k", timeout=600)
    assert rc == 0, err[-3000:]
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["config"]["workload"].startswith("c2_10k") and "model" not in d["config"]
    n_edges = 109999
    assert abs(d["value"] - n_edges / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["us_per_launch"] * 1e-6) / 1e9) <= 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "host", "all_cores", "one_thread", "reference_dense_c1"):
        assert k in c, k
    assert c["kind"] == "port" and c["one_thread"]["cores"] == 1 and c["host"]["nproc"] >= 1 and c["host"]["model"]
    for prec in ("f32", "f64"):
        ph = c["reference_dense_c1"][prec]
        assert ph["ms_dense_solve"] > 10 * ph["ms_linearize"] > 0 and ph["ms_update"] >= 0
    assert d["rccl_ranks"] == 1 and d["request_parallel"] is None
    rc, out, err, _ = bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "c2_10k", "--no-cpu", "--no-conv", "--force-collective", timeout=600)
    assert rc == 0, err[-3000:]
    d2 = json.loads([l for l in out.splitlines() if l.strip()][-1])
    assert d2["rccl_ranks"] == 1 and "collective code path forced" in d2["config"]["parallelism"]
    assert d["allreduce_us"] is None and set(d2["allreduce_us"]) == {"product_3P", "linearisation_18P", "level0_blocks_60MB"}      # tsgo_comm_time_allreduce: RCCL calls of the solver's three buffer sizes
    assert all(v["us"] > 0 and v["bytes"] > 0 for v in d2["allreduce_us"].values()), d2["allreduce_us"]
    assert abs(d2["chi2_first_last"][0] - d["chi2_first_last"][0]) <= 1e-9 * d["chi2_first_last"][0]

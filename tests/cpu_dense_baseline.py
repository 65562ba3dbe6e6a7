"""the reference's own algorithm (dense H, column-pivoted QR, single thread) timed on this
host through its restatement oracle/oracle_dense.cpp, at config 1 (the only BASELINE config it can run) and
at growing n to show the O(n^3) wall; beside it the HIP path on the same graphs when a GPU is present.
Measurement tool: imports oracle/ as the timed CPU baseline, exactly like bench.py's cpu_baseline leg."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import util
from oracle import oracle
from toyslam_amd import synth


def timed(f, reps=1):
    best = None
    for _ in range(reps):
        t = time.perf_counter(); r = f(); dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    return best, r


def gpu_run(g, iters):
    import torch
    if not torch.cuda.is_available():
        return None
    from toyslam_amd.optimizer import HipOptimizer
    o = HipOptimizer(); o.set_graph(g); o.optimize(2)          # warm the device path
    o.set_graph(g)
    t = time.perf_counter(); r = o.optimize(iters); dt = time.perf_counter() - t
    v = o.vertices(); o.close()
    return dt, r, v


def main():
    oracle.set_threads(1)
    g = util.c1_arrays(as_wire=True)
    og = util.to_oracle(g)
    n_edges = len(g.e_type)
    print("config 1: %d vertices, %d edges, n = %d unknowns; host: %d logical CPUs" % (len(g.v_id), n_edges, 3 * int((g.v_type == 0).sum()) + 2 * int((g.v_type == 1).sum()), os.cpu_count()))
    for prec in ("f32", "f64"):
        t_lin, _ = timed(lambda: oracle.linearize(og, precision=prec), 3)
        for solver in ("qr", "chol"):
            its = 10 if solver == "qr" else 50          # QR: ~2 s per iteration at n = 1134 in scalar code
            dt, r = timed(lambda: oracle.optimize(og, its, mode="cpp", solver=solver, precision=prec))
            per = dt / r["iters"]
            print("  dense %s %-4s: %d GN iterations (stop: %s) in %.2f s = %.1f ms/iteration (linearise %.1f ms, solve ~%.1f ms) -> %.3g edges/s, chi2 %.4f -> %.4f"
                  % (prec, solver, r["iters"], r["stop"], dt, 1e3 * per, 1e3 * t_lin, 1e3 * (per - t_lin), n_edges / per, r["chi2"][0], r["chi2"][-1]), flush=True)
    res = gpu_run(g, 50)
    if res:
        dt, r, v = res
        ref = oracle.optimize(og, 50, mode="cpp", solver="chol", precision="f64")
        print("  HIP f64      : %d GN iterations (stop: %s) in %.4f s = %.2f ms/iteration -> %.3g edges/s; max vertex difference to dense f64: %.2e"
              % (r["iters"], r["stop"], dt, 1e3 * dt / r["iters"], n_edges * r["iters"] / dt, util.max_vertex_diff(v, ref["v_pos"], g.v_type)))
    print("the O(n^3) wall (one GN iteration, dense f64, QR as the reference / Cholesky as the cheapest exact stand-in):")
    max_p = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    for P in [p for p in (150, 300, 600, 1200) if p <= max_p]:
        gs = synth.make(P, seed=1); os_ = util.to_oracle(gs)
        n = 3 * P + 2 * int((gs.v_type == 1).sum())
        line = "  P = %5d, n = %6d:" % (P, n)
        for solver in ("qr", "chol"):
            if solver == "qr" and n > 2500:
                line += "  qr      (skipped)"; continue
            dt, _ = timed(lambda: oracle.optimize(os_, 1, mode="cpp", solver=solver, precision="f64"))
            line += "  %s %8.3f s" % (solver, dt)
        res = gpu_run(gs, 1)
        if res:
            line += "  | HIP %.2f ms" % (1e3 * res[0])
        print(line, flush=True)


if __name__ == "__main__":
    main()

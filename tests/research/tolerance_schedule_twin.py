"""Which Gauss-Newton iteration's PCG tolerance decides the final vertices?  (CPU twin; round 4)

For N iterations at config 2 (or --poses): the reference run solves every iteration to 1e-13; run k solves iteration k alone to
`--probe` (1e-6) and everything else to 1e-13 — the final vertex difference is the weight w_k of that solve's error.  Then a few
schedules (fixed, loose-then-tight) with their PCG iteration totals.  TSGO_TWIN_TOL_SCHED (oracle_sparse.cpp) carries the schedule.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from tests import util  # noqa: E402
from toyslam_amd import synth  # noqa: E402


def run(o, n, sched):
    os.environ["TSGO_TWIN_TOL_SCHED"] = ",".join("%g" % t for t in sched)
    t0 = time.time()
    r = oracle.sparse_optimize(o, n, pcg_tol=1e-13, precond="amg")
    r["wall"] = time.time() - t0
    return r


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--poses", type=int, default=10000)
    ap.add_argument("--iters", type=int, default=25)
    ap.add_argument("--probe", type=float, default=1e-6)
    ap.add_argument("--weights", action="store_true")
    ARGS = ap.parse_args()
    g = synth.make(ARGS.poses, 10, seed=0)
    o = util.to_oracle(g)
    N = ARGS.iters
    ref = run(o, N, [1e-13])
    print("reference: %d iterations, stop %s, %d PCG iterations, chi2 %.6f" % (ref["iters"], ref["stop"], ref["cg_iters"].sum(), ref["chi2"][-1]), flush=True)

    def report(name, sched):
        r = run(o, N, sched)
        dv = util.max_vertex_diff(r["v_pos"], ref["v_pos"], g.v_type)
        pose = g.v_type == 0
        dpose = np.abs(r["v_pos"][pose][:, :2] - ref["v_pos"][pose][:, :2]).max()
        dc = abs(r["chi2"][-1] - ref["chi2"][-1]) / ref["chi2"][-1]
        print("%-44s PCG %5d (%.2f per solve)  max vertex diff %.2e (poses x,y %.2e)  chi2 rel %.1e  stop %s/%d" % (name, r["cg_iters"].sum(), r["cg_iters"].mean(), dv, dpose, dc, r["stop"], r["iters"]), flush=True)
        return r

    if ARGS.weights:
        for k in range(N):
            sched = [1e-13] * N
            sched[k] = ARGS.probe
            report("only iteration %2d at %g" % (k, ARGS.probe), sched)
    for tol in (1e-10, 1e-9, 3e-9, 1e-8, 1e-7):
        report("fixed %g" % tol, [tol])
    for loose in (1e-8, 1e-7, 1e-6):
        for m in (2, 4, 8):
            report("%g, last %d at 1e-10" % (loose, m), [loose] * (N - m) + [1e-10] * m)
    # geometric: tolerance tightens by the factor the damped step leaves (0.8 per iteration) towards 1e-10 at the end
    for start in (1e-7, 1e-6):
        sched = [max(1e-10, start * 0.8 ** k) for k in range(N)]
        report("%g x 0.8^k, floor 1e-10" % start, sched)
        sched = [min(start, 1e-10 / 0.8 ** (N - 1 - k)) for k in range(N)]
        report("1e-10 / 0.8^(N-1-k), cap %g" % start, sched)

"""Research (GPU): replays trial 33 of `soak_gpu.py 420 31 30000 160000` (the soak's RNG sequence, graphs rebuilt on the host)
and runs the device at three PCG tolerances: how far apart do two converged runs of the SAME implementation end on that
graph?  (profiles/r02t_soak_trial33_tolerance_study.txt)"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import util
from toyslam_amd import synth
from toyslam_amd.graph import GraphArrays
from toyslam_amd.optimizer import HipOptimizer
rng = np.random.default_rng(31)
n_lo, n_hi = 30000, 160000
for trial in range(34):
    n = int(rng.integers(n_lo, n_hi)); k = int(rng.integers(2, 15)); lc = int(rng.integers(0, max(1, n // 40)))
    seed = int(rng.integers(0, 10 ** 6))
    if trial < 33:
        # vertex count without building the graph is not available: build (host only)
        g = synth.make(n, k, loop_closures=lc, seed=seed)
    else:
        g = synth.make(n, k, loop_closures=lc, seed=seed)
    if trial % 7 == 3:
        keep = g.e_type == 0; pose = g.v_type == 0
        g = GraphArrays(g.v_id[pose], g.v_type[pose], g.v_pos[pose], g.e_type[keep], g.e_ids[keep], g.e_meas[keep], g.e_inf[keep], g.fixed)
    fx = [0] + [int(v) for v in rng.choice(g.v_id, size=int(rng.integers(0, 3)), replace=False)]
    g.fixed = np.array(fx, np.uint32)
    if trial % 5 == 2:
        rng.choice([0.2, 0.5, 1.0])
print("trial 33: n", n, "k", k, "lc", lc, "seed", seed, "fixed", fx, "extent", float(np.abs(g.v_pos).max()), flush=True)
res = {}
for tol in (1e-9, 1e-11, 1e-13):
    o = HipOptimizer(pcg_rel_tol=tol, odom_jacobian="analytic")
    o.set_graph(g); r = o.optimize(12); res[tol] = (r, o.vertices()); o.close()
    print(tol, list(map(int, r["cg_iters"])), r["chi2"][-1], flush=True)
for a, b in ((1e-9, 1e-13), (1e-11, 1e-13)):
    print("device tol %g vs %g: max vertex diff %.3e, chi2 rel diff %.2e" % (a, b, util.max_vertex_diff(res[a][1], res[b][1], g.v_type), abs(res[a][0]["chi2"][-1] - res[b][0]["chi2"][-1]) / res[b][0]["chi2"][-1]))

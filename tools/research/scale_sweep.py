"""Research: multigrid-PCG iteration counts and fallbacks across graph sizes and shapes (GPU)."""
import sys, os; sys.path.insert(0, ".")
from toyslam_amd import synth
from toyslam_amd.optimizer import HipOptimizer
cases = [(n, 10, 0, 0) for n in (2000, 50000, 200000, 400000, 700000)] + [(1000000, 8, 500000, 0), (300000, 4, 0, 1), (300000, 20, 30000, 2)]
for n, k, lc, seed in cases:
    g = synth.make(n, k, loop_closures=lc, seed=seed)
    o = HipOptimizer(pcg_max_iters=600); o.set_graph(g); r = o.optimize(6)
    print("n", n, "lm/pose", k, "closures", lc, "cg", list(r["cg_iters"]), "fallbacks", r["fallbacks"], "ms_solve", round(r["ms_solve"], 1),
          "ms_lin", round(r["ms_linearize"], 1), flush=True)
    o.close()

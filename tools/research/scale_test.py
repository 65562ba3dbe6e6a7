import sys, os; sys.path.insert(0, ".")
from toyslam_amd import synth
from toyslam_amd.optimizer import HipOptimizer
for n in (200000, 400000, 700000, 1000000):
    g = synth.make(n, 8, loop_closures=n // 2, seed=0) if n == 1000000 else synth.make(n, 10, seed=0)
    o = HipOptimizer(pcg_max_iters=600); o.set_graph(g); r = o.optimize(2)
    print("agg", os.environ.get("TSGO_AGGC", "8"), "n", n, "cg", list(r["cg_iters"]), "fallbacks", r["fallbacks"], "ms_solve", round(r["ms_solve"], 1), flush=True)
    o.close()

"""Research: smoothing sweeps on small coarse levels — wall time of 10 GN iterations on small graphs (GPU)."""
import os, sys, time; sys.path.insert(0, ".")
from toyslam_amd import synth
from toyslam_amd.optimizer import HipOptimizer
for n in (150, 1000, 3000, 10000, 30000):
    g = synth.make(n, 10, seed=2)
    o = HipOptimizer(); o.set_graph(g); o.optimize(3); o.set_graph(g)
    t = time.perf_counter(); r = o.optimize(10); dt = time.perf_counter() - t
    print("sweeps", os.environ.get("TSGO_SWEEPS_LIST", "rule"), "n", n, "cg", int(r["cg_total"]), "ms/GN iteration %.3f" % (1e2 * dt), flush=True)
    o.close()

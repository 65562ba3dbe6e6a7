import sys
sys.path.insert(0,'/root/repo')
import numpy as np
from toyslam_amd import synth
from toyslam_amd.optimizer import HipOptimizer
g = synth.make_config("c3_100k")
out = {}
for mode in (True, "auto"):
    o = HipOptimizer(use_graphs=mode)
    o.set_graph(g); r = o.optimize(14); v = o.vertices(); o.close()
    out[mode] = (r, v)
    print(mode, list(map(int, r["cg_iters"])), r["chi2"][-1], r["graph_replay"])
print("max vertex diff", np.abs(out[True][1] - out["auto"][1]).max(), "chi2 equal", np.array_equal(out[True][0]["chi2"], out["auto"][0]["chi2"]))

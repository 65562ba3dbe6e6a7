import sys, os
sys.path.insert(0, '/root/repo')
from toyslam_amd import synth
from toyslam_amd.optimizer import HipOptimizer
g = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else "c3_100k")
o = HipOptimizer()
o.set_graph(g); o.optimize(2)      # code objects
o.set_graph(g)
r = o.optimize(12)
print(os.environ.get("TSGO_HIER_MAX_AGE"), list(r["cg_iters"]), "ms_total %.2f solve %.2f lin %.2f" % (r["ms_total"], r["ms_solve"], r["ms_linearize"]))
o.close()

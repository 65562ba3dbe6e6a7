import sys, time
sys.path.insert(0, ".")
from toyslam_amd import synth
from toyslam_amd.optimizer import HipOptimizer
g = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else "c3_100k")
o = HipOptimizer()
for rep in range(2):
    t = time.perf_counter(); o.set_graph(g); print("set_graph %.1f ms" % (1e3 * (time.perf_counter() - t)), flush=True)
    g2 = synth.make(len(g.v_id) // 3 + 10 + rep, 10, seed=3)   # force a rebuild next time
    o.set_graph(g2)
o.close()

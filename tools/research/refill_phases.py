"""Research (GPU): where a same-structure tsgo_set_graph (refill) spends its time.  TSGO_STAGE_TIMING=1 prints the laps."""
import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np
from toyslam_amd import synth
from toyslam_amd.optimizer import HipOptimizer
g = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else "c3_100k")
o = HipOptimizer()
o.set_graph(g); o.optimize(2)
for k in range(3):
    g.v_pos[:, 0] += 1e-6
    t = time.perf_counter(); o.set_graph(g); dt = 1e3 * (time.perf_counter() - t)
    r = o.optimize(1)
    print("refill %d: set_graph %.2f ms (reported %.2f), reused %s" % (k, dt, r["ms_setup"], r["structure_reused"]), flush=True)
o.close()

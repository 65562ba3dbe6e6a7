"""The multigrid numeric set-up alone (tsgo_time_kernel which = 6), for rocprofv3 --kernel-trace --stats: which of its kernels cost what."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from toyslam_amd import synth
from toyslam_amd.optimizer import HipOptimizer
g = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else "c3_100k")
o = HipOptimizer(); o.set_graph(g); o.optimize(2)
print("us per set-up:", o.time_kernel(6, reps=20)[0])
o.close()

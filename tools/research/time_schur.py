"""Launch only the two Schur kernels many times (for PMC passes)."""
import sys
sys.path.insert(0, ".")
from toyslam_amd import synth
from toyslam_amd.optimizer import HipOptimizer
g = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else "c3_100k")
o = HipOptimizer(preconditioner="jacobi"); o.set_graph(g)
print("schur_lm us", o.time_kernel(0, 100)[0], "schur_pose us", o.time_kernel(1, 100)[0])

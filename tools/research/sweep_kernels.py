"""GPU-side sweep (research): table-kernel time vs lanes-per-vertex and the XCD map, C3 graph."""
import sys
sys.path.insert(0, ".")
from toyslam_amd import synth
from toyslam_amd.optimizer import HipOptimizer
g = synth.make_config("c3_100k")
names = {0: "schur_lm", 1: "schur_pose", 3: "lin_lm", 4: "lin_pose"}
for xcd in (1,):
    for gp, gl in ((2, 1), (2, 2), (2, 4), (2, 8), (1, 4), (4, 4)):
        o = HipOptimizer(lanes_per_pose=gp, lanes_per_lm=gl, xcd_map=xcd, preconditioner="jacobi")
        o.set_graph(g)
        t = {k: o.time_kernel(k, 200)[0] for k in names}
        o.close()
        print("xcd=%d lanes pose=%d lm=%d : " % (xcd, gp, gl) + "  ".join("%s %.1f" % (names[k], t[k]) for k in names), flush=True)

"""One PCG iteration kernel by kernel (tsgo_profile_iteration) with the library given on the command line — for A/B runs of experimental
builds of libtsgo_hip.so (timing only: such a build need not compute right answers).
    python tests/research/iteration_profile.py toyslam_amd/libtsgo_hip.so [workload]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from toyslam_amd import build, synth
build.HIP_SO = os.path.abspath(sys.argv[1])
from toyslam_amd.optimizer import HipOptimizer
g = synth.make_config(sys.argv[2] if len(sys.argv) > 2 else "c3_100k")
o = HipOptimizer(); o.set_graph(g); o.optimize(2)
for rep in range(2):
    prof = o.profile_iteration(reps=20)
tot = 0.0
for e in prof:
    print("%-44s %-40s %8.2f us" % (e["name"][:44], e["where"][:40], e["us"])); tot += e["us"]
print("sum of the marks: %.1f us; iteration without marks: %.1f us" % (tot, o.time_kernel(5, reps=20)[0]))
o.close()

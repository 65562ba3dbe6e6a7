"""bench.py with another build of libtsgo_hip.so (A/B runs of experimental builds whose NUMERICS differ, e.g. another prolongator damping:
iteration counts matter, so the whole bench step is run, not one iteration's profile).
    python tests/research/bench_with_lib.py toyslam_amd/libtsgo_hip_variant.so [bench.py arguments]"""
import os
import runpy
import sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from toyslam_amd import build
build.HIP_SO = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")

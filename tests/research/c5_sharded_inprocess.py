"""Research run (GPU): BASELINE config 5 (1 M poses / 10.1 M edges) through the edge-sharded DEVICE path with in-process ranks on the one GPU of
a box (tsgo_comm_init_local, TSGO_TESTING build) — the size at which a sharded run keeps the implicit cycle (three sharded products per
PCG iteration, bench.py) and at which k_cg_step's partials are folded first (> 1 024 workgroups).  Against the single handle: same stop,
PCG counts within 2, chi^2 to 1e-10, vertices to 1e-8, ranks bit-identical.
    python tests/research/c5_sharded_inprocess.py [world=2] [iterations=2] [cycle=implicit]"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from tests import util
from tests.test_gpu_sharded_inprocess import _merge_landmarks, _run_sharded
from toyslam_amd import synth
from toyslam_amd.optimizer import HipOptimizer

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cycle = sys.argv[3] if len(sys.argv) > 3 else "implicit"
g = synth.make_config("c5_1m")
print("c5_1m: %d vertices, %d edges; %d in-process ranks, %d GN iterations, cycle_level0=%s" % (len(g.v_id), len(g.e_type), world, iters, cycle), flush=True)
t = time.time()
single = HipOptimizer(pcg_rel_tol=1e-12, cycle_level0=cycle)
try:
    single.set_graph(g); rs = single.optimize(iters); vs = single.vertices()
finally:
    single.close()
print("single handle: %.1f s, chi2 %s, PCG %s" % (time.time() - t, rs["chi2"], list(map(int, rs["cg_iters"]))), flush=True)
t = time.time()
outs = _run_sharded(g, world, iters, pcg_rel_tol=1e-12, cycle_level0=cycle)
print("%d ranks: %.1f s, PCG %s" % (world, time.time() - t, list(map(int, outs[0][0]["cg_iters"]))), flush=True)
ok = True
for r, _ in outs:
    ok &= bool(np.allclose(r["chi2"], rs["chi2"], rtol=1e-10)) and r["stop"] == rs["stop"] and r["iters"] == rs["iters"]
    ok &= bool(np.array_equal(r["chi2"], outs[0][0]["chi2"])) and bool(np.array_equal(r["cg_iters"], outs[0][0]["cg_iters"]))
    ok &= bool(np.all(np.abs(r["cg_iters"] - rs["cg_iters"]) <= 2)) and r["fallbacks"] == 0
v = _merge_landmarks(g, outs)
d = util.max_vertex_diff(v, vs, g.v_type)
print("chi2 sharded %s; max vertex difference to the single handle %.2e; %s" % (outs[0][0]["chi2"], d, "ok" if ok and d < 1e-8 else "MISMATCH"), flush=True)
sys.exit(0 if ok and d < 1e-8 else 1)

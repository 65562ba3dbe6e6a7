"""Odometry-only graphs under the analytic ODOM Jacobians are beam-like chains: thousands of multigrid PCG
iterations per solve.  How the cycle's storage (packed half / f32) and the tolerance change that.  GPU box.
usage: python tools/research/hard_chain.py [n_poses] [seed]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from toyslam_amd import synth
from toyslam_amd.graph import GraphArrays
from toyslam_amd.optimizer import HipOptimizer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 23753
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
g = synth.make(n, 13, loop_closures=2, seed=seed)
keep = g.e_type == 0; pose = g.v_type == 0
g = GraphArrays(g.v_id[pose], g.v_type[pose], g.v_pos[pose], g.e_type[keep], g.e_ids[keep], g.e_meas[keep], g.e_inf[keep], np.array([0], np.uint32))
for bits in (32, 16):
    for tol in (1e-11, 1e-10):
        o = HipOptimizer(pcg_rel_tol=tol, odom_jacobian="analytic", cycle_storage=bits)
        try:
            t = time.time(); o.set_graph(g); r = o.optimize(6)
            print("cycle storage %d, tol %.0e: GN %d stop %s  PCG %s  fallbacks %d  chi2 %s  %.1f s" % (bits, tol, r["iters"], r["stop"], list(map(int, r["cg_iters"])), r["fallbacks"], ["%.6g" % c for c in r["chi2"]], time.time() - t), flush=True)
        finally:
            o.close()

import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import util
from oracle import oracle
from toyslam_amd import synth
from toyslam_amd.graph import GraphArrays
from toyslam_amd.optimizer import HipOptimizer
oracle.set_threads(16)
g = synth.make(22368, 7, loop_closures=466, seed=11)
keep = g.e_type == 0; pose = g.v_type == 0
g = GraphArrays(g.v_id[pose], g.v_type[pose], g.v_pos[pose], g.e_type[keep], g.e_ids[keep], g.e_meas[keep], g.e_inf[keep], g.fixed)
print("position scale", np.abs(g.v_pos).max())
ref = oracle.sparse_optimize(util.to_oracle(g), 12, pcg_tol=1e-13, precond="amg")
refj = oracle.sparse_optimize(util.to_oracle(g), 12, pcg_tol=1e-13, precond="jacobi")
print("twin amg vs twin jacobi:", util.max_vertex_diff(ref["v_pos"], refj["v_pos"], g.v_type), ref["stop"], refj["stop"], ref["chi2"], ref["delta_norm"])
for tol in (1e-10, 1e-11, 1e-12, 1e-13):
    o = HipOptimizer(pcg_rel_tol=tol); o.set_graph(g); r = o.optimize(12); v = o.vertices(); o.close()
    print("tol %.0e: iters %d stop %s cg %s diff to twin %.2e" % (tol, r["iters"], r["stop"], list(map(int, r["cg_iters"])), util.max_vertex_diff(v, ref["v_pos"], g.v_type)))

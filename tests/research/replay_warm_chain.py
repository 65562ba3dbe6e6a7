"""Research (GPU): replays, through ONE handle with tsgo_config.warm_requests, the "second request" sequence of the given trials of
`soak_gpu.py <budget> SEED` (each: the graph, then the graph again with the returned estimates), and compares every result with a
fresh cold handle on the same input.  usage: replay_warm_chain.py SEED TRIAL [TRIAL ...]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import util
from toyslam_amd import synth
from toyslam_amd.graph import GraphArrays
from toyslam_amd.optimizer import HipOptimizer
seed0 = int(sys.argv[1]); wanted = [int(a) for a in sys.argv[2:]]
rng = np.random.default_rng(seed0)
graphs = {}
for trial in range(max(wanted) + 1):
    n = int(rng.integers(300, 30000)); k = int(rng.integers(2, 15)); lc = int(rng.integers(0, max(1, n // 40)))
    seed = int(rng.integers(0, 10 ** 6))
    g = synth.make(n, k, loop_closures=lc, seed=seed) if trial in wanted else None
    nv = None
    if g is None:      # the fixed-vertex draw needs the vertex ids: build anyway (host only)
        g = synth.make(n, k, loop_closures=lc, seed=seed)
    if trial % 7 == 3:
        keep = g.e_type == 0; pose = g.v_type == 0
        g = GraphArrays(g.v_id[pose], g.v_type[pose], g.v_pos[pose], g.e_type[keep], g.e_ids[keep], g.e_meas[keep], g.e_inf[keep], g.fixed)
    fx = [0] + [int(v) for v in rng.choice(g.v_id, size=int(rng.integers(0, 3)), replace=False)]
    g.fixed = np.array(fx, np.uint32)
    if trial % 5 == 2:
        rng.choice([0.2, 0.5, 1.0])
    if trial in wanted:
        graphs[trial] = g
        print("trial %d: n %d k %d closures %d fixed %s" % (trial, n, k, lc, fx), flush=True)
ow = HipOptimizer(pcg_rel_tol=1e-11, warm_requests=True)
def cold(g):
    o = HipOptimizer(pcg_rel_tol=1e-11)
    try:
        o.set_graph(g); r = o.optimize(12); return r, o.vertices()
    finally:
        o.close()
for t in wanted:
    g = graphs[t]
    for which in ("first", "second"):
        ow.set_graph(g); r = ow.optimize(12); v = ow.vertices()
        rc, vc = cold(g)
        print("trial %d %s request: history %d stop %s/%s cg %s | cold %s | chi2 last %.9g / %.9g | max vertex diff %.3e" %
              (t, which, r["history_carried"], r["stop"], rc["stop"], list(map(int, r["cg_iters"])), list(map(int, rc["cg_iters"])), r["chi2"][-1], rc["chi2"][-1], util.max_vertex_diff(v, vc, g.v_type)), flush=True)
        g = GraphArrays(g.v_id, g.v_type, v.astype(np.float32).astype(np.float64), g.e_type, g.e_ids, g.e_meas, g.e_inf, g.fixed)
ow.close()

"""Research (GPU): replays one trial of `soak_gpu.py <budget> <seed> [n_lo n_hi]` (the soak's RNG sequence, graphs rebuilt on the
host) and its "second request with the returned estimates" on handles with and without tsgo_config.warm_requests, at two PCG
tolerances, against the twin: is a mismatch there the carried history's doing, or the graph's conditioning?
usage: replay_trial_r3.py SEED TRIAL [n_lo n_hi]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import util
from oracle import oracle
from toyslam_amd import synth
from toyslam_amd.graph import GraphArrays
from toyslam_amd.optimizer import HipOptimizer
seed0, want = int(sys.argv[1]), int(sys.argv[2])
n_lo, n_hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (300, 30000)
rng = np.random.default_rng(seed0)
for trial in range(want + 1):
    n = int(rng.integers(n_lo, n_hi)); k = int(rng.integers(2, 15)); lc = int(rng.integers(0, max(1, n // 40)))
    seed = int(rng.integers(0, 10 ** 6))
    g = synth.make(n, k, loop_closures=lc, seed=seed)
    shape = "landmarks"
    if trial % 7 == 3:
        keep = g.e_type == 0; pose = g.v_type == 0
        g = GraphArrays(g.v_id[pose], g.v_type[pose], g.v_pos[pose], g.e_type[keep], g.e_ids[keep], g.e_meas[keep], g.e_inf[keep], g.fixed)
        shape = "pose graph"
    fx = [0] + [int(v) for v in rng.choice(g.v_id, size=int(rng.integers(0, 3)), replace=False)]
    g.fixed = np.array(fx, np.uint32)
    oj = "analytic" if trial % 4 == 1 else "constant"
    if oj == "analytic" and shape == "pose graph" and n > 15000:
        oj = "constant"
    rules, lr = ("python", float(rng.choice([0.2, 0.5, 1.0]))) if trial % 5 == 2 else ("cpp", 0.2)
print("trial %d: %s n %d k %d closures %d seed %d fixed %s oj %s rules %s lr %g extent %.1f" % (want, shape, n, k, lc, seed, fx, oj, rules, lr, float(np.abs(g.v_pos).max())), flush=True)
oracle.set_threads(16)
kw = dict(odom_jacobian=oj, rules=rules, lr=lr)
o = HipOptimizer(pcg_rel_tol=1e-11, **kw)
o.set_graph(g); r0 = o.optimize(12); v0 = o.vertices(); o.close()
print("first request: cg", list(map(int, r0["cg_iters"])), "chi2", r0["chi2"][0], "->", r0["chi2"][-1], flush=True)
g2 = GraphArrays(g.v_id, g.v_type, v0.astype(np.float32).astype(np.float64), g.e_type, g.e_ids, g.e_meas, g.e_inf, g.fixed)
res = {}
for name, tol, warm in (("cold 1e-11", 1e-11, False), ("cold 1e-13", 1e-13, False), ("warm 1e-11", 1e-11, True), ("warm 1e-13", 1e-13, True)):
    o = HipOptimizer(pcg_rel_tol=tol, warm_requests=warm, **kw)
    if warm:
        o.set_graph(g); o.optimize(12)
    o.set_graph(g2); r = o.optimize(12); res[name] = (r, o.vertices()); o.close()
    print("%s: history %d cg %s chi2 %s" % (name, r["history_carried"], list(map(int, r["cg_iters"])), ["%.9g" % c for c in r["chi2"][[0, 5, -1]]]), flush=True)
oracle.set_odom_jacobian(oj)
for tol in (1e-12, 1e-14):
    ref = oracle.sparse_optimize(util.to_oracle(g2), 12, pcg_tol=tol, precond="amg", rules=rules, lr=lr)
    res["twin %g" % tol] = (ref, ref["v_pos"])
    print("twin %g: cg %s chi2 %s" % (tol, list(map(int, ref["cg_iters"])), ["%.9g" % c for c in ref["chi2"][[0, 5, -1]]]), flush=True)
oracle.set_odom_jacobian("constant")
names = list(res)
for i in range(len(names)):
    for j in range(i + 1, len(names)):
        print("%-12s vs %-12s max vertex diff %.3e" % (names[i], names[j], util.max_vertex_diff(res[names[i]][1], res[names[j]][1], g.v_type)))

"""Research (GPU): replays one trial of `soak_gpu.py <budget> <seed> [n_lo n_hi]` (round 4's RNG sequence, virtual-landmark trials included) on the
DEVICE only: the first request and the "second request with the returned estimates" (f32-rounded, warm handle), under the sweep rule of the
library and under the lists given (TSGO_SWEEPS_LIST, TSGO_TESTING build) — is a second request that needs hundreds of PCG iterations the
rule's doing or the graph's?     usage: replay_trial_r4.py SEED TRIAL n_lo n_hi [LIST ...]"""
import os
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import util
from toyslam_amd import build, synth
if os.environ.get("TSGO_REPLAY_LIB"):
    build.HIP_SO = os.path.abspath(os.environ["TSGO_REPLAY_LIB"])      # another build of the product library (A/B of a replayed trial)
from toyslam_amd.graph import GraphArrays
from toyslam_amd.optimizer import HipOptimizer
seed0, want, n_lo, n_hi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
lists = sys.argv[5:]
rng = np.random.default_rng(seed0)
for trial in range(want + 1):
    n = int(rng.integers(n_lo, n_hi)); k = int(rng.integers(2, 15)); lc = int(rng.integers(0, max(1, n // 40)))
    seed = int(rng.integers(0, 10 ** 6))
    last = trial == want
    g = synth.make(n, k, loop_closures=lc, seed=seed) if last else None
    shape = "landmarks"
    if trial % 7 == 3:
        shape = "pose graph"
        if last:
            keep = g.e_type == 0; pose = g.v_type == 0
            g = GraphArrays(g.v_id[pose], g.v_type[pose], g.v_pos[pose], g.e_type[keep], g.e_ids[keep], g.e_meas[keep], g.e_inf[keep], g.fixed)
    if trial % 6 == 4 and shape == "landmarks":
        frac = float(rng.choice([0.2, 0.5, 0.9])); s2 = int(rng.integers(0, 10 ** 6)); keep_lm = bool(rng.integers(0, 2))
        if last:
            g = util.with_virtual_landmarks(g, frac, seed=s2, keep_lm=keep_lm)
        shape = "landmarks+vlm"
    if last:
        fx = [0] + [int(v) for v in rng.choice(g.v_id, size=int(rng.integers(0, 3)), replace=False)]
    else:      # the same draws without building the graph: its vertex ids are 0 .. n_vertices-1 in synth.make's output only for the LAST trial we need them
        gg = synth.make(n, k, loop_closures=lc, seed=seed)
        if shape == "pose graph":
            gg_ids = gg.v_id[gg.v_type == 0]
        elif shape == "landmarks+vlm":
            gg_ids = util.with_virtual_landmarks(gg, frac, seed=s2, keep_lm=keep_lm).v_id
        else:
            gg_ids = gg.v_id
        fx = [0] + [int(v) for v in rng.choice(gg_ids, size=int(rng.integers(0, 3)), replace=False)]
    oj = "analytic" if trial % 4 == 1 else "constant"
    if oj == "analytic" and shape == "pose graph" and n > 15000:
        oj = "constant"
    rules, lr = ("python", float(rng.choice([0.2, 0.5, 1.0]))) if trial % 5 == 2 else ("cpp", 0.2)
g.fixed = np.array(fx, np.uint32)
print("trial %d: %s n %d k %d closures %d seed %d fixed %s oj %s rules %s lr %g extent %.1f" % (want, shape, n, k, lc, seed, fx, oj, rules, lr, float(np.abs(g.v_pos).max())), flush=True)
kw = dict(odom_jacobian=oj, rules=rules, lr=lr)
keep = {}
for lst in [None, "tol13"] + lists:
    tol = 1e-11
    if lst == "tol13":      # the library's rule again at a tighter tolerance: how far apart do two device runs of the SAME request end?
        lst = None; tol = 1e-13
    os.environ.pop("TSGO_SWEEPS_LIST", None); os.environ.pop("TSGO_HIER_SHIFT", None)
    if lst is not None and lst.startswith("shift="):
        os.environ["TSGO_HIER_SHIFT"] = lst[6:]
    elif lst is not None:
        os.environ["TSGO_SWEEPS_LIST"] = lst
    o = HipOptimizer(pcg_rel_tol=tol, warm_requests=True, testing=lst is not None, **kw)
    o.set_graph(g); r0 = o.optimize(12); v0 = o.vertices()
    g2 = GraphArrays(g.v_id, g.v_type, v0.astype(np.float32).astype(np.float64), g.e_type, g.e_ids, g.e_meas, g.e_inf, g.fixed)
    o.set_graph(g2); r = o.optimize(12); v2 = o.vertices(); o.close()
    keep[(lst, tol)] = v2
    if (None, 1e-11) in keep and (lst, tol) != (None, 1e-11):
        print("   max vertex difference of this second request's result to the library's rule at 1e-11: %.3e" % util.max_vertex_diff(v2, keep[(None, 1e-11)], g.v_type), flush=True)
    print("%-22s first: cg %s chi2 %.6g -> %.6g | second (f32 estimates): cg %s chi2 %.9g -> %.9g fallbacks %d" % (("library's rule, tol %g" % tol) if lst is None else "list " + lst, list(map(int, r0["cg_iters"])), r0["chi2"][0], r0["chi2"][-1], list(map(int, r["cg_iters"])), r["chi2"][0], r["chi2"][-1], r["fallbacks"]), flush=True)

"""How tight must pcg_rel_tol be for the north star's 1e-6 bar (final chi^2 relative, vertices absolute) under round 3's stopping rule
(sqrt(r^T D^-1 r / b^T D^-1 b))?  Device runs at several tolerances against the device at 1e-13.  GPU box.
usage: python tools/research/tolerance_study.py [workload] [gn_iterations]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from toyslam_amd import synth
from toyslam_amd.optimizer import HipOptimizer


def vdiff(a, b, vt):
    d = np.abs(a - b); d[:, 2] = np.where(vt == 0, np.abs((a[:, 2] - b[:, 2] + np.pi) % (2 * np.pi) - np.pi), 0.0)
    return float(d.max())


name = sys.argv[1] if len(sys.argv) > 1 else "c2_10k"
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 50
g = synth.make_config(name)
runs = {}
for tol in (1e-13, 1e-10, 3e-10, 1e-9, 3e-9, 1e-8):
    o = HipOptimizer(pcg_rel_tol=tol)
    try:
        o.set_graph(g); r = o.optimize(n_it); runs[tol] = (r, o.vertices())
    finally:
        o.close()
ref, vref = runs[1e-13]
for tol, (r, v) in runs.items():
    print("%s %d GN iterations, tol %.0e: PCG iterations per solve %.2f, final chi2 rel diff %.2e, max vertex diff %.2e, stop %s/%d"
          % (name, n_it, tol, r["cg_iters"].mean(), abs(r["chi2_last"] / ref["chi2_last"] - 1), vdiff(v, vref, g.v_type), r["stop"], r["iters"]), flush=True)

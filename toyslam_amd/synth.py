"""Synthetic 2-D SLAM graphs for BASELINE.json configs 2-5 (generator: csrc/host/synth.cpp)."""
import ctypes as C

import numpy as np

from . import _lib
from .graph import GraphArrays, tsgo_graph

CONFIGS = {
    # name: (poses, LM edges per pose, loop closures)
    "c2_10k": (10_000, 10, 0),
    "c3_100k": (100_000, 10, 0),
    # config 5 is "1M poses / 10M edges (ODOM+LM mixed, loop closures)": 9 LM edges per pose = 9.0 M, the odometry chain 1.0 M and the
    # loop closures the walk offers (500 k asked for, ~0.12 M pose pairs closer than 2 m and 1000 steps apart exist): 10.1 M edges
    # (rounds 1-3 ran 8 per pose = 9.12 M and said so)
    "c5_1m": (1_000_000, 9, 500_000),
}


def make(n_poses, lm_per_pose=10, lm_obs_target=5.0, loop_closures=0, seed=0, with_truth=False):
    lib = _lib.host_lib()
    cfg = _lib.tsgo_synth_config(n_poses, lm_per_pose, lm_obs_target, loop_closures, seed)
    h = C.c_void_p()
    _lib.check(lib, lib.tsgo_synth_create(C.byref(cfg), C.byref(h)), "tsgo_synth_create")
    try:
        view = tsgo_graph()
        lib.tsgo_synth_view(h, C.byref(view))
        g = GraphArrays.from_c_struct(view)
        truth = np.ctypeslib.as_array(lib.tsgo_synth_truth(h), shape=(len(g.v_id), 3)).copy() if with_truth else None
    finally:
        lib.tsgo_synth_free(h)
    return (g, truth) if with_truth else g


def make_config(name, seed=0, **kw):
    p, k, lc = CONFIGS[name]
    return make(p, k, loop_closures=lc, seed=seed, **kw)

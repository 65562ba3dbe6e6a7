"""ctypes bindings of include/tsgo.h.  Loading fails loudly: there is no Python/CPU fallback."""
import ctypes as C
import os

from . import build
from .graph import tsgo_graph

TSGO_MAX_TRACE = 256


class tsgo_config(C.Structure):
    _fields_ = [("device", C.c_int32), ("precision", C.c_int32), ("pcg_rel_tol", C.c_double),
                ("pcg_max_iters", C.c_int32), ("lanes_per_pose", C.c_int32), ("lanes_per_lm", C.c_int32),
                ("use_graphs", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32), ("verbose", C.c_int32),
                ("preconditioner", C.c_int32), ("xcd_map", C.c_int32), ("warm_start", C.c_int32), ("rules", C.c_int32), ("lr", C.c_double), ("odom_jacobian", C.c_int32), ("reuse_structure", C.c_int32), ("cycle_level0", C.c_int32), ("cycle_storage", C.c_int32), ("warm_requests", C.c_int32)]


class tsgo_stats(C.Structure):
    _fields_ = [("iterations_run", C.c_int32), ("stop_reason", C.c_int32), ("chi2", C.c_double * TSGO_MAX_TRACE),
                ("pcg_iters", C.c_int32 * TSGO_MAX_TRACE), ("last_delta_norm", C.c_double),
                ("ms_total", C.c_double), ("ms_linearize", C.c_double), ("ms_solve", C.c_double),
                ("ms_update", C.c_double), ("ms_setup", C.c_double), ("structure_reused", C.c_int32), ("cycle_storage_now", C.c_int32), ("lambda_last", C.c_double), ("n_pose", C.c_int64), ("n_lm", C.c_int64),
                ("n_odom_edges", C.c_int64), ("n_lm_edges", C.c_int64), ("pcg_iters_total", C.c_int64),
                ("pcg_fallbacks", C.c_int32), ("trace_len", C.c_int32), ("chi2_last", C.c_double), ("history_carried", C.c_int32), ("graph_replay", C.c_int32)]


class tsgo_cycle_level(C.Structure):
    _fields_ = [("rows", C.c_int64), ("blocks", C.c_int64), ("sweeps_per_cycle", C.c_int32), ("lanes_per_row", C.c_int32),
                ("us_per_sweep", C.c_double), ("bytes_per_sweep", C.c_double)]


class tsgo_prof_entry(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("where", C.c_char * 32), ("launches_per_iteration", C.c_int32), ("reserved", C.c_int32),
                ("us", C.c_double), ("bytes", C.c_double)]


class tsgo_synth_config(C.Structure):
    _fields_ = [("n_poses", C.c_int64), ("lm_per_pose", C.c_int32), ("lm_obs_target", C.c_double),
                ("loop_closures", C.c_int32), ("seed", C.c_uint64)]


class tsgo_layout_info(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_pose", "n_lm_local", "n_lm_total", "n_lm_edges_local", "n_odom_slots",
                                         "rows_by_pose", "rows_by_lm", "rows_odom")] + \
               [("lanes_per_pose", C.c_int32), ("lanes_per_lm", C.c_int32)] + \
               [(n, C.c_int64) for n in ("lm_first", "lm_last", "pose_first", "pose_last")]


class tsgo_amg_info(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("rows", C.c_int64 * 8), ("blocks", C.c_int64 * 8), ("p_blocks", C.c_int64 * 8),
                ("schur_contribs", C.c_int64), ("ms_layout", C.c_double), ("ms_symbolic", C.c_double),
                ("agg_min", C.c_int32 * 8), ("agg_max", C.c_int32 * 8), ("checksum", C.c_uint64)]


HOST_SYMBOLS = ["tsgo_default_config", "tsgo_last_error", "tsgo_wire_decode", "tsgo_wire_new", "tsgo_wire_decode_into", "tsgo_wire_view", "tsgo_wire_free",
                "tsgo_wire_encode_response", "tsgo_wire_encode_request", "tsgo_synth_create", "tsgo_synth_view",
                "tsgo_synth_truth", "tsgo_synth_free", "tsgo_layout_probe", "tsgo_amg_probe", "tsgo_amg_probe_shard"]
DEVICE_SYMBOLS = ["tsgo_device_count", "tsgo_create", "tsgo_destroy", "tsgo_set_graph", "tsgo_reset_history", "tsgo_optimize", "tsgo_get_vertices",
                  "tsgo_linearize", "tsgo_solve_step", "tsgo_comm_unique_id", "tsgo_comm_init", "tsgo_comm_selftest", "tsgo_comm_time_allreduce", "tsgo_time_kernel", "tsgo_cycle_probe", "tsgo_profile_iteration"]
TESTING_SYMBOLS = ["tsgo_local_group_create", "tsgo_local_group_destroy", "tsgo_comm_init_local"]      # include/tsgo_testing.h: libtsgo_hip_testing.so only


def _declare_host(L):
    vp, u8p = C.c_void_p, C.POINTER(C.c_uint8)
    L.tsgo_default_config.argtypes = [C.POINTER(tsgo_config)]; L.tsgo_default_config.restype = None
    L.tsgo_last_error.restype = C.c_char_p
    L.tsgo_wire_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(vp)]
    L.tsgo_wire_new.argtypes = []; L.tsgo_wire_new.restype = vp
    L.tsgo_wire_decode_into.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.tsgo_wire_view.argtypes = [vp, C.POINTER(tsgo_graph)]; L.tsgo_wire_view.restype = None
    L.tsgo_wire_free.argtypes = [vp]; L.tsgo_wire_free.restype = None
    L.tsgo_wire_encode_response.argtypes = [vp, vp, vp, C.c_size_t]; L.tsgo_wire_encode_response.restype = C.c_int64
    L.tsgo_wire_encode_request.argtypes = [C.POINTER(tsgo_graph), vp, C.c_size_t]; L.tsgo_wire_encode_request.restype = C.c_int64
    L.tsgo_synth_create.argtypes = [C.POINTER(tsgo_synth_config), C.POINTER(vp)]
    L.tsgo_synth_view.argtypes = [vp, C.POINTER(tsgo_graph)]; L.tsgo_synth_view.restype = None
    L.tsgo_synth_truth.argtypes = [vp]; L.tsgo_synth_truth.restype = C.POINTER(C.c_double)
    L.tsgo_synth_free.argtypes = [vp]; L.tsgo_synth_free.restype = None
    L.tsgo_layout_probe.argtypes = [C.POINTER(tsgo_graph), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.POINTER(tsgo_layout_info)]
    L.tsgo_amg_probe.argtypes = [C.POINTER(tsgo_graph), C.POINTER(tsgo_amg_info)]
    L.tsgo_amg_probe_shard.argtypes = [C.POINTER(tsgo_graph), C.c_int32, C.c_int32, C.POINTER(tsgo_amg_info), C.POINTER(C.c_int64)]
    del u8p


def _declare_device(L):
    vp = C.c_void_p
    L.tsgo_create.argtypes = [C.POINTER(tsgo_config), C.POINTER(vp)]
    L.tsgo_destroy.argtypes = [vp]; L.tsgo_destroy.restype = None
    L.tsgo_set_graph.argtypes = [vp, C.POINTER(tsgo_graph)]
    L.tsgo_optimize.argtypes = [vp, C.c_int32, C.POINTER(tsgo_stats)]
    L.tsgo_reset_history.argtypes = [vp]; L.tsgo_reset_history.restype = None
    L.tsgo_get_vertices.argtypes = [vp, vp]
    L.tsgo_linearize.argtypes = [vp, vp, vp, C.POINTER(C.c_double)]
    L.tsgo_solve_step.argtypes = [vp, vp, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    L.tsgo_comm_unique_id.argtypes = [vp]
    L.tsgo_comm_init.argtypes = [vp, vp]
    L.tsgo_comm_selftest.argtypes = [vp, C.POINTER(C.c_int32)]
    L.tsgo_device_count.argtypes = []
    L.tsgo_comm_time_allreduce.argtypes = [vp, C.c_int64, C.c_int32, C.POINTER(C.c_double)]
    L.tsgo_time_kernel.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.tsgo_cycle_probe.argtypes = [vp, C.c_int32, C.POINTER(tsgo_cycle_level), C.c_int32]
    L.tsgo_profile_iteration.argtypes = [vp, C.c_int32, C.POINTER(tsgo_prof_entry), C.c_int32]


def _declare_testing(L):
    vp = C.c_void_p
    L.tsgo_local_group_create.argtypes = [C.c_int32, C.POINTER(vp)]
    L.tsgo_local_group_destroy.argtypes = [vp]; L.tsgo_local_group_destroy.restype = None
    L.tsgo_comm_init_local.argtypes = [vp, vp]


_host = None
_hip = None
_hip_testing = None


def host_lib():
    """libtsgo_host.so: codec, synthetic graphs, layout probe.  No GPU needed."""
    global _host
    if _host is None:
        # TSGO_HOST_SO: an instrumented build of the same sources (tools/sanitize_host.sh: ASan + UBSan, CPU only)
        path = os.environ.get("TSGO_HOST_SO") or (build.HOST_SO if os.path.exists(build.HOST_SO) else build.build_host())
        _host = C.CDLL(path)
        _declare_host(_host)
    return _host


def hip_lib():
    """libtsgo_hip.so: the device path.  Raises if the library is missing or cannot be loaded."""
    global _hip
    if _hip is None:
        if not os.path.exists(build.HIP_SO):
            raise RuntimeError("toyslam_amd: %s is missing — run __graft_entry__.build() (hipcc, gfx950). "
                               "There is no CPU fallback." % build.HIP_SO)
        _hip = C.CDLL(build.HIP_SO)
        _declare_host(_hip)
        _declare_device(_hip)
    return _hip


def hip_testing_lib():
    """libtsgo_hip_testing.so: the same sources built with -DTSGO_TESTING — test hooks, research variables and the in-process
    all-reduce group (include/tsgo_testing.h).  Loaded by the tests that need one of those; never by the product path."""
    global _hip_testing
    if _hip_testing is None:
        if not os.path.exists(build.HIP_TESTING_SO):
            raise RuntimeError("toyslam_amd: %s is missing — run __graft_entry__.build()" % build.HIP_TESTING_SO)
        _hip_testing = C.CDLL(build.HIP_TESTING_SO)
        _declare_host(_hip_testing)
        _declare_device(_hip_testing)
        _declare_testing(_hip_testing)
    return _hip_testing


def check(lib, rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, lib.tsgo_last_error().decode()))

"""Python host side of the device optimizer.

HipOptimizer wraps the C ABI one-to-one.  GraphOptimizer mirrors the reference's in-process
python/optimizer/graph_optimizer.py:11-92 (`GraphOptimizer(graph).optimize(iterations)`) but runs the
`cpu eigen` rules of remote/optimizer/OptimizerCpu.h:25-183 on the GPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from .graph import GraphArrays

STOP = {0: "cap", 1: "worse", 2: "plateau", 3: "converged", 4: "solver_failed"}


class HipOptimizer:
    def __init__(self, device=0, precision=64, pcg_rel_tol=1e-10, pcg_max_iters=20000, lanes_per_pose=0,
                 lanes_per_lm=0, use_graphs="auto", rank=0, world=1, preconditioner="amg", xcd_map=None, warm_start=None,
                 reuse_structure=None, rules="cpp", lr=0.2, odom_jacobian="constant", cycle_level0="implicit", cycle_storage=16, warm_requests=False, testing=False):
        # testing=True: libtsgo_hip_testing.so (the same sources with -DTSGO_TESTING: hooks, research variables, in-process group)
        # (TSGO_PY_TESTING_LIB=1: research scripts under tools/research and tests/research that set hook / research variables pick the
        # testing library without being edited; it selects which LIBRARY this Python wrapper loads, the product library reads nothing)
        import os
        self.lib = _lib.hip_testing_lib() if (testing or os.environ.get("TSGO_PY_TESTING_LIB") == "1") else _lib.hip_lib()
        cfg = _lib.tsgo_config()
        self.lib.tsgo_default_config(C.byref(cfg))
        cfg.device, cfg.precision, cfg.pcg_rel_tol, cfg.pcg_max_iters = device, precision, pcg_rel_tol, pcg_max_iters
        cfg.lanes_per_pose, cfg.lanes_per_lm, cfg.use_graphs = lanes_per_pose, lanes_per_lm, self._use_graphs(use_graphs)
        cfg.rank, cfg.world = rank, world
        cfg.preconditioner = {"jacobi": 0, "amg": 1}[preconditioner]
        if xcd_map is not None:
            cfg.xcd_map = int(xcd_map)
        if warm_start is not None:
            cfg.warm_start = int(warm_start)
        if reuse_structure is not None:
            cfg.reuse_structure = int(reuse_structure)
        cfg.rules, cfg.lr = {"cpp": 0, "python": 1}[rules], float(lr)
        cfg.odom_jacobian = {"constant": 0, "analytic": 1}[odom_jacobian]
        cfg.cycle_level0 = {"implicit": 0, "explicit": 1}[cycle_level0]
        cfg.cycle_storage = {16: 16, 32: 32}[cycle_storage]
        cfg.warm_requests = 1 if warm_requests else 0
        self.cfg = cfg
        self.h = C.c_void_p()
        _lib.check(self.lib, self.lib.tsgo_create(C.byref(cfg), C.byref(self.h)), "tsgo_create")
        self.n_vertices = 0
        self._v_in = None

    @staticmethod
    def _use_graphs(v):
        """tsgo_config.use_graphs: "auto" / 2 (eager while the host keeps ahead), True / 1 (replay), False / 0 (eager)."""
        if v == "auto":
            return 2
        if isinstance(v, bool):
            return int(v)
        if v in (0, 1, 2):
            return int(v)
        raise ValueError("use_graphs must be 'auto', True, False, 0, 1 or 2")

    def close(self):
        if self.h:
            self.lib.tsgo_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_graph(self, g: GraphArrays):
        cg = g.c_struct()
        _lib.check(self.lib, self.lib.tsgo_set_graph(self.h, C.byref(cg)), "tsgo_set_graph")
        self.n_vertices = len(g.v_id)
        self._v_in = g.v_pos.copy() if self.cfg.world > 1 else None    # a shard returns its own landmarks; the others keep their input

    def reset_history(self):
        """warm_requests: the next set_graph starts the solver from nothing (a pooled handle changing hands)."""
        self.lib.tsgo_reset_history(self.h)

    def optimize(self, iterations):
        st = _lib.tsgo_stats()
        _lib.check(self.lib, self.lib.tsgo_optimize(self.h, iterations, C.byref(st)), "tsgo_optimize")
        n = st.trace_len
        return dict(iters=st.iterations_run, stop=STOP[st.stop_reason], chi2=np.array(st.chi2[:n]), chi2_last=st.chi2_last,
                    cg_iters=np.array(st.pcg_iters[:n]), delta_norm=st.last_delta_norm, ms_total=st.ms_total,
                    ms_linearize=st.ms_linearize, ms_solve=st.ms_solve, ms_update=st.ms_update, ms_setup=st.ms_setup, structure_reused=bool(st.structure_reused), lambda_last=st.lambda_last,
                    n_pose=st.n_pose, n_lm=st.n_lm, n_odom_edges=st.n_odom_edges, n_lm_edges=st.n_lm_edges,
                    cg_total=st.pcg_iters_total, fallbacks=st.pcg_fallbacks, cycle_storage_now=st.cycle_storage_now, history_carried=st.history_carried, graph_replay=bool(st.graph_replay))

    def vertices(self):
        out = np.zeros((self.n_vertices, 3)) if self._v_in is None else np.ascontiguousarray(self._v_in.copy())
        _lib.check(self.lib, self.lib.tsgo_get_vertices(self.h, out.ctypes.data), "tsgo_get_vertices")
        return out

    def linearize(self):
        diag = np.zeros((self.n_vertices, 9)); grad = np.zeros((self.n_vertices, 3)); chi = C.c_double()
        _lib.check(self.lib, self.lib.tsgo_linearize(self.h, diag.ctypes.data, grad.ctypes.data, C.byref(chi)),
                   "tsgo_linearize")
        return diag, grad, chi.value

    def solve_step(self):
        d = np.zeros((self.n_vertices, 3)); chi = C.c_double(); it = C.c_int32()
        _lib.check(self.lib, self.lib.tsgo_solve_step(self.h, d.ctypes.data, C.byref(chi), C.byref(it)),
                   "tsgo_solve_step")
        return dict(delta=d, chi2=chi.value, cg_iters=it.value)

    def time_kernel(self, which, reps=50):
        us = C.c_double(); nbytes = C.c_double()
        _lib.check(self.lib, self.lib.tsgo_time_kernel(self.h, which, reps, C.byref(us), C.byref(nbytes)),
                   "tsgo_time_kernel")
        return us.value, nbytes.value

    def level_sweep_times(self, reps=100):
        """Per coarse level of the V-cycle: (us per smoothing sweep, algorithmic bytes per sweep, sweeps per cycle)."""
        arr = (_lib.tsgo_cycle_level * 16)()
        n = self.lib.tsgo_cycle_probe(self.h, reps, arr, 16)
        _lib.check(self.lib, min(n, 0), "tsgo_cycle_probe")
        return [(arr[k].us_per_sweep, arr[k].bytes_per_sweep, arr[k].sweeps_per_cycle) for k in range(n)]

    def profile_iteration(self, reps=20):
        """In-situ per-kernel timing of one PCG iteration: list of dict(name, where, launches, us, bytes) in launch order."""
        arr = (_lib.tsgo_prof_entry * 128)()
        n = self.lib.tsgo_profile_iteration(self.h, reps, arr, 128)
        _lib.check(self.lib, min(n, 0), "tsgo_profile_iteration")
        return [dict(name=arr[k].name.decode(), where=arr[k].where.decode(), launches=arr[k].launches_per_iteration, us=arr[k].us, bytes=arr[k].bytes)
                for k in range(n)]

    def comm_init_local(self, group):
        """group: a handle from local_group(world) shared by the handles of this process (one thread each)."""
        _lib.check(self.lib, self.lib.tsgo_comm_init_local(self.h, group), "tsgo_comm_init_local")

    def comm_unique_id(self):
        buf = (C.c_uint8 * 128)()
        _lib.check(self.lib, self.lib.tsgo_comm_unique_id(buf), "tsgo_comm_unique_id")
        return bytes(buf)

    def comm_init(self, uid: bytes):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        _lib.check(self.lib, self.lib.tsgo_comm_init(self.h, buf), "tsgo_comm_init")

    def comm_selftest(self):
        """One element through the solver's all-reduce; returns the communicator's own rank count (1 without one)."""
        n = C.c_int32()
        _lib.check(self.lib, self.lib.tsgo_comm_selftest(self.h, C.byref(n)), "tsgo_comm_selftest")
        return n.value

    def comm_time_allreduce(self, n_elements, reps=50):
        """Microseconds per all-reduce of n_elements numbers of the handle's precision on its communicator (every rank calls it alike)."""
        us = C.c_double()
        _lib.check(self.lib, self.lib.tsgo_comm_time_allreduce(self.h, int(n_elements), int(reps), C.byref(us)), "tsgo_comm_time_allreduce")
        return us.value


def local_group(world):
    """An in-process all-reduce group for `world` HipOptimizer(testing=True) handles (tests of the sharded path on a one-GPU box)."""
    lib = _lib.hip_testing_lib()
    g = C.c_void_p()
    _lib.check(lib, lib.tsgo_local_group_create(world, C.byref(g)), "tsgo_local_group_create")
    return g


def free_local_group(group):
    _lib.hip_testing_lib().tsgo_local_group_destroy(group)


class GraphOptimizer:
    """Same call shape as python/optimizer/graph_optimizer.py:11-20: GraphOptimizer(graph).optimize(n)."""

    def __init__(self, graph, **kw):
        self.graph = graph
        self.kw = kw
        self.last = None

    def optimize(self, iterations, lr=0.2):
        """rules="cpp" (default): the C++ server's loop, whose step is fixed at 0.2 (remote/optimizer/OptimizerCpu.h:164).
        rules="python": the reference's own GraphOptimizer.optimize(iterations, lr) — damping lambda*I, any lr."""
        kw = dict(self.kw)
        if kw.get("rules", "cpp") == "cpp":
            if lr != 0.2:
                raise ValueError("the remote optimizer's step is fixed at 0.2 (remote/optimizer/OptimizerCpu.h:164); pass rules=\"python\" for GraphOptimizer.optimize(iterations, lr)")
        else:
            kw["lr"] = lr
        arr = GraphArrays.from_optgraph(self.graph)
        opt = HipOptimizer(**kw)
        try:
            opt.set_graph(arr)
            self.last = opt.optimize(iterations)
            arr.write_back(self.graph, opt.vertices())
        finally:
            opt.close()
        return self.last

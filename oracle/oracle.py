"""ctypes access to the CPU oracle (oracle/libtsgo_oracle.so) — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
Nothing in toyslam_amd/ may.  Build the library with `make -C oracle` (or __graft_entry__.build()).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
ALLREDUCE_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int64, C.c_void_p)
_GRAPH = [C.c_int, u32p, u32p, f64p, C.c_int, u32p, u32p, f64p, f64p, C.c_int, u32p]


def build(force=False):
    so = os.path.join(_HERE, "libtsgo_oracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return so


def default_threads():
    """Threads for the twin's OpenMP loops: the process's CPU share, capped at 8 (a GPU box shows every
    logical CPU of the host; oversubscribed OpenMP barriers spin for milliseconds)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(8, n))


def set_threads(n):
    lib().oracle_set_threads(int(n))


def set_odom_jacobian(kind):
    """"constant" (the reference: A = -I, B = +I, EdgeSe2.h:35-37) or "analytic" (the extension of SURVEY 8f rank 4).  Process-wide:
    every oracle entry point, dense and twin, follows it.  Tests that switch it switch it back."""
    lib().oracle_set_odom_jacobian({"constant": 0, "analytic": 1}[kind])


def set_cycle_level0(kind):
    """"implicit" (default) or "explicit": what the twin's two products inside the multigrid cycle read (tsgo_config.cycle_level0)."""
    lib().oracle_set_cycle_level0({"implicit": 0, "explicit": 1}[kind])


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        for sfx in ("f64", "f32"):
            getattr(L, "oracle_edge_eval_" + sfx).argtypes = _GRAPH + [f64p, f64p, f64p]
            getattr(L, "oracle_linearize_" + sfx).argtypes = _GRAPH + [C.c_int, C.c_void_p, f64p, f64p, i32p]
            getattr(L, "oracle_optimize_" + sfx).argtypes = _GRAPH + [f64p, C.c_int, C.c_int, C.c_int, C.c_double,
                                                                      f64p, i32p, i32p, f64p]
        L.oracle_solve_f64.argtypes = [f64p, f64p, C.c_int, C.c_int]
        L.oracle_update_pose_f64.argtypes = [f64p, f64p]
        L.oracle_last_phase_seconds.argtypes = [f64p]; L.oracle_last_phase_seconds.restype = None
        L.oracle_num_unknowns.argtypes = [C.c_int, u32p]
        L.oracle_sparse_step.argtypes = _GRAPH + [C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, ALLREDUCE_FN, C.c_void_p,
                                                  f64p, f64p, i32p]
        L.oracle_sparse_optimize.argtypes = _GRAPH + [f64p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                                      ALLREDUCE_FN, C.c_void_p, f64p, i32p, i32p, i32p, f64p,
                                                      f64p, f64p, C.c_int, C.c_double]
        L.oracle_set_threads.argtypes = [C.c_int]
        L.oracle_set_threads.restype = None
        L.oracle_set_threads(default_threads())
        _LIB = L
    return _LIB


class Graph:
    """SoA OptGraph as the oracle consumes it (doubles on the host side).

    v_pos[i] = (x, y, theta) for poses, (x, y, 0) for landmarks; e_meas[i] = 3x3 row-major for ODOM,
    (range, bearing, 0...) for LM; e_inf[i] = diagonal of the information matrix (3 slots)."""

    def __init__(self, v_id, v_type, v_pos, e_type, e_ids, e_meas, e_inf, fixed):
        self.v_id = np.ascontiguousarray(v_id, np.uint32)
        self.v_type = np.ascontiguousarray(v_type, np.uint32)
        self.v_pos = np.ascontiguousarray(v_pos, np.float64).reshape(-1, 3)
        self.e_type = np.ascontiguousarray(e_type, np.uint32)
        self.e_ids = np.ascontiguousarray(e_ids, np.uint32).reshape(-1, 2)
        self.e_meas = np.ascontiguousarray(e_meas, np.float64).reshape(-1, 9)
        self.e_inf = np.ascontiguousarray(e_inf, np.float64).reshape(-1, 3)
        self.fixed = np.ascontiguousarray(fixed, np.uint32)

    @classmethod
    def from_npz(cls, z, as_wire=False):
        g = cls(z["v_id"], z["v_type"], z["v_pos"], z["e_type"], z["e_ids"], z["e_meas"], z["e_inf"], z["fixed"])
        return g.rounded_to_wire() if as_wire else g

    def rounded_to_wire(self):
        r = lambda a: a.astype(np.float32).astype(np.float64)  # noqa: E731
        return Graph(self.v_id, self.v_type, r(self.v_pos), self.e_type, self.e_ids, r(self.e_meas),
                     r(self.e_inf), self.fixed)

    def copy(self):
        return Graph(self.v_id.copy(), self.v_type.copy(), self.v_pos.copy(), self.e_type.copy(),
                     self.e_ids.copy(), self.e_meas.copy(), self.e_inf.copy(), self.fixed.copy())

    @property
    def n_unknowns(self):
        return int(3 * (self.v_type == 0).sum() + 2 * (self.v_type == 1).sum())

    def args(self):
        fx = self.fixed if len(self.fixed) else np.zeros(1, np.uint32)
        return [len(self.v_id), self.v_id, self.v_type, self.v_pos, len(self.e_type), self.e_type, self.e_ids,
                self.e_meas, self.e_inf, len(self.fixed), fx]


def _sfx(precision):
    return {"f64": "f64", "f32": "f32", 64: "f64", 32: "f32"}[precision]


def edge_eval(g, precision="f64"):
    E = len(g.e_type)
    e = np.zeros((E, 3)); A = np.zeros((E, 9)); B = np.zeros((E, 9))
    rc = getattr(lib(), "oracle_edge_eval_" + _sfx(precision))(*g.args(), e, A, B)
    if rc:
        raise RuntimeError("oracle_edge_eval rc=%d" % rc)
    return e, A, B


def linearize(g, python_mode=False, precision="f64", want_H=True):
    n = g.n_unknowns
    H = np.zeros((n, n)) if want_H else None
    b = np.zeros(n); err = np.zeros(1); idx = np.zeros(len(g.v_id), np.int32)
    rc = getattr(lib(), "oracle_linearize_" + _sfx(precision))(
        *g.args(), int(python_mode), H.ctypes.data if want_H else None, b, err, idx)
    if rc < 0:
        raise RuntimeError("oracle_linearize rc=%d" % rc)
    return H, b, float(err[0]), idx


STOP = {0: "cap", 1: "worse", 2: "plateau", 3: "converged", 4: "solver_failed"}


def optimize(g, iterations, mode="cpp", solver="chol", lr=0.2, precision="f64"):
    """Dense `cpu eigen` restatement.  Returns dict(v_pos, chi2, iters, stop, delta_norm)."""
    out = np.zeros_like(g.v_pos); chi2 = np.zeros(max(iterations, 1))
    ir = np.zeros(1, np.int32); sr = np.zeros(1, np.int32); dn = np.zeros(1)
    rc = getattr(lib(), "oracle_optimize_" + _sfx(precision))(
        *g.args(), out, iterations, {"cpp": 0, "python": 1, "cpp_on_python_linearisation": 2}[mode], {"qr": 0, "chol": 1}[solver], lr, chi2, ir, sr, dn)
    if rc:
        raise RuntimeError("oracle_optimize rc=%d" % rc)
    return dict(v_pos=out, chi2=chi2[:ir[0]].copy(), iters=int(ir[0]), stop=STOP[int(sr[0])], delta_norm=float(dn[0]))


def last_phase_seconds():
    """(linearise, dense solve, vertex update) wall seconds of this thread's last optimize() call."""
    out = np.zeros(3)
    lib().oracle_last_phase_seconds(out)
    return out


def solve(H, b, solver="qr"):
    h = np.ascontiguousarray(H, np.float64).copy(); x = np.ascontiguousarray(b, np.float64).copy()
    rc = lib().oracle_solve_f64(h, x, len(x), {"qr": 0, "chol": 1}[solver])
    if rc:
        raise RuntimeError("oracle_solve rc=%d" % rc)
    return x


def update_pose(xyt, d):
    x = np.ascontiguousarray(xyt, np.float64).copy()
    lib().oracle_update_pose_f64(x, np.ascontiguousarray(d, np.float64))
    return x


def _hook(allreduce):
    """Wrap a python callable(np.ndarray) -> None (in-place sum over shards) as the C hook."""
    if allreduce is None:
        return C.cast(None, ALLREDUCE_FN)

    def cb(ptr, n, _ctx):
        allreduce(np.ctypeslib.as_array(ptr, shape=(n,)))
    return ALLREDUCE_FN(cb)


PRECOND = {"jacobi": 0, "amg": 1}


def sparse_step(g, pcg_tol=1e-12, max_cg=100000, rank=0, world=1, allreduce=None, precond="jacobi"):
    """One GN step by the sparse CPU twin (implicit-Schur PCG): dict(delta, chi2, cg_iters)."""
    d = np.zeros((len(g.v_id), 3)); chi = np.zeros(1); it = np.zeros(1, np.int32)
    h = _hook(allreduce)
    rc = lib().oracle_sparse_step(*g.args(), pcg_tol, max_cg, PRECOND[precond], rank, world, h, None, d, chi, it)
    if rc:
        raise RuntimeError("oracle_sparse_step rc=%d" % rc)
    return dict(delta=d, chi2=float(chi[0]), cg_iters=int(it[0]))


def sparse_optimize(g, iterations, pcg_tol=1e-12, max_cg=100000, rank=0, world=1, allreduce=None, precond="jacobi", rules="cpp", lr=0.2):
    """Full GN loop by the sparse CPU twin with the reference's stop rules (rules="python": the loop of the reference's
    in-process Python optimizer instead — lambda * I, step lr)."""
    out = np.zeros_like(g.v_pos); chi2 = np.zeros(max(iterations, 1)); cg = np.zeros(max(iterations, 1), np.int32)
    ir = np.zeros(1, np.int32); sr = np.zeros(1, np.int32); dn = np.zeros(1); tl = np.zeros(1); ts = np.zeros(1)
    h = _hook(allreduce)
    rc = lib().oracle_sparse_optimize(*g.args(), out, iterations, pcg_tol, max_cg, PRECOND[precond], rank, world, h, None, chi2, ir, sr,
                                      cg, dn, tl, ts, {"cpp": 0, "python": 1}[rules], float(lr))
    if rc:
        raise RuntimeError("oracle_sparse_optimize rc=%d" % rc)
    n = int(ir[0])
    return dict(v_pos=out, chi2=chi2[:n].copy(), iters=n, stop=STOP[int(sr[0])], delta_norm=float(dn[0]),
                cg_iters=cg[:n].copy(), seconds_linearize=float(tl[0]), seconds_solve=float(ts[0]))

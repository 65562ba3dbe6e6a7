"""The OptGraph model on the Python host side.

Two views of the same thing:
  * OptGraph / VertexPose2d / Vertex2d / EdgeOdometry2d / EdgeLandmark2d (+ EdgeVirtualLandmark2d, the reference's commented-out sketch) — same names, constructor
    arguments and get_type()/get_dims() values as the reference's python/optimizer/opt_graph.py:1-31,
    vertices.py:18-46 and edges2d.py:14-81, so code written against those keeps working;
  * GraphArrays — the structure-of-arrays form the C ABI takes (include/tsgo.h, struct tsgo_graph).
"""
import ctypes as C

import numpy as np


class VertexPose2d:
    """SE(2) pose; `position` is the 3x3 homogeneous matrix (vertices.py:18-33)."""

    def __init__(self, position):
        self.position = np.asarray(position, dtype=np.float64)

    def get_type(self):
        return 0

    def get_dims(self):
        return 3


class Vertex2d:
    """2-D landmark; `position` is (x, y) (vertices.py:35-46)."""

    def __init__(self, position):
        self.position = np.asarray(position, dtype=np.float64)

    def get_type(self):
        return 1

    def get_dims(self):
        return 2


class _Edge:
    def __init__(self, id_1, id_2, measurement, information):
        self.id_1, self.id_2 = id_1, id_2
        self.measurement = np.asarray(measurement, dtype=np.float64)
        self.information = np.asarray(information, dtype=np.float64)

    def get_id(self, index):
        return self.id_1 if index == 0 else self.id_2


class EdgeOdometry2d(_Edge):
    """ODOM edge: measurement is the 3x3 relative transform (edges2d.py:57-81)."""

    def get_type(self):
        return 0


class EdgeLandmark2d(_Edge):
    """LM edge: measurement is (range, bearing) (edges2d.py:14-55)."""

    def get_type(self):
        return 1


class EdgeVirtualLandmark2d:
    """Virtual landmark measurement (edge type 2, include/tsgo.h): two poses that observed the same physical point, no landmark vertex.
    Constructor arguments as in the sketch the reference keeps commented out (python/optimizer/edges2d.py:83-89):
    (pos_id_1, pos_id_2, lm_meas_1, lm_meas_2, information) with lm_meas_k = (range, bearing) as seen from pose k and a 2 x 2
    information matrix (its diagonal is used, as for every other edge).  Behind the C ABI only: remote.graph_to_bytes refuses it."""

    def __init__(self, pos_id_1, pos_id_2, lm_meas_1, lm_meas_2, information):
        self.id_1, self.id_2 = pos_id_1, pos_id_2
        self.pos_id_1, self.pos_id_2 = pos_id_1, pos_id_2
        self.lm_meas_1 = np.asarray(lm_meas_1, dtype=np.float64)
        self.lm_meas_2 = np.asarray(lm_meas_2, dtype=np.float64)
        self.measurement = np.concatenate([self.lm_meas_1[:2], self.lm_meas_2[:2]])
        self.information = np.asarray(information, dtype=np.float64)

    def get_type(self):
        return 2

    def get_id(self, index):
        return self.id_1 if index == 0 else self.id_2


class OptGraph:
    """Container the optimizers consume: id -> vertex (insertion-ordered), a list of edges, the set of gauge-fixed ids.

    Interface-compatible with the reference's python/optimizer/opt_graph.py (the accessor names are what
    graph_to_bytes.py:43-64 and slam_main.py:157-211 call); the body is this package's own.  Unknown ids raise
    KeyError-derived UnknownVertex (a RuntimeError too, which is what the reference's callers catch)."""

    class UnknownVertex(KeyError, RuntimeError):
        def __str__(self):
            return "vertex id %r is not in the graph" % (self.args[0],)

    __slots__ = ("_v", "_e", "_fixed")

    def __init__(self, vertices=None, edges=(), fixed=()):
        self._v = dict(vertices or {})
        self._e = list(edges)
        self._fixed = set()
        for i in fixed:
            self.fix_vertex(i)

    def _known(self, vid):
        try:
            return self._v[vid]
        except KeyError:
            raise OptGraph.UnknownVertex(vid) from None

    # ---- building ----
    def add_vertex(self, id, vertex, fixed=False):
        self._v[id] = vertex
        if fixed:
            self._fixed.add(id)

    def add_edge(self, edge):
        self._e.append(edge)

    def fix_vertex(self, id):
        self._known(id)
        self._fixed.add(id)

    # ---- reading ----
    def get_vertex(self, id):
        return self._known(id)

    def get_vertices(self):
        return self._v

    def get_edges(self):
        return self._e

    def get_fixed_vertices(self):
        return self._fixed

    # attribute spellings some callers of the reference use directly
    vertices = property(get_vertices)
    edges = property(get_edges)
    fixed_vertices = property(get_fixed_vertices)

    def __len__(self):
        return len(self._v)


class tsgo_graph(C.Structure):
    _fields_ = [("n_vertices", C.c_int32), ("v_id", C.c_void_p), ("v_type", C.c_void_p), ("v_pos", C.c_void_p),
                ("n_edges", C.c_int32), ("e_type", C.c_void_p), ("e_ids", C.c_void_p), ("e_meas", C.c_void_p),
                ("e_inf", C.c_void_p), ("n_fixed", C.c_int32), ("fixed", C.c_void_p)]


class GraphArrays:
    """SoA form of an OptGraph (struct tsgo_graph).  v_pos: (x, y, theta) / (x, y, 0); e_meas: 3x3
    row-major for ODOM, (range, bearing, 0..) for LM; e_inf: information diagonal (3 slots)."""

    def __init__(self, v_id, v_type, v_pos, e_type, e_ids, e_meas, e_inf, fixed):
        self.v_id = np.ascontiguousarray(v_id, np.uint32)
        self.v_type = np.ascontiguousarray(v_type, np.uint32)
        self.v_pos = np.ascontiguousarray(v_pos, np.float64).reshape(-1, 3)
        self.e_type = np.ascontiguousarray(e_type, np.uint32)
        self.e_ids = np.ascontiguousarray(e_ids, np.uint32).reshape(-1, 2)
        self.e_meas = np.ascontiguousarray(e_meas, np.float64).reshape(-1, 9)
        self.e_inf = np.ascontiguousarray(e_inf, np.float64).reshape(-1, 3)
        self.fixed = np.ascontiguousarray(fixed, np.uint32)
        assert len(self.v_id) == len(self.v_type) == len(self.v_pos)
        assert len(self.e_type) == len(self.e_ids) == len(self.e_meas) == len(self.e_inf)

    @property
    def n_poses(self):
        return int((self.v_type == 0).sum())

    @property
    def n_landmarks(self):
        return int((self.v_type == 1).sum())

    @property
    def n_edges(self):
        return len(self.e_type)

    def rounded_to_wire(self):
        """Every float rounded to f32 — what a server receives (graph_to_bytes.py:4-7)."""
        r = lambda a: a.astype(np.float32).astype(np.float64)  # noqa: E731
        return GraphArrays(self.v_id, self.v_type, r(self.v_pos), self.e_type, self.e_ids, r(self.e_meas),
                           r(self.e_inf), self.fixed)

    def copy(self):
        return GraphArrays(self.v_id.copy(), self.v_type.copy(), self.v_pos.copy(), self.e_type.copy(),
                           self.e_ids.copy(), self.e_meas.copy(), self.e_inf.copy(), self.fixed.copy())

    def c_struct(self):
        """ctypes view; keeps the arrays alive through the returned object's _owner."""
        g = tsgo_graph(len(self.v_id), self.v_id.ctypes.data, self.v_type.ctypes.data, self.v_pos.ctypes.data,
                       len(self.e_type), self.e_type.ctypes.data, self.e_ids.ctypes.data, self.e_meas.ctypes.data,
                       self.e_inf.ctypes.data, len(self.fixed), self.fixed.ctypes.data if len(self.fixed) else None)
        g._owner = self
        return g

    @classmethod
    def from_c_struct(cls, g):
        def arr(ptr, n, dt):
            if n == 0 or not ptr:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(n,)).copy()
        nV, nE, nF = g.n_vertices, g.n_edges, g.n_fixed
        return cls(arr(g.v_id, nV, np.uint32), arr(g.v_type, nV, np.uint32), arr(g.v_pos, 3 * nV, np.float64),
                   arr(g.e_type, nE, np.uint32), arr(g.e_ids, 2 * nE, np.uint32), arr(g.e_meas, 9 * nE, np.float64),
                   arr(g.e_inf, 3 * nE, np.float64), arr(g.fixed, nF, np.uint32))

    @classmethod
    def from_optgraph(cls, graph):
        """Flatten an OptGraph in dict / list order (the order graph_to_bytes.py:43-64 walks)."""
        vid, vtype, vpos = [], [], []
        for k, v in graph.get_vertices().items():
            vid.append(k); vtype.append(v.get_type())
            p = np.asarray(v.position, dtype=np.float64)
            if v.get_type() == 0:
                vpos.append([p[0, 2], p[1, 2], np.arctan2(p[1, 0], p[0, 0])])
            else:
                vpos.append([p[0], p[1], 0.0])
        etype, eids, emeas, einf = [], [], [], []
        for e in graph.get_edges():
            m = np.zeros(9); w = np.zeros(3)
            meas = np.asarray(e.measurement, dtype=np.float64); inf = np.asarray(e.information, dtype=np.float64)
            if e.get_type() == 0:
                m[:] = meas.reshape(-1); w[:] = np.diag(inf)
            elif e.get_type() == 2:
                m[:4] = meas.reshape(-1)[:4]; w[:2] = np.diag(inf)[:2]
            else:
                m[:2] = meas.reshape(-1)[:2]; w[:2] = np.diag(inf)[:2]
            etype.append(e.get_type()); eids.append([e.id_1, e.id_2]); emeas.append(m); einf.append(w)
        return cls(np.array(vid, np.uint32), np.array(vtype, np.uint32), np.array(vpos).reshape(-1, 3),
                   np.array(etype, np.uint32), np.array(eids, np.uint32).reshape(-1, 2),
                   np.array(emeas).reshape(-1, 9), np.array(einf).reshape(-1, 3),
                   np.array(list(graph.get_fixed_vertices()), np.uint32))

    def write_back(self, graph, v_pos):
        """Copy optimised vertex positions into an OptGraph (what python/slam_main.py:196-211 consumes)."""
        for k, p in zip(self.v_id, np.asarray(v_pos).reshape(-1, 3)):
            v = graph.get_vertex(int(k))
            if v.get_type() == 0:
                c, s = np.cos(p[2]), np.sin(p[2])
                v.position = np.array([[c, -s, p[0]], [s, c, p[1]], [0, 0, 1.0]])
            else:
                v.position = np.array([p[0], p[1]])

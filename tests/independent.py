"""A checker for the BIG configurations that shares no code with the product.

The CPU twin (oracle/oracle_sparse.cpp) reuses the product's slot-table builder and per-edge arithmetic, so agreement
with it above the sizes the dense restatement reaches cannot expose a mistake the two have in common.  This module
uses ONLY oracle/oracle_dense.cpp's own per-edge functions (lm_edge / odom_edge, exported as oracle_edge_eval_f64 —
the restatement of remote/graph/edge/EdgeSe2Point2d.h:27-70 and EdgeSe2.h:23-38, O(E)) and numpy:

  chi2, gradient b, diagonal blocks of H      as OptimizerCpu.h:88-119,132-138 accumulates them
  H @ delta, matrix-free                      J^T (Omega_w (J delta)) + gauge, edge by edge

so that a device linearisation and a device solve can be checked at 100k and 1M poses:  ||H delta - b|| / ||b||.

Round 4: the UPDATE and the STOP RULES too (apply_update, GnRules), so that a multi-iteration trajectory at 100k poses can be followed
step by step without the twin: vertices after an iteration = the reference's own update rule applied to the device's delta, and the
loop's decisions = the reference's rules evaluated on the checker's own chi^2 and ||delta||.
"""
import numpy as np

from oracle import oracle

HUBER_DELTA = 1.5          # remote/optimizer/OptimizerCpu.h:92
GAUGE = 1e6                # :136


class Linearisation:
    """Per-edge Jacobians and Huber-scaled weights of one graph at its current vertex positions."""

    def __init__(self, g):
        o = oracle.Graph(g.v_id, g.v_type, g.v_pos, g.e_type, g.e_ids, g.e_meas, g.e_inf, g.fixed)
        e, A, B = oracle.edge_eval(o)
        E, V = len(g.e_type), len(g.v_id)
        lm = g.e_type == 1
        # uniform 3x3 blocks: an LM edge's 2x3 / 2x2 Jacobians sit in the top rows (third residual row = 0)
        A3 = A.reshape(E, 3, 3).copy(); B3 = B.reshape(E, 3, 3).copy()
        A3[lm] = 0; B3[lm] = 0
        A3[lm, :2, :] = A[lm, :6].reshape(-1, 2, 3)
        B3[lm, :2, :2] = B[lm, :4].reshape(-1, 2, 2)
        vl = g.e_type == 2                                              # virtual landmark measurement: 2 x 3 and 2 x 3 (oracle_dense.cpp: vlm_edge)
        A3[vl] = 0; B3[vl] = 0
        A3[vl, :2, :] = A[vl, :6].reshape(-1, 2, 3)
        B3[vl, :2, :] = B[vl, :6].reshape(-1, 2, 3)
        two = lm | vl
        w = g.e_inf.copy(); w[two, 2] = 0
        e = e.copy(); e[two, 2] = 0
        chi = (w * e * e).sum(axis=1)                                   # e^T Omega e  (:91)
        tail = chi > HUBER_DELTA ** 2                                   # robustify (:36-46)
        sq = np.sqrt(np.where(tail, chi, 1.0))
        rho = np.where(tail, 2 * sq * HUBER_DELTA - HUBER_DELTA ** 2, chi)
        scale = np.where(tail, HUBER_DELTA / sq, 1.0)
        self.n_tail = int(tail.sum())
        self.chi2 = float(rho.sum())                                    # err += rho (:117)
        self.w = w * scale[:, None]                                     # Omega_w (:93)
        self.A, self.B, self.e = A3, B3, e
        order = np.argsort(g.v_id, kind="stable"); ids = g.v_id[order]
        self.i1 = order[np.searchsorted(ids, g.e_ids[:, 0])]
        self.i2 = order[np.searchsorted(ids, g.e_ids[:, 1])]
        self.V = V
        self.gauge = np.zeros(V)
        for f in g.fixed:                                               # once per occurrence (:132-138)
            self.gauge[order[np.searchsorted(ids, f)]] += GAUGE
        self.dims = np.where(g.v_type == 0, 3, 2)

    def _scatter(self, idx, vals):
        out = np.zeros((self.V,) + vals.shape[1:])
        flat = out.reshape(self.V, -1); v = vals.reshape(len(idx), -1)
        for k in range(v.shape[1]):
            flat[:, k] = np.bincount(idx, weights=v[:, k], minlength=self.V)
        return out

    def gradient(self):
        """b = -sum J^T Omega_w e per vertex, (V, 3) (AddSegment subtracts: MatrixEigen.h:93-96)."""
        we = self.w * self.e
        return -(self._scatter(self.i1, np.einsum("eki,ek->ei", self.A, we)) +
                 self._scatter(self.i2, np.einsum("eki,ek->ei", self.B, we)))

    def diag_blocks(self):
        """Diagonal 3x3 blocks of H incl. the gauge term, (V, 9) row-major; landmarks use the leading 2x2."""
        d = (self._scatter(self.i1, np.einsum("eki,ek,ekj->eij", self.A, self.w, self.A)) +
             self._scatter(self.i2, np.einsum("eki,ek,ekj->eij", self.B, self.w, self.B)))
        for k in range(3):
            d[:, k, k] += np.where(k < self.dims, self.gauge, 0.0)
        return d.reshape(self.V, 9)

    def apply_H(self, delta):
        """H @ delta for a per-vertex (V, 3) delta (third component of a landmark ignored)."""
        d = np.where(np.arange(3)[None, :] < self.dims[:, None], delta, 0.0)
        jd = np.einsum("eki,ei->ek", self.A, d[self.i1]) + np.einsum("eki,ei->ek", self.B, d[self.i2])
        wjd = self.w * jd
        out = (self._scatter(self.i1, np.einsum("eki,ek->ei", self.A, wjd)) +
               self._scatter(self.i2, np.einsum("eki,ek->ei", self.B, wjd)))
        return out + self.gauge[:, None] * d

    def residual_of(self, delta):
        """||H delta - b|| / ||b||."""
        b = self.gradient()
        r = self.apply_H(delta) - b
        mask = np.arange(3)[None, :] < self.dims[:, None]
        return float(np.linalg.norm(r[mask]) / np.linalg.norm(b[mask]))


STEP = 0.2                 # remote/optimizer/OptimizerCpu.h:164
PLATEAU_TOL = 1e-3         # :167
DELTA_TOL = 1e-3           # :173


def apply_update(v_pos, v_type, delta, step=STEP):
    """The reference's vertex update with the loop's fixed step (OptimizerCpu.h:159-165):
    VertexSe2::Update (remote/graph/vertex/VertexSe2.h:16-27): theta <- atan2(R10, R00) + step * d_theta, translation added in the
    WORLD frame (no R * d); Vertex2d::Update (vertex/Vertex2d.h:16-19): p <- p + step * d.  (x, y, theta) in and out."""
    out = np.array(v_pos, dtype=np.float64, copy=True)
    pose = v_type == 0
    th = np.arctan2(np.sin(out[pose, 2]), np.cos(out[pose, 2])) + step * delta[pose, 2]
    out[pose, 0] += step * delta[pose, 0]; out[pose, 1] += step * delta[pose, 1]
    out[pose, 2] = np.arctan2(np.sin(th), np.cos(th))              # what a read-out of the rotation block gives (SerializeGraphFuncCpu.h:28)
    lm = ~pose
    out[lm, 0] += step * delta[lm, 0]; out[lm, 1] += step * delta[lm, 1]
    return out


def delta_norm(delta, v_type):
    """||delta||_2 over the unknowns (3 per pose, 2 per landmark), unscaled (OptimizerCpu.h:173)."""
    mask = np.arange(3)[None, :] < np.where(v_type == 0, 3, 2)[:, None]
    return float(np.sqrt((delta[mask] ** 2).sum()))


class GnRules:
    """The loop rules of OptimizerCpu.h:140-153,167-179, fed one iteration at a time with the checker's own numbers:
    before_solve(chi2) -> "worse" or None (penalty > 2 on rising chi^2: break BEFORE the solve);
    after_update(chi2, norm) -> "plateau" / "converged" / None (plateau first, then the short step)."""

    def __init__(self):
        self.prev = -1.0
        self.penalty = 0

    def before_solve(self, chi2):
        if self.prev > 0 and chi2 > self.prev:
            self.penalty += 1
            if self.penalty > 2:
                return "worse"
        else:
            self.penalty = 0
        return None

    def after_update(self, chi2, norm):
        if abs(chi2 - self.prev) < PLATEAU_TOL:
            return "plateau"
        if norm < DELTA_TOL:
            return "converged"
        self.prev = chi2
        return None

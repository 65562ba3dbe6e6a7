#!/usr/bin/env python3
"""Round-3 golden fixtures, produced by IMPORTING the reference's Python modules (authoring container only; needs
/root/reference, read-only).  Companion of make_golden.py: it starts from the fixtures that script wrote (c1_graph.npz,
tiny_*.npz — the config-1 sim itself is not replayed again) and writes only data.

  c1_reply_ref.npz, tiny_{a,b,c}_reply_ref.npz
      A REPLY encoded by this repo's codec (csrc/host/codec.cpp through toyslam_amd.remote.encode_response) is handed to the
      reference's own reader, python/remote/bytes_to_graph.py:49-108; what that reader returns (vertex ids in its order,
      pose matrices / landmark vectors, edge types, ids, measurement and information matrices, fixed set) is stored together
      with the SHA-1 of the reply bytes it was given.  tests/test_codec.py re-encodes the same reply, checks the SHA-1 and
      compares its own reading with the reference's.

  c1_cpprules_ref.npz
      The `cpu eigen` loop of remote/optimizer/OptimizerCpu.h:80-180 (no lambda, step 0.2, penalty / plateau / short-step
      stops) DRIVEN WITH THE REFERENCE'S OWN ARITHMETIC: GraphOptimizer.calculate_H_b (python/optimizer/graph_optimizer.py:94-155)
      for err, H, b; numpy.linalg.solve for the dense solve; VertexPose2d.update / Vertex2d.update (vertices.py:28-33,45-46)
      for the step.  Only the loop control (a dozen lines below, each citing the C++ line it follows) is written here.
      calculate_H_b zeroes b at fixed vertices (:150), which OptimizerCpu.h:137 does not: the oracle's mode 2 ("cpp rules on
      the Python linearisation") is what reproduces this fixture; mode 0 differs from it only by that one assignment, which
      test_c1_linearisation_matches_reference_python pins separately.
"""
import contextlib
import hashlib
import io
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
REF = "/root/reference/python"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)

import numpy as np  # noqa: E402

from optimizer.graph_optimizer import GraphOptimizer  # noqa: E402
from remote.bytes_to_graph import bytes_to_graph  # noqa: E402
from remote.graph_to_bytes import graph_to_bytes  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402  (graph_from_soa / graph_soa helpers; importing it runs nothing)

OUT = os.path.dirname(os.path.abspath(__file__))


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def reply_vertices(v_pos, v_type):
    """The vertex positions put into the reply: a deterministic, f32-exact perturbation of the request's."""
    v = np.asarray(v_pos, np.float64).copy()
    k = np.arange(len(v), dtype=np.float64)
    v[:, 0] += 0.125 + 0.5 * (k % 7); v[:, 1] -= 0.25 * (k % 5)
    v[:, 2] = np.where(np.asarray(v_type) == 0, ((v[:, 2] + 0.0625 * (k % 9) + np.pi) % (2 * np.pi)) - np.pi, 0.0)
    return v.astype(np.float32).astype(np.float64)


def parsed_reply(reply_payload):
    g = bytes_to_graph(reply_payload)                     # the reference's reader
    vid = np.array(list(g.vertices.keys()), np.uint32)
    vtype = np.array([g.vertices[k].get_type() for k in g.vertices], np.uint32)
    vmat = np.zeros((len(vid), 9), np.float32)            # pose: the 3x3 it builds; landmark: (x, y)
    for i, k in enumerate(g.vertices):
        p = np.asarray(g.vertices[k].position, np.float32).reshape(-1)
        vmat[i, :p.size] = p
    etype = np.array([e.get_type() for e in g.edges], np.uint32)
    eids = np.array([[e.id_1, e.id_2] for e in g.edges], np.uint32).reshape(-1, 2)
    mshape = np.array([np.asarray(e.measurement).shape + (1,) * (2 - np.asarray(e.measurement).ndim) for e in g.edges], np.int32).reshape(-1, 2)
    meas = np.zeros((len(g.edges), 9), np.float32); inf = np.zeros((len(g.edges), 9), np.float32)
    ishape = np.zeros((len(g.edges), 2), np.int32)
    for i, e in enumerate(g.edges):
        m = np.asarray(e.measurement, np.float32).reshape(-1); meas[i, :m.size] = m
        w = np.asarray(e.information, np.float32); ishape[i] = w.shape; inf[i, :w.size] = w.reshape(-1)
    return dict(v_id=vid, v_type=vtype, v_mat=vmat, e_type=etype, e_ids=eids, e_meas_shape=mshape, e_meas=meas,
                e_inf_shape=ishape, e_inf=inf, fixed=np.array(sorted(g.fixed_vertices), np.uint32))


def reply_fixture(name, request_bytes, soa):
    from toyslam_amd import remote
    v = reply_vertices(soa["v_pos"], soa["v_type"])
    rep = remote.encode_response(request_bytes[4:], v)    # THIS repo's codec
    assert int(np.frombuffer(rep[:4], "<u4")[0]) == len(rep) - 4
    out = parsed_reply(rep[4:])
    out["reply_vertices_in"] = v
    out["reply_sha1"] = np.frombuffer(hashlib.sha1(rep).digest(), np.uint8)
    np.savez_compressed(os.path.join(OUT, name + "_reply_ref.npz"), **out)
    print(name, "reply of %d bytes parsed by the reference's bytes_to_graph: %d vertices, %d edges, LM meas shape %s, fixed %s"
          % (len(rep), len(out["v_id"]), len(out["e_type"]), tuple(out["e_meas_shape"][out["e_type"] == 1][0]) if (out["e_type"] == 1).any() else None, out["fixed"]))


def cpp_rules_with_reference_arithmetic(g, iterations):
    """Loop control of OptimizerCpu.h:80-180 around the reference's Python linearisation, numpy's dense solve and the
    reference's vertex update."""
    opt = GraphOptimizer(g)
    chi2, prev_err, penalty, stop, nrm = [], -1.0, 0, "cap", 0.0
    for _ in range(iterations):                                        # OptimizerCpu.h:80
        err = quiet(opt.calculate_H_b)                                 # :82-138 (py: +b, b zeroed at fixed vertices)
        chi2.append(float(err))
        if prev_err > 0 and err > prev_err:                            # :140-153
            penalty += 1
            if penalty > 2:
                stop = "worse"; break
        else:
            penalty = 0
        dx = np.linalg.solve(opt.H, -opt.b)                            # :155-157 (C++ b already carries the minus)
        for v_id in g.vertices:                                        # :159-165, step 0.2
            v = g.vertices[v_id]
            i = opt.vertex_ids_map[v_id]
            v.update(0.2 * dx[i:i + v.get_dims()])
        nrm = float(np.linalg.norm(dx))
        if abs(err - prev_err) < 1e-3:                                 # :167-171
            stop = "plateau"; break
        if nrm < 1e-3:                                                 # :173-177
            stop = "converged"; break
        prev_err = err                                                 # :179
    return np.array(chi2), stop, nrm


def main():
    z = np.load(os.path.join(OUT, "c1_graph.npz"))
    s = {k: z[k] for k in z.files}
    with open(os.path.join(OUT, "c1_request.bin"), "rb") as f:
        req = f.read()
    assert quiet(graph_to_bytes, mg.graph_from_soa(s, as_wire=False)) == req     # same graph as make_golden.py's
    reply_fixture("c1", req, s)
    for name in ("tiny_a", "tiny_b", "tiny_c"):
        t = np.load(os.path.join(OUT, name + ".npz"))
        reply_fixture(name, t["request"].tobytes(), {k: t[k] for k in ("v_pos", "v_type")})

    gw = mg.graph_from_soa(s, as_wire=True)                            # what a server receives
    chi2, stop, nrm = cpp_rules_with_reference_arithmetic(gw, 50)
    fin = mg.graph_soa(gw)
    print("cpp rules on the reference's arithmetic: %d iterations, stop %s, chi2 %.6f -> %.6f, |dx| %.3e" % (len(chi2), stop, chi2[0], chi2[-1], nrm))
    np.savez_compressed(os.path.join(OUT, "c1_cpprules_ref.npz"), chi2=chi2, stop=np.array(stop), delta_norm=np.float64(nrm),
                        v_id=fin["v_id"], v_type=fin["v_type"], v_pos=fin["v_pos"])


if __name__ == "__main__":
    main()

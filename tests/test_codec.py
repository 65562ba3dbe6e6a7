"""Wire codec (csrc/host/codec.cpp) against bytes produced by the reference's own graph_to_bytes
(tests/golden/c1_request.bin, tiny_*.npz["request"]) and against an independent reply reader."""
import os
import struct

import numpy as np
import pytest

from tests import util
from toyslam_amd import remote
from toyslam_amd.graph import (EdgeLandmark2d, EdgeOdometry2d, GraphArrays, OptGraph, Vertex2d,
                               VertexPose2d)


def golden_request():
    with open(util.GOLDEN + "/c1_request.bin", "rb") as f:
        return f.read()


def test_request_encoding_is_byte_identical_to_reference():
    req = golden_request()
    assert len(req) == 106672 and struct.unpack("<I", req[:4])[0] == 106668
    ours = remote.graph_to_bytes(util.c1_arrays(as_wire=False))
    assert ours == req


@pytest.mark.parametrize("name", ["tiny_a", "tiny_b", "tiny_c"])
def test_tiny_request_bytes(name):
    z = util.load(name + ".npz")
    ours = remote.graph_to_bytes(util.tiny_arrays(name))
    ref = z["request"].tobytes()
    # python iterates a set of fixed ids; for one fixed vertex the bytes are identical
    assert ours == ref


def test_decode_matches_fixture_graph():
    req = golden_request()
    g = remote.decode_request(req[4:])
    ref = util.c1_arrays(as_wire=True)
    np.testing.assert_array_equal(g.v_id, ref.v_id)
    np.testing.assert_array_equal(g.v_type, ref.v_type)
    np.testing.assert_array_equal(g.v_pos, ref.v_pos)          # exact: both are f32 values held in f64
    np.testing.assert_array_equal(g.e_type, ref.e_type)
    np.testing.assert_array_equal(g.e_ids, ref.e_ids)
    np.testing.assert_array_equal(g.e_meas, ref.e_meas)
    np.testing.assert_array_equal(g.e_inf, ref.e_inf)
    np.testing.assert_array_equal(g.fixed, ref.fixed)


def test_response_form_is_the_asymmetric_one():
    req = golden_request()
    g = remote.decode_request(req[4:])
    v = g.v_pos + 0.125
    rep = remote.encode_response(req[4:], v)
    size = struct.unpack("<I", rep[:4])[0]
    assert size == len(rep) - 4 == len(req) - 4              # reply length == request length
    out = remote.bytes_to_arrays(rep[4:])
    np.testing.assert_array_equal(out.v_id, g.v_id)
    np.testing.assert_allclose(out.v_pos[:, :2], v[:, :2], atol=1e-5)
    np.testing.assert_array_equal(out.e_ids, g.e_ids)
    np.testing.assert_array_equal(out.e_meas, g.e_meas)      # measurements are echoed bit-exactly
    np.testing.assert_array_equal(out.e_inf, g.e_inf)
    # LM edges are sent back as a 2x1 matrix, ODOM as 3x3 (SerializeGraphFuncCpu.h:56-58)
    first_edge = 4 + 150 * 20 + 342 * 16 + 4
    t, a, b, rows, cols = struct.unpack_from("<5I", rep, 4 + first_edge)
    assert (t, rows, cols) == (0, 3, 3)
    lm_edge = first_edge + 149 * 76
    t, a, b, rows, cols = struct.unpack_from("<5I", rep, 4 + lm_edge)
    assert (t, rows, cols) == (1, 2, 1)


@pytest.mark.parametrize("name", ["c1", "tiny_a", "tiny_b", "tiny_c"])
def test_reply_as_read_by_the_reference(name):
    """The reply form pinned by the REFERENCE's reader: tests/golden/make_golden_r3.py handed the reply this codec encodes to
    python/remote/bytes_to_graph.py:49-108 and stored what it returned.  Here: the codec still writes those very bytes (SHA-1),
    and this repo's reader (remote.bytes_to_arrays, remote.bytes_to_vertices) reads them as the reference did."""
    import hashlib
    z = util.load(name + "_reply_ref.npz")
    req = golden_request() if name == "c1" else util.load(name + ".npz")["request"].tobytes()
    rep = remote.encode_response(req[4:], z["reply_vertices_in"])
    assert hashlib.sha1(rep).digest() == z["reply_sha1"].tobytes()
    out = remote.bytes_to_arrays(rep[4:])
    np.testing.assert_array_equal(out.v_id, z["v_id"])          # the reference's dict keeps the reply's vertex order
    np.testing.assert_array_equal(out.v_type, z["v_type"])
    pose = z["v_type"] == 0
    vm = z["v_mat"].astype(np.float64)
    # poses: the reader rebuilds [[c, -s, x], [s, c, y], [0, 0, 1]] in float32 from (x, y, theta)
    np.testing.assert_array_equal(out.v_pos[pose, 0], vm[pose, 2]); np.testing.assert_array_equal(out.v_pos[pose, 1], vm[pose, 5])
    th = out.v_pos[pose, 2]
    np.testing.assert_allclose(np.cos(th), vm[pose, 0], atol=1e-7); np.testing.assert_allclose(np.sin(th), vm[pose, 3], atol=1e-7)
    np.testing.assert_allclose(-np.sin(th), vm[pose, 1], atol=1e-7); np.testing.assert_allclose(np.cos(th), vm[pose, 4], atol=1e-7)
    assert np.all(vm[pose, 6:] == [0, 0, 1])
    np.testing.assert_array_equal(out.v_pos[~pose, :2], vm[~pose, :2])
    # what went in comes out (f32-exact inputs)
    np.testing.assert_array_equal(out.v_pos, z["reply_vertices_in"])
    np.testing.assert_array_equal(out.e_type, z["e_type"]); np.testing.assert_array_equal(out.e_ids, z["e_ids"])
    odom = z["e_type"] == 0
    assert np.all(z["e_meas_shape"][odom] == [3, 3]) and np.all(z["e_meas_shape"][~odom] == [2, 1])      # SerializeGraphFuncCpu.h:56-58
    np.testing.assert_array_equal(out.e_meas, z["e_meas"].astype(np.float64))
    # information: the reader builds a k x k diagonal matrix from (0, k) + k floats
    assert np.all(z["e_inf_shape"][odom] == [3, 3]) and np.all(z["e_inf_shape"][~odom] == [2, 2])
    inf = z["e_inf"].astype(np.float64)
    np.testing.assert_array_equal(out.e_inf[odom], inf[odom][:, [0, 4, 8]])
    np.testing.assert_array_equal(out.e_inf[~odom][:, :2], inf[~odom][:, [0, 3]])
    assert np.all(inf[odom][:, [1, 2, 3, 5, 6, 7]] == 0) and np.all(inf[~odom][:, [1, 2]] == 0)
    np.testing.assert_array_equal(np.sort(np.asarray(out.fixed)), z["fixed"])
    like = remote.decode_request(req[4:])
    np.testing.assert_array_equal(remote.bytes_to_vertices(rep[4:], like), z["reply_vertices_in"])


def test_request_form_cannot_be_read_as_reply_and_vice_versa():
    req = golden_request()
    with pytest.raises(Exception):
        remote.bytes_to_arrays(req[4:])                      # same finding as SURVEY 8b [probe]


def test_odom_xytheta_form_and_short_information():
    # rows == 0 with an ODOM edge means (x, y, theta) (DeserializeGraph.h:76-85); k < dim keeps identity
    p = b"".join([
        struct.pack("<I", 2),
        struct.pack("<IIfff", 5, 0, 0.0, 0.0, 0.0), struct.pack("<IIfff", 9, 0, 1.0, 0.0, 0.1),
        struct.pack("<I", 1),
        struct.pack("<IIIII", 0, 5, 9, 0, 3), struct.pack("<fff", 1.0, 0.5, 0.25),
        struct.pack("<II", 0, 2), struct.pack("<ff", 4.0, 5.0),
        struct.pack("<I", 1), struct.pack("<I", 5)])
    g = remote.decode_request(p)
    m = g.e_meas[0].reshape(3, 3)
    np.testing.assert_allclose(m, [[np.cos(0.25), -np.sin(0.25), 1.0], [np.sin(0.25), np.cos(0.25), 0.5], [0, 0, 1]], atol=1e-7)
    np.testing.assert_array_equal(g.e_inf[0], [4.0, 5.0, 1.0])
    rep = remote.bytes_to_arrays(remote.encode_response(p, g.v_pos)[4:])
    assert rep.e_meas.shape == (1, 9)


@pytest.mark.parametrize("mutate,msg", [
    (lambda b: b[:100], "does not fit"),
    (lambda b: b[:80000], "fit|truncated"),
    (lambda b: struct.pack("<I", 0xFFFFFFF0) + b[4:], "vertex count"),
    (lambda b: b[:8] + struct.pack("<I", 7) + b[12:], "unknown vertex type"),
    (lambda b: b[:-4], "truncated"),
])
def test_malformed_payloads_are_rejected(mutate, msg):
    req = golden_request()[4:]
    with pytest.raises(RuntimeError, match=msg):
        remote.decode_request(mutate(req))


def test_non_diagonal_information_is_rejected():
    p = b"".join([struct.pack("<I", 1), struct.pack("<IIfff", 0, 0, 0, 0, 0), struct.pack("<I", 1),
                  struct.pack("<IIIII", 1, 0, 0, 0, 2), struct.pack("<ff", 1, 1), struct.pack("<II", 2, 2),
                  struct.pack("<ffff", 1, 0, 0, 1), struct.pack("<I", 0)])
    with pytest.raises(RuntimeError, match="diagonal"):
        remote.decode_request(p)                              # DeserializeGraph.h:144-147 throws here


def test_optgraph_mirror_round_trip():
    g = OptGraph()
    T = np.array([[np.cos(.3), -np.sin(.3), 1.], [np.sin(.3), np.cos(.3), 2.], [0, 0, 1.]])
    g.add_vertex(0, VertexPose2d(np.eye(3)), True)
    g.add_vertex(1, VertexPose2d(T))
    g.add_vertex(2, Vertex2d(np.array([3., 4.])))
    g.add_edge(EdgeOdometry2d(0, 1, T, np.diag([4., 4., 65.])))
    g.add_edge(EdgeLandmark2d(1, 2, np.array([2., .5]), np.diag([44., 44.])))
    with pytest.raises(RuntimeError):
        g.fix_vertex(17)
    a = GraphArrays.from_optgraph(g)
    assert a.n_poses == 2 and a.n_landmarks == 1 and list(a.fixed) == [0]
    np.testing.assert_allclose(a.v_pos[1], [1., 2., .3])
    b = remote.decode_request(remote.graph_to_bytes(g)[4:])
    np.testing.assert_allclose(b.v_pos, a.v_pos, atol=1e-6)
    a.write_back(g, a.v_pos + [[0, 0, 0], [1, 0, .1], [0, 1, 0]])
    np.testing.assert_allclose(g.get_vertex(1).position[:2, 2], [2., 2.])
    np.testing.assert_allclose(g.get_vertex(2).position, [3., 5.])


def test_mutated_requests_never_crash_the_decoder():
    """tools/fuzz_codec.py (truncations, bit flips, corrupted count words, trailing garbage): every mutated request is either
    rejected with an error or decoded and re-encoded; tools/sanitize_host.sh runs the same under ASan + UBSan."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_codec.py"), "600"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "no crash" in out.stdout

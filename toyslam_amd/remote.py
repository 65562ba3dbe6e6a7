"""Wire codec + client for ToySlam's remote optimizer protocol (counterpart of python/remote/*).

graph_to_bytes(graph)  -> the exact bytes python/remote/graph_to_bytes.py:32-67 produces
bytes_to_arrays(reply) -> parses what the server sends back (python/remote/bytes_to_graph.py:49-108 rules)
GraphClient            -> blocking-socket twin of python/remote/graph_client.py:6-59
The encoding/decoding itself is the C++ codec (csrc/host/codec.cpp) through the C ABI.
"""
import ctypes as C
import socket
import struct

import numpy as np

from . import _lib
from .graph import GraphArrays, OptGraph, tsgo_graph


def _arrays(graph):
    return graph if isinstance(graph, GraphArrays) else GraphArrays.from_optgraph(graph)


def graph_to_bytes(graph):
    lib = _lib.host_lib()
    g = _arrays(graph).c_struct()
    n = lib.tsgo_wire_encode_request(C.byref(g), None, 0)
    if n < 0:      # e.g. an edge type the wire format cannot carry (virtual landmark measurements: behind the C ABI only)
        raise RuntimeError("graph_to_bytes: " + lib.tsgo_last_error().decode())
    buf = (C.c_uint8 * n)()
    lib.tsgo_wire_encode_request(C.byref(g), buf, n)
    return bytes(buf)


def decode_request(payload):
    """payload: the request WITHOUT its 4-byte length prefix -> GraphArrays (as a server sees it)."""
    lib = _lib.host_lib()
    h = C.c_void_p()
    _lib.check(lib, lib.tsgo_wire_decode(payload, len(payload), C.byref(h)), "tsgo_wire_decode")
    try:
        view = tsgo_graph()
        lib.tsgo_wire_view(h, C.byref(view))
        return GraphArrays.from_c_struct(view)
    finally:
        lib.tsgo_wire_free(h)


def encode_response(payload, v_pos):
    """Server-side reply for a request payload with new vertex positions (prefix included)."""
    lib = _lib.host_lib()
    h = C.c_void_p()
    _lib.check(lib, lib.tsgo_wire_decode(payload, len(payload), C.byref(h)), "tsgo_wire_decode")
    try:
        v = np.ascontiguousarray(v_pos, np.float64)
        n = lib.tsgo_wire_encode_response(h, v.ctypes.data, None, 0)
        buf = (C.c_uint8 * n)()
        lib.tsgo_wire_encode_response(h, v.ctypes.data, buf, n)
        return bytes(buf)
    finally:
        lib.tsgo_wire_free(h)


def bytes_to_arrays(b):
    """Independent pure-Python reader of a REPLY payload (no prefix): rows x cols measurement matrices,
    (0, k) + diagonal information — the rules of python/remote/bytes_to_graph.py:16-108."""
    off = 0

    def u32():
        nonlocal off
        v = struct.unpack_from("<I", b, off)[0]; off += 4
        return v

    def f32(n):
        nonlocal off
        v = np.frombuffer(b, dtype="<f4", count=n, offset=off).astype(np.float64); off += 4 * n
        return v
    nV = u32()
    vid, vtype, vpos = [], [], []
    for _ in range(nV):
        i, t = u32(), u32()
        p = f32(3) if t == 0 else np.append(f32(2), 0.0)
        vid.append(i); vtype.append(t); vpos.append(p)
    nE = u32()
    etype, eids, emeas, einf = [], [], [], []
    for _ in range(nE):
        t, a, c = u32(), u32(), u32()
        rows, cols = u32(), u32()
        m = np.zeros(9); m[:rows * cols] = f32(rows * cols)
        r0, k = u32(), u32()
        assert r0 == 0
        w = np.zeros(3); w[:k] = f32(k)
        etype.append(t); eids.append([a, c]); emeas.append(m); einf.append(w)
    nF = u32()
    fixed = [u32() for _ in range(nF)]
    assert off == len(b), "trailing bytes in reply"
    return GraphArrays(vid, vtype, np.array(vpos).reshape(-1, 3), etype, np.array(eids, np.uint32).reshape(-1, 2),
                       np.array(emeas).reshape(-1, 9), np.array(einf).reshape(-1, 3), fixed)


def bytes_to_vertices(b, like):
    """Vertex positions (n, 3) of a REPLY payload (no prefix) for a request that was `like` (GraphArrays), in `like`'s vertex
    order — what python/slam_main.py:196-211 consumes.  Vectorised: a 100k-pose reply holds 300k vertex records.  The reply
    may list the vertices in any order (the reference's is unordered_map order): records are matched by id."""
    buf = np.frombuffer(b, dtype=np.uint8)
    n = int(np.frombuffer(b, dtype="<u4", count=1, offset=0)[0])
    if n != len(like.v_id):
        raise ValueError("reply holds %d vertices, the request had %d" % (n, len(like.v_id)))
    # record lengths depend on the type field of each record: walk them with the request's types first (same order is the
    # common case), verify the ids, and fall back to a sequential walk if the server reordered them
    size = np.where(like.v_type == 0, 20, 16).astype(np.int64)
    off = 4 + np.concatenate(([0], np.cumsum(size)[:-1]))
    words = buf[: 4 + int(size.sum())]

    def u32_at(o):
        return (words[o].astype(np.uint32) | (words[o + 1].astype(np.uint32) << 8) | (words[o + 2].astype(np.uint32) << 16) | (words[o + 3].astype(np.uint32) << 24))

    def f32_at(o):
        return u32_at(o).view(np.float32).astype(np.float64)
    ids, types = u32_at(off), u32_at(off + 4)
    if np.array_equal(ids, like.v_id) and np.array_equal(types, like.v_type):
        out = np.zeros((n, 3))
        out[:, 0] = f32_at(off + 8); out[:, 1] = f32_at(off + 12)
        pose = like.v_type == 0
        out[pose, 2] = f32_at(off[pose] + 16)
        return out
    got = bytes_to_arrays(b)
    order = {int(i): k for k, i in enumerate(got.v_id)}
    return got.v_pos[[order[int(i)] for i in like.v_id]]


class GraphClient:
    """connect() / optimize(graph) / close() as python/remote/graph_client.py:13-59, blocking sockets."""

    def __init__(self, host, port):
        self.host, self.port, self.sock = host, port, None

    def connect(self):
        self.sock = socket.create_connection((self.host, self.port))

    def _read(self, n):
        # straight into one buffer of the final size (a 100k-pose reply is 57 MB: appending chunk by chunk copies it several times over)
        out = bytearray(n)
        view = memoryview(out)
        got = 0
        while got < n:
            k = self.sock.recv_into(view[got:], n - got)
            if k == 0:
                raise ValueError("Connection closed by the server")
            got += k
        return out

    def optimize(self, graph):
        arr = _arrays(graph)
        self.sock.sendall(graph_to_bytes(arr))
        size = struct.unpack("<I", self._read(4))[0]
        reply = bytes_to_arrays(self._read(size))
        if isinstance(graph, OptGraph):
            arr.write_back(graph, reply.v_pos)
            return graph
        return reply

    def close(self):
        if self.sock:
            self.sock.close()
            self.sock = None

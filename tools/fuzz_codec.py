"""Mutation fuzz of the wire decoder (host only): truncations, bit flips and count-field corruption of a valid request must
end in a clean error or a successful decode + re-encode — never in a crash or an out-of-bounds access (run under
tools/sanitize_host.sh with ASan/UBSan; also fine without)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from toyslam_amd import _lib, remote, synth  # noqa: E402
from toyslam_amd.graph import tsgo_graph  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
lib = _lib.host_lib()
good = remote.graph_to_bytes(synth.make(40, 5, loop_closures=2, seed=3))[4:]
rng = np.random.default_rng(0)
h = lib.tsgo_wire_new()
ok = bad = 0
for k in range(n):
    b = bytearray(good)
    kind = k % 4
    if kind == 0:
        b = b[: int(rng.integers(0, len(b)))]
    elif kind == 1:
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 2:                                   # corrupt a 32-bit word (counts, types, rows/cols live in them)
        o = 4 * int(rng.integers(0, len(b) // 4)); b[o:o + 4] = int(rng.integers(0, 2 ** 32)).to_bytes(4, "little")
    else:
        b = b + bytes(rng.integers(0, 256, size=int(rng.integers(1, 64)), dtype=np.uint8))
    rc = lib.tsgo_wire_decode_into(h, bytes(b), len(b))
    if rc == 0:
        view = tsgo_graph(); lib.tsgo_wire_view(h, C.byref(view))
        v = np.zeros((max(view.n_vertices, 1), 3))
        size = lib.tsgo_wire_encode_response(h, v.ctypes.data, None, 0)
        buf = (C.c_uint8 * size)()
        assert lib.tsgo_wire_encode_response(h, v.ctypes.data, buf, size) == size
        ok += 1
    else:
        bad += 1
lib.tsgo_wire_free(h)
print("fuzz: %d mutated requests, %d decoded (and re-encoded), %d rejected with an error, no crash" % (n, ok, bad))

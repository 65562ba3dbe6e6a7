"""Print the multigrid hierarchy's shape (rows, blocks per row) for a synthetic workload: host-only, no GPU."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from toyslam_amd import synth, _lib

name = sys.argv[1] if len(sys.argv) > 1 else "c3_100k"
g = synth.make_config(name)
lib = _lib.host_lib()
info = _lib.tsgo_amg_info()
cg = g.c_struct()
_lib.check(lib, lib.tsgo_amg_probe(C.byref(cg), C.byref(info)), "tsgo_amg_probe")
for l in range(info.n_levels):
    r, b, p = info.rows[l], info.blocks[l], info.p_blocks[l]
    print("level %d: %8d rows  %9d blocks (%.1f per row, %.2f MB f32)  P %9d blocks" % (l, r, b, b / max(r, 1), b * 36 / 1e6, p))
print("symbolic %.0f ms" % info.ms_symbolic)

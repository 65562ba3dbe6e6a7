"""Research soak (GPU): the TCP server under a stream of random requests from three persistent connections at once (two engines:
connections share them) — new structures of very different sizes, the same structure again with the estimates that came back,
grown graphs — every reply compared with what an in-process handle gives for the same (wire-rounded) request.  Exercises engine
affinity, structure reuse on a shared engine, the device slabs and the connection's reusable decode arrays."""
import os, socket, struct, subprocess, sys, threading, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from tests import util
from toyslam_amd import build, remote, synth
from toyslam_amd.graph import GraphArrays
from toyslam_amd.optimizer import HipOptimizer

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5
DEVICES = sys.argv[3] if len(sys.argv) > 3 else None      # e.g. "0,0": the engine pool over a device list (one handle per entry then)
ITERS = 6


def roundtrip(sock, req):
    sock.sendall(req)
    hdr = b""
    while len(hdr) < 4:
        c = sock.recv(4 - len(hdr))
        if not c:
            raise RuntimeError("server closed the connection")
        hdr += c
    size = struct.unpack("<I", hdr)[0]
    body = bytearray(size); view = memoryview(body); got = 0
    while got < size:
        k = sock.recv_into(view[got:], size - got)
        if not k:
            raise RuntimeError("server closed the connection")
        got += k
    return bytes(body)


s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
log = open("/tmp/soak_server.log", "w")
proc = subprocess.Popen([build.SERVER, "127.0.0.1", str(port), str(ITERS), "gpu", "cuda"] + (["64", "1e-10", DEVICES, "1"] if DEVICES else []), stdout=log, stderr=subprocess.STDOUT)
for _ in range(300):
    try:
        socket.create_connection(("127.0.0.1", port), timeout=0.5).close(); break
    except OSError:
        time.sleep(0.2)
t_end = time.time() + budget
check_lock = threading.Lock()          # one in-process reference handle at a time
failures, counts = [], [0, 0, 0]


def client(cid):
    rng = np.random.default_rng(seed0 * 100 + cid)
    sock = socket.create_connection(("127.0.0.1", port)); sock.settimeout(300)
    g = None; n_req = 0
    try:
        while time.time() < t_end and not failures:
            kind = int(rng.integers(0, 3)) if g is not None else 0
            if kind == 0:                                  # a new structure
                n = int(rng.choice([40, 150, 900, 4000, 15000])) + int(rng.integers(0, 40))
                g = synth.make(n, int(rng.integers(3, 12)), loop_closures=int(rng.integers(0, 1 + n // 100)), seed=int(rng.integers(0, 10 ** 6))).rounded_to_wire()
                what = "new   "
            elif kind == 1:                                # the same structure, the estimates of the last reply
                g = GraphArrays(g.v_id, g.v_type, last_v, g.e_type, g.e_ids, g.e_meas, g.e_inf, g.fixed)
                what = "repeat"
            else:                                          # grown by a few poses (same seed: the same walk, longer)
                n = int((g.v_type == 0).sum()) + int(rng.integers(1, 30))
                g = synth.make(n, 6, loop_closures=2, seed=77 + cid).rounded_to_wire()
                what = "grown "
            body = roundtrip(sock, remote.graph_to_bytes(g))
            got = remote.bytes_to_vertices(body, g)
            with check_lock:
                o = HipOptimizer()
                try:
                    o.set_graph(g); r = o.optimize(ITERS); want = o.vertices()
                finally:
                    o.close()
            scale = max(1.0, float(np.abs(want).max()))
            d = util.max_vertex_diff(got, want, g.v_type)
            ok = d < 2e-6 * scale                              # the reply is f32 on the wire
            n_req += 1; counts[cid] = n_req
            print("client %d request %3d %s %6d poses %7d edges: max |server - in-process| = %.2e (f32 reply, extent %.0f)  %s"
                  % (cid, n_req, what, int((g.v_type == 0).sum()), len(g.e_type), d, scale, "ok" if ok else "MISMATCH"), flush=True)
            if not ok:
                failures.append((cid, n_req)); break
            last_v = got
    except Exception as e:                                     # noqa: BLE001
        failures.append((cid, repr(e)))
    finally:
        sock.close()


th = [threading.Thread(target=client, args=(k,)) for k in range(3)]
for t in th: t.start()
for t in th: t.join()
alive = proc.poll() is None
proc.terminate(); proc.wait(timeout=20); log.close()
txt = open("/tmp/soak_server.log").read()
print("server log: %d requests built a structure, %d reused one" % (txt.count("structure=built"), txt.count("structure=reused")))
print("server soak: %s requests per client, server alive at the end: %s, failures: %s" % (counts, alive, failures))
sys.exit(0 if (alive and not failures) else 1)

"""End-to-end latency of requests through the TCP server: starts toyslam_amd/graph_optimizer and sends,
over ONE connection: the synthetic request of the named workload (first request: layout, multigrid patterns, uploads), the same
structure again with the estimates the server returned (what a SLAM front-end resends: structure reused, values refilled), once
more, and then the graph grown by 5 % (new structure: full rebuild).  Prints the client-side wall time of each and the server's own
phase lines (reference captions: DeserializeGraph / Optimize / SerializeGraph / Total).  GPU box only."""
import os, socket, struct, subprocess, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from toyslam_amd import build, remote, synth
from toyslam_amd.graph import GraphArrays

name = sys.argv[1] if len(sys.argv) > 1 else "c3_100k"
iters = sys.argv[2] if len(sys.argv) > 2 else "10"
p, k, lc = synth.CONFIGS[name]
big = synth.make(int(p * 1.05), k, loop_closures=lc, seed=0)          # the same walk, 5 % longer
g = synth.make_config(name)


def send(sock, req):
    t = time.perf_counter()
    sock.sendall(req)
    hdr = b""
    while len(hdr) < 4:
        hdr += sock.recv(4 - len(hdr))
    size = struct.unpack("<I", hdr)[0]
    body = bytearray(size); view = memoryview(body); got = 0
    while got < size:
        got += sock.recv_into(view[got:], size - got)
    return 1e3 * (time.perf_counter() - t), bytes(body)


s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
log = open("/tmp/request_latency_server.log", "w")
env = dict(os.environ)
if len(sys.argv) > 3:
    env["TSGO_WARMUP_POSES"] = sys.argv[3]          # the server's warm-up at the size of the requests to come
proc = subprocess.Popen([build.SERVER, "127.0.0.1", str(port), iters, "gpu", "cuda"], stdout=log, stderr=subprocess.STDOUT, env=env)
try:
    for _ in range(300):
        try:
            sock = socket.create_connection(("127.0.0.1", port), timeout=0.5); break
        except OSError:
            time.sleep(0.2)
    sock.settimeout(600)
    sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
    req = remote.graph_to_bytes(g)
    print("request: %.1f MB (%d poses, %d edges), %s GN iterations per request" % (len(req) / 1e6, g.n_poses, g.n_edges, iters), flush=True)
    ms, body = send(sock, req)
    print("request 0 (first: structure built)        client wall %7.1f ms" % ms, flush=True)
    for rep in (1, 2):
        v = remote.bytes_to_vertices(body, g)                    # the estimates that came back, in request order
        g = GraphArrays(g.v_id, g.v_type, v, g.e_type, g.e_ids, g.e_meas, g.e_inf, g.fixed)
        ms, body = send(sock, remote.graph_to_bytes(g))
        print("request %d (same structure, new estimates) client wall %7.1f ms" % (rep, ms), flush=True)
    ms, body = send(sock, remote.graph_to_bytes(big))
    print("request 3 (graph grown by 5 %%: rebuilt)    client wall %7.1f ms" % ms, flush=True)
    sock.close()
finally:
    proc.terminate(); proc.wait(timeout=20); log.close()
print(open("/tmp/request_latency_server.log").read())

"""End-to-end latency of one request through the TCP server: starts toyslam_amd/graph_optimizer,
sends a synthetic request of the named workload twice over one connection (the second shows what a warm process
pays), prints the server's own phase lines (reference captions: DeserializeGraph / Optimize / SerializeGraph / Total)
and the client-side wall time.  GPU box only."""
import os, socket, struct, subprocess, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from toyslam_amd import build, remote, synth

name = sys.argv[1] if len(sys.argv) > 1 else "c3_100k"
iters = sys.argv[2] if len(sys.argv) > 2 else "10"
g = synth.make_config(name)
t = time.perf_counter(); req = remote.graph_to_bytes(g); t_enc = time.perf_counter() - t
print("request: %.1f MB, client-side encode %.0f ms" % (len(req) / 1e6, 1e3 * t_enc), flush=True)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
log = open("/tmp/request_latency_server.log", "w")
proc = subprocess.Popen([build.SERVER, "127.0.0.1", str(port), iters, "gpu", "cuda"], stdout=log, stderr=subprocess.STDOUT)
try:
    for _ in range(300):
        try:
            sock = socket.create_connection(("127.0.0.1", port), timeout=0.5); break
        except OSError:
            time.sleep(0.2)
    sock.settimeout(300)
    for rep in range(2):
        t = time.perf_counter()
        sock.sendall(req)
        hdr = b""
        while len(hdr) < 4:
            hdr += sock.recv(4 - len(hdr))
        size = struct.unpack("<I", hdr)[0]
        got = 0
        while got < size:
            got += len(sock.recv(min(1 << 22, size - got)))
        print("request %d: client wall %.0f ms (reply %.1f MB)" % (rep, 1e3 * (time.perf_counter() - t), size / 1e6), flush=True)
    sock.close()
finally:
    proc.terminate(); proc.wait(timeout=20); log.close()
print(open("/tmp/request_latency_server.log").read())

"""Research (GPU box): requests per second through the TCP server with 1, 2, 3 concurrent clients (ENGINES = 3)."""
import os, socket, struct, subprocess, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from toyslam_amd import build, remote, synth

name = sys.argv[1] if len(sys.argv) > 1 else "c2_10k"
iters = sys.argv[2] if len(sys.argv) > 2 else "10"
reqs = [remote.graph_to_bytes(synth.make_config(name, seed=k)) for k in range(3)]
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
proc = subprocess.Popen([build.SERVER, "127.0.0.1", str(port), iters, "gpu", "cuda", "64", "1e-10", "0", "3"], stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
for _ in range(300):
    try:
        socket.create_connection(("127.0.0.1", port), timeout=0.5).close(); break
    except OSError:
        time.sleep(0.2)


def client(k, n):
    sock = socket.create_connection(("127.0.0.1", port)); sock.settimeout(300)
    for _ in range(n):
        sock.sendall(reqs[k])
        hdr = b""
        while len(hdr) < 4:
            hdr += sock.recv(4 - len(hdr))
        size = struct.unpack("<I", hdr)[0]; got = 0
        while got < size:
            got += len(sock.recv(min(1 << 22, size - got)))
    sock.close()


try:
    client(0, 2)        # warm the process
    for n_clients in (1, 2, 3):
        n = 6
        th = [threading.Thread(target=client, args=(k, n)) for k in range(n_clients)]
        t = time.perf_counter()
        [x.start() for x in th]; [x.join() for x in th]
        dt = time.perf_counter() - t
        print("%s, %s GN iterations per request: %d client(s): %.2f requests/s (%.0f ms per request per client)" % (name, iters, n_clients, n_clients * n / dt, 1e3 * dt / n), flush=True)
finally:
    proc.terminate(); proc.wait(timeout=20)

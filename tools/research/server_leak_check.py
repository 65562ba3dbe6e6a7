"""Research (GPU box): resident memory of the server process over many requests (host leaks show as growth)."""
import os, socket, struct, subprocess, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from toyslam_amd import build, remote, synth

reqs = [remote.graph_to_bytes(synth.make(3000 + 500 * k, 8, loop_closures=4, seed=k)) for k in range(4)]
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
proc = subprocess.Popen([build.SERVER, "127.0.0.1", str(port), "5", "gpu", "cuda"], stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
for _ in range(300):
    try:
        sock = socket.create_connection(("127.0.0.1", port), timeout=0.5); break
    except OSError:
        time.sleep(0.2)
sock.settimeout(120)


def rss_mb():
    for line in open("/proc/%d/status" % proc.pid):
        if line.startswith("VmRSS"):
            return int(line.split()[1]) / 1024.0


try:
    marks = {}
    for n in range(1, 81):
        sock.sendall(reqs[n % 4])
        hdr = b""
        while len(hdr) < 4:
            hdr += sock.recv(4 - len(hdr))
        size = struct.unpack("<I", hdr)[0]; got = 0
        while got < size:
            got += len(sock.recv(min(1 << 22, size - got)))
        if n in (10, 20, 40, 80):
            marks[n] = rss_mb(); print("after %3d requests: server RSS %.1f MB" % (n, marks[n]), flush=True)
    print("growth from request 20 to 80: %.1f MB" % (marks[80] - marks[20]))
finally:
    sock.close(); proc.terminate(); proc.wait(timeout=20)

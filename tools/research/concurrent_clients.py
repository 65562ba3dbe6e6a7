import os, socket, subprocess, sys, threading, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from toyslam_amd import build, remote, synth
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
log = open("/tmp/srv.log", "w")
proc = subprocess.Popen([build.SERVER, "127.0.0.1", str(port), "20", "gpu", "cuda"], stdout=log, stderr=subprocess.STDOUT)
for _ in range(300):
    try:
        socket.create_connection(("127.0.0.1", port), timeout=0.5).close(); break
    except OSError:
        time.sleep(0.2)
graphs = [synth.make(1500 + 400 * k, 6 + k, loop_closures=5, seed=40 + k) for k in range(3)]
def client(k):
    try:
        c = remote.GraphClient("127.0.0.1", port); c.connect()
        for _ in range(2):
            out = c.optimize(graphs[k])
        c.close(); print("client", k, "ok", len(out.v_id))
    except Exception as e:
        print("client", k, "failed", repr(e))
th = [threading.Thread(target=client, args=(k,)) for k in range(3)]
[t.start() for t in th]; [t.join() for t in th]
proc.terminate(); proc.wait(); log.close()
print(open("/tmp/srv.log").read()[-3000:])

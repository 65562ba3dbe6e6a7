"""End-to-end drop-in check on the GPU box: the C++ `graph_optimizer` server speaking the reference's TCP
protocol (remote/app/ConnectionHandlerGraph.h:20-52) to a client that sends the golden request bytes
produced by the reference's own graph_to_bytes."""
import os
import socket
import struct
import subprocess
import time

import numpy as np
import pytest

from oracle import oracle
from tests import util
from toyslam_amd import build, remote

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _start(iterations, *extra):
    exe = build.SERVER
    if not os.path.exists(exe):
        pytest.fail("toyslam_amd/graph_optimizer is not built (run __graft_entry__.build())")
    port = _free_port()
    # HOST PORT ITERATIONS PIPELINE SOLVER — the reference's positional CLI (remote/app/main.cpp:12-16)
    proc = subprocess.Popen([exe, "127.0.0.1", str(port), str(iterations), "gpu", "cuda", *extra], stdout=subprocess.PIPE,
                            stderr=subprocess.STDOUT, text=True)
    deadline = time.time() + 60
    while time.time() < deadline:
        try:
            socket.create_connection(("127.0.0.1", port), timeout=0.5).close()
            break
        except OSError:
            if proc.poll() is not None:
                pytest.fail("server exited: " + proc.stdout.read())
            time.sleep(0.2)
    return port, proc


def _stop(proc):
    proc.terminate()
    try:
        proc.wait(timeout=10)
    except subprocess.TimeoutExpired:
        proc.kill()


@pytest.fixture(scope="module")
def server():
    port, proc = _start(50)
    yield port, proc
    _stop(proc)


def _roundtrip(sock, request):
    sock.sendall(request)
    hdr = b""
    while len(hdr) < 4:
        hdr += sock.recv(4 - len(hdr))
    size = struct.unpack("<I", hdr)[0]
    body = b""
    while len(body) < size:
        chunk = sock.recv(size - len(body))
        assert chunk, "server closed the connection"
        body += chunk
    return body


def test_golden_request_round_trip_matches_cpu_eigen(server):
    port, _ = server
    with open(os.path.join(util.GOLDEN, "c1_request.bin"), "rb") as f:
        req = f.read()
    g = util.c1_arrays()
    ref = oracle.optimize(util.to_oracle(g), 50, mode="cpp", solver="chol")
    with socket.create_connection(("127.0.0.1", port)) as s:
        body = _roundtrip(s, req)
        assert len(body) == len(req) - 4
        out = remote.bytes_to_arrays(body)
        np.testing.assert_array_equal(out.v_id, g.v_id)
        # the reply carries f32: compare at f32 resolution of coordinates up to ~70
        assert util.max_vertex_diff(out.v_pos, ref["v_pos"], g.v_type) < 1e-5
        np.testing.assert_array_equal(out.e_ids, g.e_ids)
        np.testing.assert_array_equal(out.fixed, g.fixed)
        # the connection is persistent: a second, different graph on the same socket
        t = util.tiny_arrays("tiny_a")
        body2 = _roundtrip(s, remote.graph_to_bytes(t))
        out2 = remote.bytes_to_arrays(body2)
        ref2 = oracle.optimize(util.to_oracle(t.rounded_to_wire()), 50, mode="cpp", solver="qr")
        assert util.max_vertex_diff(out2.v_pos, ref2["v_pos"], t.v_type) < 1e-5


def test_client_class_and_malformed_request(server):
    port, proc = server
    g = util.tiny_arrays("tiny_c")
    c = remote.GraphClient("127.0.0.1", port)
    c.connect()
    out = c.optimize(g)
    c.close()
    ref = oracle.optimize(util.to_oracle(g.rounded_to_wire()), 50, mode="cpp", solver="qr")
    assert util.max_vertex_diff(out.v_pos, ref["v_pos"], g.v_type) < 1e-5
    # garbage: the server drops that connection and keeps serving (the reference would terminate)
    with socket.create_connection(("127.0.0.1", port)) as s:
        s.sendall(struct.pack("<i", 64) + b"\xff" * 64)
        s.settimeout(5)
        assert s.recv(4) == b""
    assert proc.poll() is None
    c = remote.GraphClient("127.0.0.1", port)
    c.connect()
    assert len(c.optimize(g).v_id) == len(g.v_id)
    c.close()


def test_requests_from_several_connections_run_concurrently_and_stay_correct(server):
    """Three clients at once (the server keeps up to ENGINES = 2 requests in flight on the device, each on its own
    engine handle and stream): every reply must be the answer to ITS request."""
    import threading
    from toyslam_amd import synth
    port, proc = server
    graphs = [synth.make(1500 + 400 * k, 6 + k, loop_closures=5, seed=40 + k) for k in range(3)]
    refs = [oracle.sparse_optimize(util.to_oracle(g.rounded_to_wire()), 50, pcg_tol=1e-12, precond="amg") for g in graphs]
    outs = [None] * 3
    errs = []

    def client(k):
        try:
            c = remote.GraphClient("127.0.0.1", port)
            c.connect()
            for _ in range(2):                      # two requests per connection, interleaving with the others
                outs[k] = c.optimize(graphs[k])
            c.close()
        except Exception as e:                      # noqa: BLE001 - reported below
            errs.append((k, repr(e)))

    th = [threading.Thread(target=client, args=(k,)) for k in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errs, errs
    assert proc.poll() is None
    for k in range(3):
        assert outs[k] is not None and len(outs[k].v_id) == len(graphs[k].v_id)
        assert util.max_vertex_diff(outs[k].v_pos, refs[k]["v_pos"], graphs[k].v_type) < 2e-4    # f32 on the wire


def test_same_structure_then_a_grown_graph_on_one_connection(server):
    """SURVEY 8f rank 2.  What a SLAM front-end does over one connection: send a graph, send the SAME structure again with
    the estimates that came back (the server refills values and keeps layout, patterns and tables), then a graph grown by
    5 % (new structure: everything is rebuilt).  Every reply is checked against the dense cpu/eigen restatement of ITS
    request; the server's log says which path each request took."""
    from toyslam_amd import synth
    from toyslam_amd.graph import GraphArrays
    port, proc = server
    g0 = synth.make(120, 8, seed=21).rounded_to_wire()
    grown = synth.make(126, 8, seed=21).rounded_to_wire()
    with socket.create_connection(("127.0.0.1", port)) as s:
        ref0 = oracle.optimize(util.to_oracle(g0), 50, mode="cpp", solver="chol")
        v0 = remote.bytes_to_vertices(_roundtrip(s, remote.graph_to_bytes(g0)), g0)
        assert util.max_vertex_diff(v0, ref0["v_pos"], g0.v_type) < 1e-5
        # the same structure, estimates replaced by the (f32) reply, measurements re-weighted
        g1 = GraphArrays(g0.v_id, g0.v_type, v0, g0.e_type, g0.e_ids, g0.e_meas, (g0.e_inf * 0.5).astype(np.float32).astype(np.float64), g0.fixed)
        ref1 = oracle.optimize(util.to_oracle(g1), 50, mode="cpp", solver="chol")
        v1 = remote.bytes_to_vertices(_roundtrip(s, remote.graph_to_bytes(g1)), g1)
        assert util.max_vertex_diff(v1, ref1["v_pos"], g1.v_type) < 1e-5
        assert util.max_vertex_diff(v1, v0, g1.v_type) > 1e-4          # it did move: the second request was really optimised
        ref2 = oracle.optimize(util.to_oracle(grown), 50, mode="cpp", solver="chol")
        v2 = remote.bytes_to_vertices(_roundtrip(s, remote.graph_to_bytes(grown)), grown)
        assert util.max_vertex_diff(v2, ref2["v_pos"], grown.v_type) < 1e-5
    assert proc.poll() is None


def test_a_connections_next_request_continues_from_the_solver_history_of_its_last_one():
    """SURVEY 8f rank 2 ("warm-starting PCG across requests").  The server runs tsgo_config.warm_requests by default (trailing
    argument WARM_REQUESTS, 0 switches it off): a front-end that sends back the estimates it was returned gets the same answer in
    fewer PCG iterations — the server's own log line counts them — and a grown graph sent next is still answered correctly."""
    import re
    from toyslam_amd import synth
    from toyslam_amd.graph import GraphArrays
    big = synth.make(3300, 8, seed=23).rounded_to_wire()
    g0 = util.first_poses(big, 3000)
    counts, replies = {}, {}
    for warm in ("1", "0"):
        port, proc = _start(8, "64", "1e-10", "0", "1", "cpp", "constant", warm)
        try:
            with socket.create_connection(("127.0.0.1", port)) as s:
                v0 = remote.bytes_to_vertices(_roundtrip(s, remote.graph_to_bytes(g0)), g0)
                g1 = GraphArrays(g0.v_id, g0.v_type, v0, g0.e_type, g0.e_ids, g0.e_meas, g0.e_inf, g0.fixed)
                v1 = remote.bytes_to_vertices(_roundtrip(s, remote.graph_to_bytes(g1)), g1)
                at = {int(i): k for k, i in enumerate(g1.v_id)}
                vp = big.v_pos.copy()
                for k, i in enumerate(big.v_id):
                    if int(i) in at:
                        vp[k] = v1[at[int(i)]]
                g2 = GraphArrays(big.v_id, big.v_type, vp, big.e_type, big.e_ids, big.e_meas, big.e_inf, big.fixed)
                v2 = remote.bytes_to_vertices(_roundtrip(s, remote.graph_to_bytes(g2)), g2)
        finally:
            _stop(proc)
        out = proc.stdout.read()
        counts[warm] = [int(m) for m in re.findall(r"pcg_iters=(\d+)", out)][-3:]        # (the warm-up solve logs nothing)
        replies[warm] = (v0, v1, v2, g1, g2)
    assert len(counts["1"]) == 3 and len(counts["0"]) == 3, counts
    assert counts["1"][0] == counts["0"][0]                      # a connection's first request has nothing to continue from
    assert counts["1"][1] < counts["0"][1], counts               # the same structure with the returned estimates: warm
    for a, b, g in zip(replies["1"][:3], replies["0"][:3], (g0, replies["1"][3], replies["1"][4])):
        assert util.max_vertex_diff(a, b, g.v_type) < 2e-6       # same answers (f32 on the wire)
    ref = oracle.sparse_optimize(util.to_oracle(replies["1"][4]), 8, pcg_tol=1e-12, precond="amg")
    assert util.max_vertex_diff(replies["1"][2], ref["v_pos"], big.v_type) < 1e-4
    print("PCG iterations per request (first / same structure again / grown): warm_requests on %s, off %s" % (counts["1"], counts["0"]))


def test_solver_history_never_crosses_from_one_connection_to_another():
    """engine handles are pooled across connections, and SLAM clients all number their vertices 0..N — a handle given to
    another connection must not start that client's solves from the previous client's deltas.  ONE engine (ENGINES = 1), two clients
    sending the same structure: the server's log line says per request whether it started from carried history."""
    import re
    from toyslam_amd import synth
    from toyslam_amd.graph import GraphArrays
    g0 = synth.make(2500, 8, seed=29).rounded_to_wire()
    port, proc = _start(6, "64", "1e-10", "0", "1", "cpp", "constant", "1")
    try:
        with socket.create_connection(("127.0.0.1", port)) as a, socket.create_connection(("127.0.0.1", port)) as b:
            va0 = remote.bytes_to_vertices(_roundtrip(a, remote.graph_to_bytes(g0)), g0)
            g1 = GraphArrays(g0.v_id, g0.v_type, va0, g0.e_type, g0.e_ids, g0.e_meas, g0.e_inf, g0.fixed)
            va1 = remote.bytes_to_vertices(_roundtrip(a, remote.graph_to_bytes(g1)), g1)          # A again: its own history
            vb0 = remote.bytes_to_vertices(_roundtrip(b, remote.graph_to_bytes(g1)), g1)          # B's first request, same structure, same handle
            vb1 = remote.bytes_to_vertices(_roundtrip(b, remote.graph_to_bytes(g1)), g1)          # B again: nothing to continue (same input: no step taken in between)
            va2 = remote.bytes_to_vertices(_roundtrip(a, remote.graph_to_bytes(g1)), g1)          # A after B used the handle: forgotten
    finally:
        _stop(proc)
    out = proc.stdout.read()
    hist = [int(m) for m in re.findall(r"history=(\d+)", out)][-5:]
    assert len(hist) == 5, out[-2000:]
    assert hist[0] == 0 and hist[1] in (1, 2), hist       # A: first request cold, second one from its own history
    assert hist[2] == 0, hist                              # B's first request on the handle A just used: cold
    assert hist[4] == 0, hist                              # A, after the handle served B: cold again
    # B's answer is what a connection of its own server would have got: the same request, cold, is deterministic
    assert np.array_equal(vb0, va2), np.abs(vb0 - va2).max()
    assert util.max_vertex_diff(va1, vb0, g1.v_type) < 2e-6       # warm or cold: the same answer to the solver's tolerance (f32 on the wire)
    assert vb1.shape == vb0.shape


def test_trailing_arguments_select_the_python_rules_and_the_analytic_odometry_jacobians():
    """The server's optional trailing arguments (after the reference's five): PRECISION PCG_TOL DEVICE ENGINES RULES ODOM_JACOBIAN.
    `python:0.5 analytic` must give what the in-process handle gives with rules="python", lr=0.5, odom_jacobian="analytic" —
    here on a pose graph with loop closures, which the reference's own constants drive into "Error is getting worse"."""
    from toyslam_amd import synth
    from toyslam_amd.graph import GraphArrays
    from toyslam_amd.optimizer import HipOptimizer
    g = synth.make(400, 6, loop_closures=12, seed=5)
    keep = g.e_type == 0; pose = g.v_type == 0
    g = GraphArrays(g.v_id[pose], g.v_type[pose], g.v_pos[pose], g.e_type[keep], g.e_ids[keep], g.e_meas[keep], g.e_inf[keep], g.fixed).rounded_to_wire()
    o = HipOptimizer(rules="python", lr=0.5, odom_jacobian="analytic")
    try:
        o.set_graph(g); r = o.optimize(8); want = o.vertices()
    finally:
        o.close()
    assert r["chi2"][-1] < 0.2 * r["chi2"][0]                              # this combination converges on such a graph
    port, proc = _start(8, "64", "1e-10", "0", "1", "python:0.5", "analytic")
    try:
        with socket.create_connection(("127.0.0.1", port)) as s:
            got = remote.bytes_to_vertices(_roundtrip(s, remote.graph_to_bytes(g)), g)
    finally:
        _stop(proc)
    out = proc.stdout.read()
    assert "lr 0.5" in out and "analytic" in out
    assert util.max_vertex_diff(got, want, g.v_type) < 1e-4               # the reply is f32 on the wire


def test_the_engine_pool_spreads_connections_over_the_listed_gpus():
    """DEVICE takes a list (or "all"): ENGINES handles on EACH listed GPU, a request goes to its connection's own last handle when that is
    idle, else to the listed GPU with the fewest requests in flight — one graph per GPU, no collective (how the server uses a node).  A
    one-GPU box rehearses it with the same GPU listed twice ("0,0", one handle each): two clients at once land on different pools, each
    connection then stays on its pool (same structure refilled, history carried), answers equal to the twin's."""
    import threading
    from toyslam_amd import synth
    graphs = [synth.make(2500 + 500 * k, 6, loop_closures=4, seed=60 + k) for k in range(2)]
    refs = [oracle.sparse_optimize(util.to_oracle(g.rounded_to_wire()), 6, pcg_tol=1e-12, precond="amg") for g in graphs]
    port, proc = _start(6, "64", "1e-10", "0,0", "1")
    outs = [None, None]
    errs = []
    start = threading.Barrier(2)

    def client(k):
        try:
            c = remote.GraphClient("127.0.0.1", port)
            c.connect()
            start.wait(timeout=30)
            for _ in range(3):
                outs[k] = c.optimize(graphs[k])
            c.close()
        except Exception as e:                      # noqa: BLE001 - reported below
            errs.append((k, repr(e)))

    try:
        th = [threading.Thread(target=client, args=(k,)) for k in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=120)
        assert not errs, errs
        assert proc.poll() is None
    finally:
        _stop(proc)
    out = proc.stdout.read()
    assert "engine pool over 2 GPUs (0,0), 1 handle(s) each" in out
    lines = [l for l in out.splitlines() if l.startswith(" [hip]") and "iterations=6" in l]
    assert len(lines) == 6, out[-3000:]
    pools = [l.rsplit("pool=", 1)[1].strip() for l in lines]
    assert sorted(set(pools)) == ["0", "1"] and pools.count("0") == 3 and pools.count("1") == 3, pools      # both pools used, each connection kept its own
    assert sum("structure=reused" in l for l in lines) == 4 and sum("history=1" in l for l in lines) == 4, lines
    for k in range(2):
        assert util.max_vertex_diff(outs[k].v_pos, refs[k]["v_pos"], graphs[k].v_type) < 2e-4    # f32 on the wire
    # a GPU that does not exist is refused at start-up, like a pipeline the reference cannot create
    exe = build.SERVER
    r = subprocess.run([exe, "127.0.0.1", str(_free_port()), "1", "gpu", "cuda", "64", "1e-10", "0,99"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "ConnectionManager error" in r.stderr

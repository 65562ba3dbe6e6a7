"""In-tree builds: libtsgo_hip.so (hipcc, gfx950), libtsgo_host.so (g++), graph_optimizer (server) — the product — and
libtsgo_hip_testing.so: the same sources with -DTSGO_TESTING (test hooks, research variables, the in-process all-reduce group:
csrc/host/knobs.h, include/tsgo_testing.h), loaded only by the tests that need a hook."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HOST_SRC = ["host/problem.cpp", "host/amg.cpp", "host/codec.cpp", "host/synth.cpp", "host/host_api.cpp", "host/errors.cpp"]
HIP_SO = os.path.join(HERE, "libtsgo_hip.so")
HOST_SO = os.path.join(HERE, "libtsgo_host.so")
HIP_TESTING_SO = os.path.join(HERE, "libtsgo_hip_testing.so")
SERVER = os.path.join(HERE, "graph_optimizer")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _all_sources():
    out = []
    for root, _d, files in os.walk(CSRC):
        out += [os.path.join(root, f) for f in files if f.endswith((".h", ".hip", ".cpp", ".inc"))]
    out.append(os.path.join(os.path.dirname(HERE), "include", "tsgo.h"))
    out.append(os.path.join(os.path.dirname(HERE), "include", "tsgo_testing.h"))
    return out


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build_host(force=False):
    if force or _newer(HOST_SO, _all_sources()):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-pthread", "-shared", "-o", HOST_SO] + HOST_SRC
        subprocess.check_call(cmd, cwd=CSRC)
    return HOST_SO


def build_hip(force=False):
    if force or _newer(HIP_SO, _all_sources()):
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
               "-o", HIP_SO, "tsgo_hip.hip"] + HOST_SRC + ["-lrccl", "-lpthread"]
        subprocess.check_call(cmd, cwd=CSRC)
    return HIP_SO


def build_hip_testing(force=False):
    if force or _newer(HIP_TESTING_SO, _all_sources()):
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result", "-DTSGO_TESTING",
               "-o", HIP_TESTING_SO, "tsgo_hip.hip"] + HOST_SRC + ["-lrccl", "-lpthread"]
        subprocess.check_call(cmd, cwd=CSRC)
    return HIP_TESTING_SO


def build_server(force=False):
    src = os.path.join(CSRC, "host", "server.cpp")
    if not os.path.exists(src):
        return None
    if force or _newer(SERVER, _all_sources()):
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-result", "-o", SERVER,
               "host/server.cpp", "tsgo_hip.hip"] + HOST_SRC + ["-lrccl", "-lpthread"]
        subprocess.check_call(cmd, cwd=CSRC)
    return SERVER


def build_all(force=False):
    """The three product binaries (returned) and the testing twin of the HIP library; the two hipcc jobs of the libraries run side by side."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(3) as ex:
        jobs = [ex.submit(f, force) for f in (build_hip, build_hip_testing, build_server)]
        host = build_host(force)
        hip, _testing, server = [j.result() for j in jobs]
    return host, hip, server


if __name__ == "__main__":
    print(build_all(force=True))

#!/bin/bash
# SURVEY section 5, "race detection / sanitizers" (the reference has one compute-sanitizer command line, cmds.txt:9):
# the host side of the product (layout builder, multigrid patterns, wire codec, synthetic graphs, C-ABI glue) built with
# AddressSanitizer + UndefinedBehaviorSanitizer and driven by the CPU tests that exercise it, malformed wire input
# included.  CPU ONLY — never on the GPU box (GPU sanitizers are not available on this pool).
#   usage: tools/sanitize_host.sh [log file]
set -euo pipefail
cd "$(dirname "$0")/.."
out=${1:-profiles/r04_sanitizers_host.log}
so=/tmp/libtsgo_host_asan.so
src="host/problem.cpp host/amg.cpp host/codec.cpp host/synth.cpp host/host_api.cpp host/errors.cpp"
(cd toyslam_amd/csrc && g++ -O1 -g -std=c++17 -fPIC -Wall -pthread -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -shared -o $so $src)
{
  echo "g++ -fsanitize=address,undefined -fno-sanitize-recover=undefined: $src"
  echo "tests: test_codec.py test_abi_symbols.py test_layout_and_twin.py (layout / shard planner / multigrid patterns), with TSGO_HOST_SO=$so"
  LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
    TSGO_HOST_SO=$so TSGO_HOST_THREADS=4 python -m pytest tests/test_codec.py tests/test_abi_symbols.py tests/test_layout_and_twin.py -x -q 2>&1 | tail -15
  echo "fuzz: 3000 truncated / bit-flipped wire requests through tsgo_wire_decode + encode_response"
  LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 \
    TSGO_HOST_SO=$so python tools/fuzz_codec.py 3000 2>&1 | tail -5
  echo "ThreadSanitizer: layout (parallel slot fill) + multigrid patterns (parallel products, helper thread for the next level's aggregates), whole graph and one shard of two, 6 threads"
  (cd toyslam_amd/csrc && g++ -O1 -g -std=c++17 -pthread -fsanitize=thread -o /tmp/tsgo_tsan_driver ../../tools/tsan_driver.cpp $src)
  TSGO_HOST_THREADS=6 /tmp/tsgo_tsan_driver 2>&1 | tail -12
} | tee "$out"

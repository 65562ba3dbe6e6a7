#!/bin/bash
# set_graph wall time at 100k poses against the number of host threads per parallel section (research; GPU box)
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)  cpu.stat:"; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
for t in 4 8 12 16 24 32 48; do
  echo "== TSGO_HOST_THREADS=$t"
  TSGO_HOST_THREADS=$t python tools/research/setgraph_timing.py c3_100k 2>&1 | grep "set_graph "
done
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8

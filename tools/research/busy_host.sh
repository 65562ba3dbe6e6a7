#!/bin/bash
# Research (GPU box): bench.py beside N CPU-burning processes: how do paced eager launches (use_graphs = 2) and hipGraph replay take a busy host?
#   usage: tools/research/busy_host.sh N
N=${1:-32}
pids=()
for i in $(seq $N); do python3 -c "
import time
t=time.time()
while time.time()-t < 75: pass" & pids+=($!); done
sleep 2
for mode in "" "--graphs" "" "--graphs"; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu --no-conv $mode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('burners $N mode [$mode]', round(d['ms_per_step'],3), round(d['roofline']['ms_solve_outside_iterations'],3), d['config']['launch_mode'])"
done
for p in "${pids[@]}"; do kill $p 2>/dev/null; done
wait 2>/dev/null

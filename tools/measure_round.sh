#!/bin/bash
# The round's measurement set for one workload, on the GPU box (run through gpurun from the repo root):
#   tools/measure_round.sh <out-tag> [workload=c3_100k] [steps=20] [warmup=5]
# 1. rocprofv3 --kernel-trace --stats around the driver-style bench command       -> <out>/kernel_stats.csv
#    (--no-cpu --no-conv: the timed steps and the probes of THIS workload only — the convergence runs at configs 1 and 2 launch the
#    same kernel symbols on graphs 10-700x smaller and would pull every average down)
# 2. rocprofv3 --pmc passes (tools/pmc_traffic.py: FETCH_SIZE, WRITE_SIZE apart)  -> <out>/<tag>_pmc_digest.txt, _pmc_traffic.json
# 3. the bench line itself, AFTER 1-2 were copied to where bench.py reads them    -> <out>/final_bench.json
# 4. tools/check_roofline.py: every fraction recomputed from the CSV              -> <out>/check_roofline.txt
# For c3_100k the CSV / JSON go to profiles/r04_rocprofv3_kernel_stats.csv / r04_pmc_traffic.json (what bench.py reads); other
# workloads keep theirs under <out>/ (copy them to profiles/ by hand with the workload in the name).
set -eo pipefail
TAG=$1; W=${2:-c3_100k}; STEPS=${3:-20}; WARM=${4:-5}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o stats -- python3 "$ROOT/bench.py" --steps $STEPS --warmup $WARM --workload $W --no-cpu --no-conv > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.err"
STATS=$(find "$OUT/prof" -name '*kernel_stats.csv' | head -1)
cp "$STATS" "$OUT/kernel_stats.csv"
echo "stats: $STATS"
python3 "$ROOT/tools/pmc_traffic.py" "$OUT" "$TAG" $W > "$OUT/pmc_stdout.txt" 2>&1
if [ "$W" = c3_100k ]; then
  cp "$OUT/kernel_stats.csv" "$ROOT/profiles/r04_rocprofv3_kernel_stats.csv"
  cp "$OUT/${TAG}_pmc_traffic.json" "$ROOT/profiles/r04_pmc_traffic.json"
fi
cd "$ROOT"
python3 bench.py --steps $STEPS --warmup $WARM --workload $W > "$OUT/final_bench.json" 2> "$OUT/bench.err"
python3 tools/check_roofline.py "$OUT/final_bench.json" "$OUT/kernel_stats.csv" > "$OUT/check_roofline.txt" 2>&1 || true
tail -3 "$OUT/check_roofline.txt"

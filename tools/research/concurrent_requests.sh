for n in 1 2 3 4; do
  echo "== $n concurrent processes"
  for k in $(seq 1 $n); do python bench.py --no-cpu --no-conv --steps 40 --warmup 4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%.1f it/s' % d['gn_iters_per_s'])" & done; wait
done

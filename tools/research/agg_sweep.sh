# usage: agg_sweep.sh "mode:list[:env=val,...]" ...   (research: aggregation variants through bench.py)
for cfg in "$@"; do
  IFS=':' read -r mode list extra <<< "$cfg"
  echo "== $mode $list $extra"
  env TSGO_AGG_MODE=$mode TSGO_AGG_LIST=$list ${extra//,/ } python bench.py --no-cpu --no-conv ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('it/s %.1f  pcg %.1f  lin %.2f solve %.2f  us/iter %.0f setup_us %.0f host_setup_ms %.0f fallbacks %s' % (d['gn_iters_per_s'], d['pcg_iters_per_gn_iter'], d['ms_per_step_device']['linearize'], d['ms_per_step_device']['solve'], r['us_per_pcg_iteration'], r['us_multigrid_numeric_setup'], d['ms_setup_once_per_graph'], d.get('pcg_fallbacks')))"
done

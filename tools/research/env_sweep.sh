# usage: env_sweep.sh "VAR=val,VAR2=val2" ...   (research: engine knobs through bench.py; "-" = defaults)
for cfg in "$@"; do
  echo "== $cfg"
  e="${cfg//,/ }"; [ "$cfg" = "-" ] && e=""
  env $e python bench.py --no-cpu ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d.get('iters_to_chi2_tol',{}); print('ms/step %.3f  it/s %.1f  pcg %.2f  lin %.2f solve %.2f  us/iter %.0f  outside %.3f  setup_us %.0f | 50 its: %.3f s, pcg total %s, chi2 last %.6f' % (d['ms_per_step'], d['gn_iters_per_s'], d['pcg_iters_per_gn_iter'], d['ms_per_step_device']['linearize'], d['ms_per_step_device']['solve'], r['us_per_pcg_iteration'], r['ms_solve_outside_iterations'], r['us_multigrid_numeric_setup'] or 0, c.get('seconds',0), c.get('pcg_iters_total'), c.get('chi2_last',0)))"
done

# usage: ab_env3.sh "A=1 B=2" "A=0" ... : the driver's bench invocation (--steps 20 --warmup 5) once per environment setting, 3 rounds interleaved
for rep in 1 2 3; do for v in "$@"; do
  echo -n "[$v] "; env $v timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=b.get('generator_fwd_bwd') or {}; print(round(b['value'],1), 'img/s', round(b['ms_per_step'],3), 'ms/step', round(b['ms_per_step_median'],3), 'median; G fwd+bwd', round(g.get('ms',0),3), 'ms', round(g.get('frac_of_fp32_mfma_peak_executed',0),4), '; launches', b.get('launches_per_step'))"
done; done

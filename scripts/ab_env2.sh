# usage: ab_env2.sh "A=1 B=2" "A=0" ... : default bench once per environment setting, interleaved twice
for rep in 1 2; do for v in "$@"; do
  echo -n "[$v] "; env $v timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 60 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(b['value'],1), 'img/s', round(b['ms_per_step_median'],3), 'ms median')"
done; done

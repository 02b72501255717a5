# usage: ab_env_long.sh VAR v1 v2 ... : 3 interleaved repetitions of a 150-step default bench per value
VAR=$1; shift
for rep in 1 2 3; do for v in "$@"; do
  echo -n "$VAR=$v "; env $VAR=$v timeout 300 python bench.py --no-cpu-baseline --no-kernel-events --steps 150 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(b['value'],1), 'img/s', round(b['ms_per_step_median'],3), 'ms median')"
done; done

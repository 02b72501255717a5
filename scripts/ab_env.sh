# usage: ab_env.sh VAR v1 v2 ... : default bench (no cpu baseline, no eager event steps) once per value of VAR, interleaved twice
VAR=$1; shift
for rep in 1 2; do for v in "$@"; do
  echo -n "$VAR=$v "; env $VAR=$v timeout 300 python bench.py --no-cpu-baseline --no-kernel-events --steps 60 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(b['value'],1), 'img/s', round(b['ms_per_step_median'],3), 'ms median')"
done; done

# usage: ab_env_w.sh "<bench args>" VAR v1 v2 ...
ARGS=$1; VAR=$2; shift; shift
for rep in 1 2; do for v in "$@"; do
  echo -n "$VAR=$v "; env $VAR=$v timeout 600 python bench.py $ARGS --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(b['value'],1), b['unit'], round(b['ms_per_step'],3), 'ms')"
done; done

# usage: ab_secondary.sh <workload: train_residual|train_mru|bg768_train|bg768|fg_infer> "ENV=1" "ENV=0" ... : one secondary bench per setting, 2 rounds interleaved
W=$1; shift
case $W in
  train_residual) ARGS="--block-type Residual --steps 20 --warmup 3 --preheat-seconds 1";;
  train_mru) ARGS="--block-type MRU --steps 6 --warmup 3 --preheat-seconds 1";;
  bg768_train) ARGS="--workload bg768_train --steps 30 --warmup 5";;
  bg768) ARGS="--workload bg768 --steps 30 --warmup 5";;
  fg_infer) ARGS="--workload fg_infer --steps 100 --warmup 10";;
esac
for rep in 1 2; do for v in "$@"; do
  echo -n "[$W $v] "; env $v timeout 600 python bench.py --no-cpu-baseline --no-secondary $ARGS 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(b['value'],1), 'img/s', round(b['ms_per_step'],3), 'ms; frac', round(b.get('step_frac_of_fp32_peak') or 0,4), '; launches', b.get('launches_per_step'))"
done; done

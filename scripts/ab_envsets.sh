# usage: ab_envsets.sh <reps> "<VAR=v VAR2=v2>" "<...>" ... : the driver's bench invocation (headline only) once per environment
# set ("-" = defaults), interleaved <reps> times on ONE box.  Output -> gpurun_out/ab_envsets.txt
REPS=$1; shift
mkdir -p gpurun_out
OUT=gpurun_out/ab_envsets.txt
: > $OUT
line() { python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=b.get('generator_fwd_bwd') or {}; print(round(b['value'],1), 'img/s', round(b['ms_per_step'],3), 'ms/step', 'gen_fb_ms', round(g.get('ms') or 0, 3))"; }
for rep in $(seq $REPS); do
  for E in "$@"; do
    echo -n "[$E rep $rep] " | tee -a $OUT
    if [ "$E" = "-" ]; then EV=""; else EV="$E"; fi
    env $EV timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 20 --warmup 5 2>/dev/null | line | tee -a $OUT
  done
done

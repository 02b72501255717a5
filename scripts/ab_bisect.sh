# usage: ab_bisect.sh <reps> <dir> <dir> ... : the driver's bench invocation (headline only) in each checkout ("." = this
# tree), interleaved on ONE box.  Output -> gpurun_out/ab_bisect.txt
REPS=$1; shift
ROOT=$PWD
mkdir -p gpurun_out
OUT=$ROOT/gpurun_out/ab_bisect.txt
: > $OUT
line() { python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=b.get('generator_fwd_bwd') or {}; print(round(b['value'],1), 'img/s', round(b['ms_per_step'],3), 'ms/step', 'gen_fb_ms', round(g.get('ms') or 0, 3))"; }
for rep in $(seq $REPS); do
  for T in "$@"; do
    cd $ROOT/$T
    echo -n "[$T rep $rep] " | tee -a $OUT
    timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 20 --warmup 5 2>/dev/null | line | tee -a $OUT
  done
done
cd $ROOT

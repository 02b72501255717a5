# usage: ab_trees.sh <other checkout dir> [reps] : the driver's bench invocation (headline only), this tree vs another checkout
# (its own Python + its own in-tree library), interleaved on ONE box.  Output -> gpurun_out/ab_trees.txt
ALT=$1; REPS=${2:-3}
ROOT=$PWD
mkdir -p gpurun_out
OUT=$ROOT/gpurun_out/ab_trees.txt
: > $OUT
line() { python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=b.get('generator_fwd_bwd') or {}; print(round(b['value'],1), 'img/s', round(b['ms_per_step'],3), 'ms/step', 'gen_fb_ms', g.get('ms'))"; }
for rep in $(seq $REPS); do
  for T in this alt; do
    if [ $T = this ]; then cd $ROOT; else cd $ROOT/$ALT; fi
    echo -n "[$T rep $rep] " | tee -a $OUT
    timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 20 --warmup 5 2>/dev/null | line | tee -a $OUT
  done
done
cd $ROOT

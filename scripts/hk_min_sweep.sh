#!/bin/bash
# lower bound (in fills of the chip's 2 x CUs workgroup slots) from which the planner may take the 128 x 128 / 16-k tile
one() { python bench.py "$@" --no-cpu-baseline --no-secondary --no-kernel-events --no-gen-fb 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%8.3f ms' % j['ms_per_step'], end='  ')"; }
for r in 1 2; do
  for v in ${HK_MINS:-0 0.5 1 2 100}; do
    export SSC_PLAN_BF_HK_MIN=$v
    printf "HK min %s: pix2pix | mru | residual | fg_infer | bg768 | bg768_train: " $v
    one --steps 30 --warmup 5 --preheat-seconds 1
    one --block-type MRU --steps 8 --warmup 3 --preheat-seconds 0
    one --block-type Residual --steps 15 --warmup 3 --preheat-seconds 0
    one --workload fg_infer --steps 100 --warmup 10
    one --workload bg768 --steps 30 --warmup 5
    one --workload bg768_train --steps 30 --warmup 5
    echo
  done
done

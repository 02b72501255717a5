#!/bin/bash
# planner price of the 128 x 128 / 16-k tile (SSC_PLAN_BF_HK: relative cost per MFMA against 64 x 128 = 1.00): train steps per value
one() { python bench.py "$@" --no-cpu-baseline --no-secondary --no-kernel-events --no-gen-fb 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%8.3f ms' % j['ms_per_step'], end='  ')"; }
for r in 1 2; do
  for v in ${HK_VALUES:-0.86 0.90 0.94 0.98 1.04}; do
    export SSC_PLAN_BF_HK=$v
    printf "HK price %s: pix2pix | mru | residual | bg768 fwd: " $v
    one --steps 30 --warmup 5 --preheat-seconds 1
    one --block-type MRU --steps 8 --warmup 3 --preheat-seconds 0
    one --block-type Residual --steps 15 --warmup 3 --preheat-seconds 0
    one --workload bg768 --steps 30 --warmup 5
    echo
  done
done

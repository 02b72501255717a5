#!/bin/bash
# 128 x 128 tiles on 16-k stages (conv_bfh_kernel, planner-priced) vs round 5's tile set: train steps and forward workloads, interleaved
one() { python bench.py "$@" --no-cpu-baseline --no-secondary --no-kernel-events --no-gen-fb 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%8.1f img/s %8.3f ms' % (j['value'], j['ms_per_step']))"; }
for r in 1 2; do
  for v in hk old; do
    if [ $v = old ]; then export SSC_DEV_SWITCHES=1 SSC_BF_HK=0; else unset SSC_DEV_SWITCHES SSC_BF_HK; fi
    echo "== $v: pix2pix train | mru train | residual train | fg_infer | bg768 | bg768_train"
    one --steps 30 --warmup 5 --preheat-seconds 1
    one --block-type MRU --steps 8 --warmup 3 --preheat-seconds 0
    one --block-type Residual --steps 15 --warmup 3 --preheat-seconds 0
    one --workload fg_infer --steps 100 --warmup 10
    one --workload bg768 --steps 30 --warmup 5
    one --workload bg768_train --steps 30 --warmup 5
  done
done

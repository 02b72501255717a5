#!/bin/bash
# bench.py under a few planner-constant settings (tuning aid for plan_const in igemm.hip)
run() { out=$(env "$@" python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1); echo "$* -> $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],3))')"; }
run X=0
run SSC_PLAN_REDUCE=3000
run SSC_PLAN_REDUCE=1000
run SSC_PLAN_OCC2=1.04
run SSC_PLAN_OCC1=1.2
run SSC_TAIL_SPLIT=0
run X=0

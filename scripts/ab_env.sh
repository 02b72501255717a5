#!/bin/bash
# usage: scripts/ab_env.sh VAR v1 v2 ... : bench.py once per value of the environment variable, alternating twice
var=$1; shift
for rep in 1 2; do for v in "$@"; do
  out=$(env $var=$v python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$var=$v $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],3))')"
done; done

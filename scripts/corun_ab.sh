# small-LDS form of the conv launches inside the train steps (SSC_CO_RUN: 0 never, 1 Pix2Pix pair (default), all)
bash scripts/ab_env3.sh "SSC_CO_RUN=0" "SSC_CO_RUN=1"
for bt in Residual MRU; do for r in 1 2; do for S in 0 all; do
  echo -n "[$bt step, SSC_CO_RUN=$S] "; SSC_CO_RUN=$S python bench.py --block-type $bt --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done; done; done

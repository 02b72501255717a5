# bf16x6 filter-gradient kernel: one LDS stage (two workgroups per CU, default) vs two stages (SSC_WGBF_DB=1, one workgroup per CU)
export SSC_DEV_SWITCHES=1
for D in 1 0; do
  export SSC_WGBF_DB=$D; echo "== SSC_WGBF_DB=$D"
  for layer in wg3 wg3p wg3n; do python scripts/conv_microbench.py $layer 100 32 2>&1 | tail -1; done
done
for r in 1 2 3; do for D in 1 0; do
  export SSC_WGBF_DB=$D; echo -n "[step, SSC_WGBF_DB=$D] "; python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done; done
for D in 1 0; do export SSC_WGBF_DB=$D; echo -n "[MRU step, SSC_WGBF_DB=$D] "; python bench.py --block-type MRU --steps 6 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; done

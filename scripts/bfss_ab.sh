# bf16x6 conv kernel: one LDS stage per operand (SSC_BF_SS=1: 42 KB, three workgroups per CU by LDS) vs two (75 KB)
export SSC_DEV_SWITCHES=1
for S in 0 1; do
  export SSC_BF_SS=$S; echo "== SSC_BF_SS=$S"
  for layer in enc2 enc3 enc4 d4 dec3 dg3; do python scripts/conv_microbench.py $layer 100 32 2>&1 | tail -1; done
done
bash scripts/ab_env3.sh "SSC_DEV_SWITCHES=1 SSC_BF_SS=0" "SSC_DEV_SWITCHES=1 SSC_BF_SS=1"
for S in 0 1; do export SSC_BF_SS=$S; echo -n "[MRU step, SSC_BF_SS=$S] "; python bench.py --block-type MRU --steps 6 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; done

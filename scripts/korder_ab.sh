# K-tile order of the bf16x6 conv kernel (SSC_BF_KORDER: 0 = chunk fastest, default = tap fastest / parity classes): layer shapes alone,
# HBM-side bytes of two layers, the train step
export SSC_DEV_SWITCHES=1
for K in 0 default; do
  if [ $K = default ]; then unset SSC_BF_KORDER; else export SSC_BF_KORDER=$K; fi
  echo "== SSC_BF_KORDER=$K"
  for layer in enc2 enc3 enc4 d4 dec3 dg3; do python scripts/conv_microbench.py $layer 100 32 2>&1 | tail -1; done
  for layer in enc2 enc3 dec3; do bash scripts/pmc_traffic_one.sh $layer 32; done
done
for r in 1 2; do
  for K in 0 default; do
    if [ $K = default ]; then unset SSC_BF_KORDER; else export SSC_BF_KORDER=$K; fi
    echo "== step, SSC_BF_KORDER=$K"; python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['generator_fwd_bwd'].get('ms') if isinstance(d.get('generator_fwd_bwd'), dict) else None)"
  done
done

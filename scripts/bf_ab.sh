# microbenchmarks of the layer shapes, bf16-split vs exact fp32 (SSC_ARITH), optionally a forced tile (SSC_FWD_CFG)
for A in bf16x6 fp32; do
  echo "== SSC_ARITH=$A ${SSC_FWD_CFG:+cfg $SSC_FWD_CFG}"
  for layer in enc2 enc3 enc4 d4 dec3 dg3; do SSC_ARITH=$A python scripts/conv_microbench.py $layer 100 ${1:-32} 2>&1 | tail -1; done
done

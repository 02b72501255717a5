export SSC_DEV_SWITCHES=1
for SK in default 4 8 12 16 24 32; do
  if [ $SK = default ]; then unset SSC_WG128_SPLITK; else export SSC_WG128_SPLITK=$SK; fi
  echo -n "splitk=$SK: "; python scripts/conv_microbench.py wg3 100 32 2>&1 | tail -1
done

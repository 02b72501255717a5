# usage: ab_lib.sh <other lib> : microbenchmarks (sustained) + the driver's bench invocation, default library vs another build
ALT=$1
for L in default $ALT; do
  if [ $L = default ]; then unset SSC_LIB_PATH; else export SSC_LIB_PATH=$PWD/$ALT; fi
  echo "== $L"
  for layer in enc2 enc3 enc4 d4 dec3 dg3; do python scripts/conv_microbench.py $layer 100 32 2>/dev/null | tail -1; done
done
for rep in 1 2 3; do for L in default $ALT; do
  if [ $L = default ]; then unset SSC_LIB_PATH; else export SSC_LIB_PATH=$PWD/$ALT; fi
  echo -n "[$L] "; timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(b['value'],1), 'img/s', round(b['ms_per_step'],3), 'ms/step')"
done; done

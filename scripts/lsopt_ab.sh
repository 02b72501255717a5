# A/B of building igemm.hip without the load/store merger (SSC_NO_LSOPT=1): rebuilds the library on the GPU box
for v in 1 0 1 0; do
  SSC_NO_LSOPT=$v python -c "from sketchyscenecolorization_amd.build import build_library; build_library(force=True, verbose=False)" > /dev/null 2>&1
  echo -n "SSC_NO_LSOPT=$v "; timeout 300 python bench.py --no-cpu-baseline --no-kernel-events --steps 150 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(b['value'],1), 'img/s', round(b['ms_per_step_median'],3), 'ms median')"
done

for x in 0 1; do export SSC_WG_XCD=$x; echo "== SSC_WG_XCD=$x"; python scripts/conv_microbench.py wg3 50 2>&1 | grep -v amdgpu; bash scripts/pmc_traffic_one.sh wg3 32 2>&1 | grep -v amdgpu | head -4; done
unset SSC_WG_XCD
bash scripts/ab_env.sh SSC_WG_XCD 0 1
timeout 600 python -m pytest tests/test_gpu_igemm.py -m gpu -x -q -k "wgrad" 2>&1 | tail -2

for x in 1 2; do export SSC_XCD_ORDER=$x; echo "== SSC_XCD_ORDER=$x"; python scripts/conv_microbench.py dg3 50 2>&1 | grep -v amdgpu; bash scripts/pmc_traffic_one.sh dg3 32 2>&1 | grep -v amdgpu | head -6; done
unset SSC_XCD_ORDER
bash scripts/ab_env.sh SSC_XCD_ORDER 1 2

# layouts of whole tiles + in-launch K slices on the probe shapes: SSC_TS_FORCE="whole tiles per CU,slices per remaining tile"
export SSC_FWD_CFG=${1:-1}
for f in "9,1" "0,2" "0,3" "0,4" "0,8" "1,2" "1,4" "1,8" "2,2" "2,4" "3,2" "4,2" "4,4"; do
  echo "== SSC_TS_FORCE=$f cfg=$SSC_FWD_CFG"; SSC_TS_FORCE=$f timeout 300 python scripts/steady_state_probe.py 2>&1 | grep "batch 32"
done

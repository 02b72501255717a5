# tail-split layouts for the bf16x6 conv kernel on the batch-32 layer shapes (SSC_TS_FORCE="whole tiles per 256 CUs,slices per remaining tile")
run() { for f in $2; do SSC_TS_FORCE=$f SSC_FWD_CFG=1 timeout 120 python scripts/shape_probe.py $1 2>&1 | grep TFLOP | sed "s/^/[$1] /"; done; }
run "32 96 96 64 128 4 2" "9,1 4,2 4,3 4,4 4,8 2,2 3,2 0,2"
run "32 48 48 128 256 4 2" "9,1 2,4 2,8 2,2 1,2 1,4 0,2 0,4"
run "32 24 24 256 512 4 2" "9,1 1,2 1,4 1,8 0,2 0,4 0,8"
run "32 12 12 512 512 4 2" "9,1 0,2 0,4 0,8"

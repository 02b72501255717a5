# diagnostic build of wgrad128.hip with cycle stamps + the probe (scripts/wg128_timing.py); SSC_WG128_SPLITK picks the K slices
set -e
L=sketchyscenecolorization_amd/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DSSC_WG128_TIMING $SSC_EXTRA -Wno-unused-value -Wno-unused-function -c sketchyscenecolorization_amd/csrc/wgrad128.hip -o $L/wgrad128_timing.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libssc_timing.so $L/igemm.o $L/wgrad128_timing.o $L/narrow.o $L/fewchan.o $L/elementwise.o $L/text_lstm.o $L/losses_optim.o $L/mru_ops.o

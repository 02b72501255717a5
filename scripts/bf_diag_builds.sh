#!/bin/bash
# What bounds conv_bf_kernel's K step?  Builds csrc/igemm_bf16.hip with SSC_BF_DIAG_BUILD = each mask in $MASKS (results are wrong,
# timing is what is asked for) into lab/diag/libssc_diag<mask>.so -- here, without a GPU -- and, with "run", times the six layer
# shapes with each of them on the GPU box (scripts/conv_microbench.py through SSC_LIB_PATH).
#   bash scripts/bf_diag_builds.sh build          (this container)
#   bash scripts/bf_diag_builds.sh run [batch]    (GPU box)
MASKS=${MASKS:-"0 1 2 4 8 16 32 48 64 24 26 94"}
cd "$(dirname "$0")/.."
PKG=sketchyscenecolorization_amd
if [ "$1" = build ]; then
  mkdir -p lab/diag
  python -m $PKG.build > /dev/null || exit 1
  for m in $MASKS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wno-unused-value -Wno-unused-function -fno-slp-vectorize \
      -mllvm -amdgpu-mfma-vgpr-form=1 -DSSC_BF_DIAG_BUILD=$m $DIAG_EXTRA -c $PKG/csrc/igemm_bf16.hip -o lab/diag/igemm_bf16_$m$TAG.o || exit 1
    objs=$(ls $PKG/lib/*.o | grep -v igemm_bf16.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lab/diag/libssc_diag$m$TAG.so $objs lab/diag/igemm_bf16_$m$TAG.o || exit 1
    rm lab/diag/igemm_bf16_$m$TAG.o
    echo built lab/diag/libssc_diag$m$TAG.so
  done
  exit 0
fi
B=${2:-32}
for m in $MASKS; do
  printf "diag %3d:" $m
  for layer in enc2 enc3 enc4 d4 dec3 dg3; do
    SSC_ALLOW_STALE_LIB=1 SSC_LIB_PATH=$PWD/lab/diag/libssc_diag$m$TAG.so python scripts/conv_microbench.py $layer 60 $B 2>/dev/null | tail -1 | awk '{printf "  %s %6.1f", $1, $4}'
  done
  echo
done

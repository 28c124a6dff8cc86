#!/bin/bash
# The library with extra compiler flags (an A/B of a compile-time choice) as convnet_amd/lib/libconvnet_hip_<suffix>.so, for same-call
# runs through CONVNET_HIP_LIB (convnet_amd/_lib.py).  Usage: bash tools/build_variant_lib.sh "-DCONVNET_GPV_FILT_LATE=0" early
set -e
R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d)
cd "$R/convnet_amd/csrc"
for f in *.hip; do
  extra=""; case $f in patch_gemm.hip|wgrad_wide.hip|fewc_conv.hip) extra="-fno-slp-vectorize";; esac
  (hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -Wno-inline-asm $1 $extra -c "$f" -o "$T/${f%.hip}.o" || touch "$T/FAILED") &
done; wait
[ -e "$T/FAILED" ] && { echo "build_variant_lib: a source failed to compile" >&2; exit 1; }
hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/convnet_amd/lib/libconvnet_hip_$2.so" "$T"/*.o -ldl
echo "$R/convnet_amd/lib/libconvnet_hip_$2.so"

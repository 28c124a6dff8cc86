#!/bin/bash
# The library with the experiment knobs compiled in (-DCONVNET_DIAG) as convnet_amd/lib/libconvnet_hip_diag.so, for A/B runs through
# CONVNET_HIP_LIB=libconvnet_hip_diag.so (convnet_amd/_lib.py).  Never what the product, the tests or bench.py's judged legs load.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d)
cd "$R/convnet_amd/csrc"
for f in *.hip; do
  extra=""; case $f in patch_gemm.hip|wgrad_wide.hip|fewc_conv.hip) extra="-fno-slp-vectorize";; esac
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -Wno-inline-asm -DCONVNET_DIAG $extra -c "$f" -o "$T/${f%.hip}.o" &
done; wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/convnet_amd/lib/libconvnet_hip_diag.so" "$T"/*.o -ldl
echo "$R/convnet_amd/lib/libconvnet_hip_diag.so"

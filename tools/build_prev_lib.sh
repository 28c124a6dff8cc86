#!/bin/bash
# Builds the library as it stood at a given commit (default: the last commit of the previous round) into
# convnet_amd/lib/libconvnet_hip_<tag>.so, for same-call A/B runs of tools/ and bench.py through CONVNET_HIP_LIB (convnet_amd/_lib.py).
# Usage: bash tools/build_prev_lib.sh 256646b r03
set -e
REV=${1:-256646b}; TAG=${2:-r03}
R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d)
git -C "$R" archive "$REV" convnet_amd/csrc include | tar -x -C "$T"
cd "$T/convnet_amd/csrc"
for f in *.hip; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -Wno-inline-asm -c "$f" -o "$T/${f%.hip}.o" & done; wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/convnet_amd/lib/libconvnet_hip_$TAG.so" "$T"/*.o -ldl
echo "$R/convnet_amd/lib/libconvnet_hip_$TAG.so"

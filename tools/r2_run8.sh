#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_run8
mkdir -p "$O"; cd "$R" || exit 1
for a in 0 2 1 0 2 1; do CONVNET_GG_PRIO=2 CONVNET_GG_ABLATE=$a timeout 120 python tools/layer_bench.py --only conv4 2>&1 | grep "fprop.*gg_kernel" | sed "s/^/ablate=$a /"; done

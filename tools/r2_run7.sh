#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_run7
mkdir -p "$O"; cd "$R" || exit 1
CONVNET_GG_PRIO=2 timeout 120 python tools/layer_bench.py --only conv4 > "$O/two.log" 2>&1
CONVNET_GG_PRIO=2 CONVNET_GG_LDS_PAD=70000 timeout 120 python tools/layer_bench.py --only conv4 > "$O/one.log" 2>&1
CONVNET_GG_PRIO=0 CONVNET_GG_LDS_PAD=70000 timeout 120 python tools/layer_bench.py --only conv4 > "$O/one_p0.log" 2>&1
grep gg_kernel "$O/two.log" "$O/one.log" "$O/one_p0.log"

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
run() { env "$@" timeout 120 python tools/layer_bench.py --only conv4 2>&1 | grep "fprop.*gg_kernel" | sed "s/^/$* /"; }
run CONVNET_GG_PRODUCER=0 CONVNET_GG_PRIO=2
run CONVNET_GG_PRODUCER=1
run CONVNET_GG_PRODUCER=1 CONVNET_GG_LDS_PAD=20000
run CONVNET_GG_PRODUCER=1 CONVNET_GG_ABLATE=1
run CONVNET_GG_PRODUCER=1 CONVNET_GG_ABLATE=1 CONVNET_GG_LDS_PAD=20000

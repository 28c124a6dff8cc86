#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_run12
mkdir -p "$O"; cd "$R" || exit 1
CONVNET_GG_PRIO=2 timeout 120 python tools/layer_bench.py --only conv > "$O/base.log" 2>&1
CONVNET_GG_PRODUCER=1 timeout 120 python tools/layer_bench.py --only conv > "$O/prod.log" 2>&1
CONVNET_GG_PRODUCER=1 CONVNET_GG_LDS_PAD=20000 timeout 120 python tools/layer_bench.py --only conv4 > "$O/prod_pad.log" 2>&1
paste <(grep "gg_kernel\|ggp_kernel" "$O/base.log" | grep -v "1,4,1,128" | sort | awk '{print $1,$2,$3,$(NF-1),$NF}') <(grep "gg_kernel\|ggp_kernel" "$O/prod.log" | grep -v "1,4,1,128" | sort | awk '{print $3,$(NF-1),$NF}')
grep ggp "$O/prod_pad.log"
CONVNET_GG_PRODUCER=1 timeout 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "conv or fc or dot" 2>&1 | tail -2

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_run4
mkdir -p "$O"; cd "$R" || exit 1
for l in "conv4 fprop" "conv5 fprop" "conv3 dgrad" "conv1 fprop" "conv2 dgrad"; do timeout 60 tools/gg_trace $l 2>&1 | tee -a "$O/trace.log"; echo; done
echo "== pool/rnorm bench: default (XCD order), no XCD, undo LT 32"
timeout 120 python tools/pool_bench.py > "$O/pool_default.log" 2>&1
CONVNET_RNORM_NO_XCD=1 timeout 120 python tools/pool_bench.py > "$O/pool_noxcd.log" 2>&1
CONVNET_RNORM_UNDO_LT=32 timeout 120 python tools/pool_bench.py > "$O/pool_lt32.log" 2>&1
CONVNET_RNORM_UNDO_LT=8 timeout 120 python tools/pool_bench.py > "$O/pool_lt8.log" 2>&1
CONVNET_RNORM_FWD_LT=32 timeout 120 python tools/pool_bench.py > "$O/pool_fwd32.log" 2>&1
paste <(grep -v amdgpu "$O/pool_default.log") <(grep -v amdgpu "$O/pool_noxcd.log" | awk '{print $(NF-1),$NF}') <(grep -v amdgpu "$O/pool_lt32.log" | awk '{print $(NF-1),$NF}') <(grep -v amdgpu "$O/pool_lt8.log" | awk '{print $(NF-1),$NF}') <(grep -v amdgpu "$O/pool_fwd32.log" | awk '{print $(NF-1),$NF}')
timeout 100 python -m pytest tests/test_hip_parity.py -q -m gpu -k "norm" 2>&1 | tail -2

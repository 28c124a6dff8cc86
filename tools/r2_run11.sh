#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_run11
mkdir -p "$O"; cd "$R" || exit 1
echo "== layer bench: default | producer"
CONVNET_GG_PRIO=2 timeout 120 python tools/layer_bench.py > "$O/base.log" 2>&1
CONVNET_GG_PRODUCER=1 timeout 120 python tools/layer_bench.py > "$O/prod.log" 2>&1
grep -v "reduce\|filter\|tail_fix\|amdgpu\|wg_kernel" "$O/base.log"; echo; grep -v "reduce\|tail_fix\|amdgpu\|wg_kernel" "$O/prod.log"
echo "== parity with the producer-wave kernel"
CONVNET_GG_PRODUCER=1 timeout 600 python -m pytest tests/test_hip_parity.py tests/test_net_gpu.py -x -q -m gpu > "$O/parity.log" 2>&1; echo "rc=$?"; tail -5 "$O/parity.log"

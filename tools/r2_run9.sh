#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_run9
mkdir -p "$O"; cd "$R" || exit 1
echo "== parity with the producer-wave kernel"
CONVNET_GG_PRODUCER=1 timeout 600 python -m pytest tests/test_hip_parity.py tests/test_net_gpu.py tests/test_full_geometry_gpu.py -x -q -m gpu -k "not batch_256 and not ref]" > "$O/parity.log" 2>&1; echo "rc=$?"; tail -15 "$O/parity.log"
echo "== layer bench: default | producer | producer prio2"
CONVNET_GG_PRIO=2 timeout 120 python tools/layer_bench.py > "$O/base.log" 2>&1
CONVNET_GG_PRODUCER=1 timeout 120 python tools/layer_bench.py > "$O/prod.log" 2>&1
paste <(awk '{print $1,$2,$3,$(NF-1),$NF}' "$O/base.log") <(awk '{print $3,$(NF-1),$NF}' "$O/prod.log") | grep -v "reduce\|filter\|tail_fix\|amdgpu\|wg_kernel"

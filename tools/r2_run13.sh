#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_run13
mkdir -p "$O"; cd "$R" || exit 1
CONVNET_GG_PRODUCER=1 timeout 120 python tools/layer_bench.py --only conv > "$O/prod.log" 2>&1
grep "ggp_kernel" "$O/prod.log"
CONVNET_GG_PRODUCER=1 timeout 900 python -m pytest tests -x -q -m gpu > "$O/suite.log" 2>&1; echo "rc=$?"; tail -4 "$O/suite.log"
CONVNET_GG_PRODUCER=1 timeout 200 python bench.py --no-cpu-baseline --no-ref-host > "$O/bench.json" 2> "$O/bench.err"; cut -c1-220 "$O/bench.json"

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_run6
mkdir -p "$O"; cd "$R" || exit 1
for g in 1 0 2; do CONVNET_GG_PRIO=$g CONVNET_WG_PRIO=$g timeout 120 python tools/layer_bench.py > "$O/layer_prio$g.log" 2>&1; done
paste <(awk '{print $1,$2,$3,$(NF-1),$NF}' "$O/layer_prio1.log") <(awk '{print $(NF-1),$NF}' "$O/layer_prio0.log") <(awk '{print $(NF-1),$NF}' "$O/layer_prio2.log") | grep -v "reduce\|filter\|tail_fix\|amdgpu"
echo "(columns: gg prio1+wg prio1 | both 0 | both 2)"

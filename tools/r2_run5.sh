#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_run5
mkdir -p "$O"; cd "$R" || exit 1
for l in "conv4 fprop" "conv5 fprop" "conv3 dgrad" "conv1 fprop" "conv2 dgrad" "conv2 fprop"; do echo "---- level 1 (block-level only)"; timeout 60 tools/gg_trace1 $l 2>&1 | tee -a "$O/trace1.log"; done
for l in "conv4 fprop" "conv1 fprop"; do echo "---- level 2"; timeout 60 tools/gg_trace2 $l 2>&1 | tee -a "$O/trace2.log"; done

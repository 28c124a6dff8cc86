#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/split
mkdir -p $O
timeout 120 python tools/layer_bench.py --only conv2 > $O/layer_pre.txt 2>&1
grep -v amdgpu $O/layer_pre.txt | grep "ggp_kernel"

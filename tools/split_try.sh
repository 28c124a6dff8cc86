#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/split
mkdir -p $O
timeout 120 python tools/layer_bench.py > $O/layer_split.txt 2>&1
grep -v amdgpu $O/layer_split.txt | grep "wgrad" | grep -v reduce | head -30
timeout 300 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "outp or golden or conv_up_down or random_geom" 2>&1 | tail -3

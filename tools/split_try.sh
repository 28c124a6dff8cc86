#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/split
mkdir -p $O
timeout 120 python tools/layer_bench.py > $O/layer_alt.txt 2>&1
grep -v amdgpu $O/layer_alt.txt | grep "ggp_kernel\|wg_kernel<\|gg_kernel" | grep -v "4.7\|4.6"
timeout 300 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "split" > $O/par.log 2>&1; tail -2 $O/par.log

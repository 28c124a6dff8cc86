#!/bin/bash
# A/B of the bf16-split products (CONVNET_GG_SPLIT=1) against the default fp32-MFMA path, one gpurun call.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/split
mkdir -p $O
export CONVNET_GG_SPLIT=1
CONVNET_SPLIT_TERMS=8 timeout 600 python -m pytest tests/test_full_geometry_gpu.py -q -m gpu -k "training_pass" > $O/geom_split8.log 2>&1; tail -5 $O/geom_split8.log
CONVNET_SPLIT_TERMS=6 timeout 120 python tools/layer_bench.py > $O/layer_split6.txt 2>&1
CONVNET_SPLIT_TERMS=8 timeout 120 python tools/layer_bench.py > $O/layer_split8.txt 2>&1
paste <(grep -v amdgpu $O/layer_split6.txt | grep -v "reduce\|filter_\|tail_fix" | awk '{print $1,$2,$3,$4,$5}') <(grep -v amdgpu $O/layer_split8.txt | grep -v "reduce\|filter_\|tail_fix" | awk '{print $4,$5}') | head -40
CONVNET_SPLIT_TERMS=8 timeout 200 python bench.py --no-ref-host > $O/bench_split8.json 2> $O/bench_split8.err; cut -c1-300 $O/bench_split8.json

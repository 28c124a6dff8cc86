#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/split
mkdir -p $O
timeout 400 python -m pytest tests/test_net_gpu.py tests/test_data_parallel_gpu.py -q -m gpu -k "side_stream or one_rank_rccl" > $O/ow_tests.log 2>&1; grep -E "passed|failed|error" $O/ow_tests.log | tail -3

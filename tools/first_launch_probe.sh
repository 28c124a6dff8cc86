#!/bin/bash
# hunt the one-off failure of tests/test_hip_parity.py::test_softmax_family_and_fused seen as the FIRST GPU
# test of the FIRST process on a freshly booted box.  Runs that test as the only test of N fresh processes, plain and with
# HIP_LAUNCH_BLOCKING=1, and keeps every failing log.  Must be the first GPU work of the gpurun call to mean anything.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/first_launch
mkdir -p "$O"; cd "$R" || exit 1
fail=0
for i in 1 2 3 4 5 6; do
  env=""; [ $i -gt 4 ] && env="HIP_LAUNCH_BLOCKING=1"
  env $env timeout 120 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k softmax_family > "$O/run$i.log" 2>&1
  rc=$?; echo "run $i ($env) rc=$rc"; [ $rc -ne 0 ] && fail=$((fail+1)) && tail -30 "$O/run$i.log"
done
echo "first-launch probe: $fail failing runs of 6"

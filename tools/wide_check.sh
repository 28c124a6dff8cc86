#!/bin/bash
# First hardware run of gpw_kernel (patch mode 3): parity, then the same-call A/B against the default kernel.
# One gpurun call, every leg under its own timeout (a hang must not become a strike):
#   gpurun --timeout 900 -- 'bash tools/wide_check.sh'
# Stops after the parity leg if that fails; results under gpurun_out/wide_check/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/wide_check
mkdir -p "$O"
cd "$R"
echo "== parity (CONVNET_TEST_PATCH_WIDE=1) =="
CONVNET_TEST_PATCH_WIDE=1 timeout 300 python -m pytest tests/test_patch_gemm_gpu.py -k "wide" -x -q > "$O/parity.log" 2>&1
rc=$?
tail -5 "$O/parity.log"
if [ $rc -ne 0 ]; then echo "parity leg rc=$rc: stopping"; exit $rc; fi
echo "== layer bench, conv3-5, default kernel then patch mode 3, twice (clock ramp) =="
for rep in 1 2; do
  for v in 0 3; do
    CONVNET_GG_PATCH=$v timeout 120 python tools/layer_bench.py --only conv --reps 5 > "$O/layers_v${v}_$rep.log" 2>&1
    echo "patch=$v rep=$rep rc=$?"; grep -E "conv[345]" "$O/layers_v${v}_$rep.log" | head -12
  done
done
echo "== the step: default, then conv3-5 fprop / dgrad on gpw_kernel =="
for v in 0 3; do
  CONVNET_GG_PATCH=$v timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ref-host --no-other-path --no-live-traffic > "$O/bench_v$v.json" 2> "$O/bench_v$v.err"
  echo "patch=$v rc=$?"; python -c "import json,sys; d=json.loads(open('$O/bench_v$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_stream_ms_per_step'))" 2>/dev/null
done
echo "== counters, conv4 =="
MODES="0 3" timeout 500 bash tools/pmc_conv.sh conv4 "$O/pmc_conv4" 2>&1 | tail -12

#!/bin/bash
# First hardware run of gpw_kernel (patch mode 3) and wgw_kernel (wgrad tile 1, §2.2b): parity, then the same-call A/B
# against the default kernels.
# One gpurun call, every leg under its own timeout (a hang must not become a strike):
#   gpurun --timeout 1500 -- 'bash tools/wide_check.sh'
# Stops after the parity leg if that fails; results under gpurun_out/wide_check/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/wide_check
mkdir -p "$O"
cd "$R"
echo "== parity (CONVNET_TEST_PATCH_WIDE=1) =="
CONVNET_TEST_PATCH_WIDE=1 timeout 420 python -m pytest tests/test_patch_gemm_gpu.py -k "wide" -x -q > "$O/parity.log" 2>&1
rc=$?
tail -5 "$O/parity.log"
if [ $rc -ne 0 ]; then echo "gpw parity leg rc=$rc: its A/B legs are skipped"; GPW=0; else GPW=3; fi
CONVNET_TEST_WGRAD_WIDE=1 timeout 420 python -m pytest tests/test_wgrad_wide_gpu.py -x -q > "$O/parity_wgw.log" 2>&1
rc=$?
tail -5 "$O/parity_wgw.log"
if [ $rc -ne 0 ]; then echo "wgw parity leg rc=$rc: its A/B legs are skipped"; WGW=0; else WGW=1; fi
if [ $GPW = 0 ] && [ $WGW = 0 ]; then exit 1; fi
echo "== layer bench, conv2-5: default kernels, then the new ones (those that passed), twice (clock ramp) =="
for rep in 1 2; do
  for v in "0 0" "$GPW $WGW"; do
    set -- $v
    CONVNET_GG_PATCH=$1 CONVNET_WG_TILE=$2 timeout 120 python tools/layer_bench.py --only conv --reps 5 > "$O/layers_p$1_w$2_$rep.log" 2>&1
    echo "patch=$1 wgrad_tile=$2 rep=$rep rc=$?"; grep -E "conv[2345]" "$O/layers_p$1_w$2_$rep.log" | head -16
  done
done
if [ $GPW = 3 ]; then
  echo "== gpw_kernel's variants on conv3-5 fprop / dgrad: 3 one load per step, 4 grouped loads, 5 two-stage filter ring =="
  for v in 3 4 5; do
    CONVNET_GG_PATCH=$v timeout 120 python tools/layer_bench.py --only conv --reps 5 > "$O/layers_variant_p$v.log" 2>&1
    echo "patch=$v rc=$?"; grep -E "conv[345]" "$O/layers_variant_p$v.log" | head -12
  done
fi
if [ $WGW = 1 ]; then
  echo "== wgw_kernel's fetch variant on conv2-5 wgrad: 1 staging loads first, 2 spread over the chunk =="
  for v in 1 2; do
    CONVNET_WG_TILE=$v timeout 120 python tools/layer_bench.py --only conv --reps 5 > "$O/layers_variant_w$v.log" 2>&1
    echo "wgrad_tile=$v rc=$?"; grep -E "conv[2345]" "$O/layers_variant_w$v.log" | head -12
  done
fi
echo "== the step: default, each new kernel alone, both =="
for v in "0 0" "$GPW 0" "0 $WGW" "$GPW $WGW"; do
  set -- $v
  [ -s "$O/bench_p$1_w$2.json" ] && continue
  CONVNET_GG_PATCH=$1 CONVNET_WG_TILE=$2 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ref-host --no-other-path --no-live-traffic > "$O/bench_p$1_w$2.json" 2> "$O/bench_p$1_w$2.err"
  echo "patch=$1 wgrad_tile=$2 rc=$?"; python -c "import json,sys; d=json.loads(open('$O/bench_p$1_w$2.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('one_stream_ms_per_step'))" 2>/dev/null
done
if [ $GPW = 3 ]; then
  echo "== counters, conv4 fprop / dgrad =="
  MODES="0 3" timeout 500 bash tools/pmc_conv.sh conv4 "$O/pmc_conv4" 2>&1 | tail -12
fi

#!/bin/bash
# Collects the round's judged artefacts on the GPU box in ONE gpurun call (copy the summaries into profiles/ afterwards):
#   bench line (roofline with LIVE PMC traffic + cpu_baseline + ref_host), rocprofv3 kernel-trace stats of the same command, the two PMC
#   traffic passes of the whole step (FETCH_SIZE and WRITE_SIZE separately — together they need 5 of the 4 TCC slots), matrix-pipe
#   counters of conv4, per-layer table, the per-GPU batch sweep of strong scaling, the split-path arithmetic table, DP self-tests.
# Usage: /usr/local/graft/bin/gpurun --timeout 1700 -- 'bash tools/profile_round.sh r06 r05'
TAG=${1:-rXX}
PREV=${2:-r05}   # the previous round's library for the same-call A/B: convnet_amd/lib/libconvnet_hip_$PREV.so
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/profile_$TAG
mkdir -p "$O"; cd "$R" || exit 1
Q="--no-cpu-baseline --no-ref-host --no-other-path --no-live-traffic"
echo "== bench (default flags: what the driver runs)"
timeout 500 python bench.py > "$O/bench_n1.json" 2> "$O/bench_n1.err"; echo "rc=$?"; cut -c1-230 "$O/bench_n1.json"
echo "== one-stream bench"
timeout 200 python bench.py --no-overlap-wgrad --no-side-stream-update $Q > "$O/bench_n1_one_stream.json" 2>/dev/null; cut -c60-160 "$O/bench_n1_one_stream.json"
echo "== same-call A/B against the previous round's library (convnet_amd/lib/libconvnet_hip_$PREV.so), where it is present"
if [ -f convnet_amd/lib/libconvnet_hip_$PREV.so ]; then
  for i in 1 2 3; do
    CONVNET_HIP_LIB=libconvnet_hip_$PREV.so timeout 200 python bench.py $Q > "$O/bench_n1_${PREV}lib_run$i.json" 2>/dev/null; echo "$PREV lib: $(cut -c60-175 "$O/bench_n1_${PREV}lib_run$i.json")"
    timeout 200 python bench.py $Q > "$O/bench_n1_now_run$i.json" 2>/dev/null; echo "now    : $(cut -c60-175 "$O/bench_n1_now_run$i.json")"
  done
  CONVNET_HIP_LIB=libconvnet_hip_$PREV.so timeout 120 python tools/layer_bench.py > "$O/layer_bench_${PREV}lib.txt" 2>&1
fi
echo "== this round's choices one at a time, same call: pooling masks off (CONVNET_POOL_MASK=0), response-norm fast kernels off (CONVNET_RNORM_FAST=0)"
for v in "CONVNET_POOL_MASK=0" "CONVNET_RNORM_FAST=0" "CONVNET_POOL_MASK=1"; do env $v timeout 200 python bench.py --steps 20 --warmup 5 $Q > "$O/bench_choice_$v.json" 2>/dev/null; echo "$v: $(cut -c60-175 "$O/bench_choice_$v.json")"; done
echo "== per-layer table"
timeout 120 python tools/layer_bench.py > "$O/layer_bench.txt" 2>&1; grep -v amdgpu "$O/layer_bench.txt" | head -60
timeout 120 python tools/pool_bench.py > "$O/pool_bench.txt" 2>&1
echo "== per-GPU batch sweep (strong scaling of a global batch of 256: 128 / 64 / 32 images per GPU), both matrix paths"
for b in 128 64 32; do
  timeout 200 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-ref-host --no-live-traffic > "$O/bench_b$b.json" 2>/dev/null; echo "b=$b $(cut -c60-170 "$O/bench_b$b.json")"
done
echo "== bench.py data-parallel paths on one rank (RCCL, 1-rank world): torch transport, C-ABI transport, strong-scaling flag"
timeout 200 python bench.py --force-exchange --steps 12 --warmup 4 $Q > "$O/bench_dp1_torch.json" 2> "$O/bench_dp1_torch.err"; echo "rc=$?"; cut -c60-200 "$O/bench_dp1_torch.json"
timeout 200 python bench.py --force-exchange --transport abi --steps 12 --warmup 4 $Q > "$O/bench_dp1_abi.json" 2> "$O/bench_dp1_abi.err"; echo "rc=$?"; cut -c60-200 "$O/bench_dp1_abi.json"
timeout 60 python bench.py --gpus 2 > "$O/bench_gpus2_on_one_gpu.txt" 2>&1; echo "--gpus 2 on this box: rc=$? $(tail -1 "$O/bench_gpus2_on_one_gpu.txt")"
echo "== other models"
timeout 200 python bench.py --model vgg --batch 128 --steps 5 --warmup 2 $Q > "$O/bench_vgg_bs128.json" 2>/dev/null; cut -c1-200 "$O/bench_vgg_bs128.json"
timeout 200 python bench.py --model alexnet_nin --steps 10 --warmup 3 $Q > "$O/bench_alexnet_nin.json" 2>/dev/null; cut -c1-200 "$O/bench_alexnet_nin.json"
echo "== split-path arithmetic vs float64 on the product kernels (the -m gpu test, table kept)"
timeout 300 python -m pytest tests/test_split_arithmetic_gpu.py -q -m gpu -s 2>&1 | grep -E "x 2\^-24|passed|failed" > "$O/split_arithmetic.txt"; tail -3 "$O/split_arithmetic.txt"
[ -x tools/split_gemm ] && timeout 60 ./tools/split_gemm > "$O/split_gemm.txt" 2>&1
echo "== rocprofv3 kernel trace + stats of the bench command"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d "$O/prof" -o b --output-format csv -- python "$R/bench.py" --steps 10 --warmup 3 $Q > "$O/bench_under_rocprof.json" 2> "$O/rocprof.err"; echo "rc=$?"
find "$O/prof" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$O/bench_kernel_stats.csv"; head -12 "$O/bench_kernel_stats.csv" | cut -c1-160
timeout 200 rocprofv3 --kernel-trace --stats -d "$O/prof1" -o b --output-format csv -- python "$R/bench.py" --steps 10 --warmup 3 --no-overlap-wgrad --no-side-stream-update $Q > /dev/null 2>&1
find "$O/prof1" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$O/bench_kernel_stats_one_stream.csv"
echo "== PMC traffic passes (whole step, every kernel)"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --kernel-trace --pmc $c -d "$O/pmc_$c" -o p --output-format csv -- python "$R/bench.py" --steps 3 --warmup 1 $Q --no-kernel-timers > "$O/pmc_$c.log" 2>&1
  echo "$c rc=$?"
  find "$O/pmc_$c" -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} "$O/pmc_${c}_counter_collection.csv"
done
echo "== matrix-pipe counters of conv4 (layer bench)"
timeout 150 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d "$O/pmc_conv4" -o p --output-format csv -- python "$R/tools/layer_bench.py" --only conv4 --reps 3 > "$O/pmc_conv4.log" 2>&1; echo "rc=$?"
find "$O/pmc_conv4" -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} "$O/pmc_conv4_layer.csv"; rm -rf "$O/pmc_conv4"
cd "$R" && python tools/pmc_traffic.py "$O/pmc_FETCH_SIZE_counter_collection.csv" "$O/pmc_WRITE_SIZE_counter_collection.csv" > "$O/pmc_traffic_bench.json" 2> "$O/pmc_traffic.err"; echo "traffic rc=$?"
rm -rf "$O/prof" "$O/prof1" "$O/pmc_FETCH_SIZE" "$O/pmc_WRITE_SIZE"
python - <<PY
import json
try:
    k = json.load(open("$O/pmc_traffic_bench.json"))["kernels"]
    for name in list(k)[:12]:
        print(f"{name[:64]:64s} {k[name]['traffic_bytes'] / 1e6:10.1f} MB/launch")
    d = json.loads(open("$O/bench_n1.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("bench:", d["value"], "img/s", d["ms_per_step"], "ms | dominant", r["kernel"], r["achieved"], "/", r["peak"], "=", r["frac"], "| traffic", r["traffic"], r.get("traffic_source", "")[:40])
except Exception as e:
    print("summary unavailable:", e)
PY

#!/bin/bash
# One gpurun call that answers every question round 1 left open (NOTES.md "To confirm on the first GPU run of round 2").
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/round2_first_run.sh'
# Everything is wrapped in its own timeout; results land in gpurun_out/r2_first/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_first
mkdir -p "$O"
cd "$R" || exit 1

echo "== 1. whole GPU suite (pooling XCD order, DAG test and rebuilt reference-host library together for the first time)"
timeout 200 python -m pytest tests -x -q -m gpu > "$O/suite.log" 2>&1; echo "rc=$?" >> "$O/suite.log"; tail -3 "$O/suite.log"

echo "== 2. dormant three-LDS-stage gg_kernel: correctness, then speed per layer against the default"
CONVNET_GG_STAGES3=1 timeout 120 python -m pytest tests/test_hip_parity.py tests/test_net_gpu.py -x -q -m gpu -k "conv or net or bprop or fprop" > "$O/stages3_tests.log" 2>&1
echo "rc=$?" >> "$O/stages3_tests.log"; tail -2 "$O/stages3_tests.log"
timeout 60 python tools/layer_bench.py > "$O/layer_default.log" 2>&1
CONVNET_GG_STAGES3=1 timeout 60 python tools/layer_bench.py > "$O/layer_stages3.log" 2>&1
paste <(grep gg_kernel "$O/layer_default.log") <(grep gg_kernel "$O/layer_stages3.log" | awk '{print $(NF-1), $NF}') | head -20

echo "== 2b. dormant wgrad fragment prefetch (CONVNET_WG_PREFETCH=1): correctness, then speed per layer"
CONVNET_WG_PREFETCH=1 timeout 120 python -m pytest tests/test_hip_parity.py tests/test_net_gpu.py -x -q -m gpu -k "conv or outp or net or bprop or fc" > "$O/wgpf_tests.log" 2>&1
echo "rc=$?" >> "$O/wgpf_tests.log"; tail -2 "$O/wgpf_tests.log"
CONVNET_WG_PREFETCH=1 timeout 60 python tools/layer_bench.py > "$O/layer_wgpf.log" 2>&1
paste <(grep wg_kernel "$O/layer_default.log") <(grep wg_kernel "$O/layer_wgpf.log" | awk '{print $(NF-1), $NF}') | head -20

echo "== 3. one-wave-per-SIMD micro-benchmark, three schedules"
for v in 0 1 2; do timeout 20 tools/wave1_gemm $v; done 2>&1 | tee "$O/wave1_gemm.log"

echo "== 4. bench line (default) and with the three-stage kernel"
timeout 120 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"; cut -c1-200 "$O/bench_default.json"
CONVNET_GG_STAGES3=1 timeout 120 python bench.py --no-cpu-baseline > "$O/bench_stages3.json" 2> "$O/bench_stages3.err"; cut -c1-200 "$O/bench_stages3.json"
CONVNET_WG_PREFETCH=1 timeout 120 python bench.py --no-cpu-baseline > "$O/bench_wgpf.json" 2> "$O/bench_wgpf.err"; cut -c1-200 "$O/bench_wgpf.json"
CONVNET_GG_STAGES3=1 CONVNET_WG_PREFETCH=1 timeout 120 python bench.py --no-cpu-baseline > "$O/bench_both.json" 2> "$O/bench_both.err"; cut -c1-200 "$O/bench_both.json"

echo "== 5. HBM-side traffic after the pooling XCD order (two PMC passes, never combined)"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 70 rocprofv3 --kernel-trace --pmc $c -d "$O/pmc_$c" -o p --output-format csv -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timers > "$O/pmc_$c.log" 2>&1
  echo "$c rc=$?"
done
cd "$R" && python tools/pmc_traffic.py "$O/pmc_FETCH_SIZE/p_counter_collection.csv" "$O/pmc_WRITE_SIZE/p_counter_collection.csv" > "$O/pmc_traffic_bench.json" 2>> "$O/pmc_FETCH_SIZE.log"
python - <<'EOF'
import json, os
p = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r2_first/pmc_traffic_bench.json")
try:
    k = json.load(open(p))["kernels"]
    for name in list(k)[:8]:
        print(f"{name[:60]:60s} {k[name]['traffic_bytes'] / 1e6:10.1f} MB/launch")
except Exception as e:   # noqa: BLE001
    print("traffic summary unavailable:", e)
EOF

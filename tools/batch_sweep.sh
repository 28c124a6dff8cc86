#!/bin/bash
# Per-GPU rate at the batch sizes strong scaling of global 256 implies (256/128/64/32 images per GPU), both matrix paths.
# Usage: gpurun --timeout 600 -- 'bash tools/batch_sweep.sh r03_base'
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/sweep_$TAG
mkdir -p "$O"; cd "$R" || exit 1
for b in 256 128 64 32; do
  timeout 200 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-ref-host --no-live-traffic > "$O/bench_b$b.json" 2> "$O/bench_b$b.err"; echo "b=$b rc=$?"
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_b$b.json").read().strip().splitlines()[-1])
    o = d.get("fp32_mfma_path") or {}
    print("batch $b: split %.0f img/s %.3f ms | fp32 %s img/s" % (d["value"], d["ms_per_step"], o.get("value")))
except Exception as e:
    print("parse failed", e)
PY
done

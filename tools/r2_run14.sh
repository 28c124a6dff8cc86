#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_run14
mkdir -p "$O"; cd "$R" || exit 1
timeout 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "fused_pool_undo or outp" 2>&1 | tail -3
timeout 200 python bench.py --no-cpu-baseline --no-ref-host > "$O/bench.json" 2> "$O/bench.err"; python - <<'PY'
import json,os
d=json.load(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r2_run14/bench.json"))); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel"], r["achieved"], "model_frac", r["model_frac"])
for k,v in list(r["ops"].items())[:14]: print(f"{v:8.4f} {k}")
PY
CONVNET_NO_POOL_FUSION=1 timeout 200 python bench.py --no-cpu-baseline --no-ref-host 2>/dev/null | cut -c1-200

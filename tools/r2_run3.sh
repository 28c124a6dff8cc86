#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r2_run3
mkdir -p "$O"; cd "$R" || exit 1
echo "== 1. whole GPU suite"
timeout 1500 python -m pytest tests -q -m gpu --durations=25 > "$O/suite.log" 2>&1; echo "rc=$?"; grep -E "passed|failed|FAILED|ERROR" "$O/suite.log" | tail -30; grep -A30 "slowest" "$O/suite.log" | head -32
echo "== 2. bench"
timeout 300 python bench.py > "$O/bench.json" 2> "$O/bench.err"; echo "rc=$?"; python - <<'PY'
import json,os
p=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r2_run3/bench.json")
try:
    d=json.load(open(p)); r=d["roofline"]
    print(d["value"], d["ms_per_step"], "dom", r["kernel"], r["achieved"], r["frac"], "exec", r["executed"], "model_frac", r["model_frac"])
    print("ref_host", d.get("ref_host")); print("cpu", d.get("cpu_baseline",{}).get("value"))
except Exception as e: print("bench parse failed", e); print(open(p).read()[:500])
PY
tail -5 "$O/bench.err"
echo "== 3. smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4

#!/bin/bash
# full GPU suite + bench at the default (bf16-split) matrix path, one gpurun call.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/split
mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu > $O/suite.log 2>&1; tail -12 $O/suite.log
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-250 $O/bench.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/split/bench.json'))
r=d['roofline']
print({k:r[k] for k in ('kernel','achieved','frac','model_frac','pipe','all_mfma_kernels')})
print(d.get('fp32_mfma_path'), d.get('ref_host'), d.get('cpu_baseline',{}).get('value'))
for k,v in list(r['families'].items())[:14]: print(k, v)
PY

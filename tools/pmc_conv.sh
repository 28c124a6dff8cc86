#!/bin/bash
# PMC passes over one conv layer of tools/layer_bench.py for the gather-GEMM kernels (patch modes, include/convnet_hip.h), same call.
# Usage (on the GPU box): [MODES="0 3"] bash tools/pmc_conv.sh conv4 gpurun_out/pmc_conv4      (default MODES: 0 2)
LAYER=${1:-conv4}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=${2:-$R/gpurun_out/pmc_$LAYER}
case $O in /*) ;; *) O=$R/$O;; esac
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  for v in ${MODES:-0 2}; do
    CONVNET_GG_PATCH=$v timeout 120 rocprofv3 --kernel-trace --pmc $set -d "$O/p${i}_v$v" -o p --output-format csv -- python "$R/tools/layer_bench.py" --only $LAYER --reps 3 > "$O/p${i}_v$v.log" 2>&1
    echo "set $i patch=$v rc=$?"
  done
done
python - "$O" <<'PY'
import csv, glob, sys, collections, os
O = sys.argv[1]
for d in sorted(glob.glob(O + "/p*_v*")):
    if not os.path.isdir(d): continue
    ctr = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gpw_kernel" in k or "gpp_kernel" in k or "ggp_kernel<2, 2, 2, 128" in k or "ggp_kernel<2,2,2,128" in k:
                name = "gpw" if "gpw_kernel" in k else "gpp" if "gpp_kernel" in k else "ggp"
                ctr[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gpw_kernel" in k or "gpp_kernel" in k or "ggp_kernel<2, 2, 2, 128" in k or "ggp_kernel<2,2,2,128" in k:
                dur["gpw" if "gpw_kernel" in k else "gpp" if "gpp_kernel" in k else "ggp"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for name in ctr:
        print(os.path.basename(d), name, {c: round(sum(v) / len(v)) for c, v in ctr[name].items()}, "dur_us %.1f" % (sum(dur[name]) / max(1, len(dur[name]))))
PY
find "$O" -name "*.csv" -size +2M -delete

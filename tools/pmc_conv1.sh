#!/bin/bash
# PMC passes over conv1 fprop of tools/layer_bench.py: gfc_kernel against ggp_kernel's generic-k path (CONVNET_GG_FEWC), same call.
# Usage (on the GPU box): bash tools/pmc_conv1.sh gpurun_out/pmc_conv1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=${1:-$R/gpurun_out/pmc_conv1}
case $O in /*) ;; *) O=$R/$O;; esac
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  for v in 1 0; do
    CONVNET_GG_FEWC=$v timeout 120 rocprofv3 --kernel-trace --pmc $set -d "$O/p${i}_v$v" -o p --output-format csv -- python "$R/tools/layer_bench.py" --only conv1 --reps 3 > "$O/p${i}_v$v.log" 2>&1
    echo "set $i fewc=$v rc=$?"
  done
done
python - "$O" <<'PY'
import csv, glob, sys, collections, os
O = sys.argv[1]
pick = lambda k: "gfc" if "gfc_kernel" in k else "ggp" if "ggp_kernel<1, 4, 3, 64" in k or "ggp_kernel<1,4,3,64" in k else None
for d in sorted(glob.glob(O + "/p*_v*")):
    if not os.path.isdir(d): continue
    ctr = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = pick(r["Kernel_Name"])
            if n: ctr[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = pick(r["Kernel_Name"])
            if n: dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for name in ctr:
        print(os.path.basename(d), name, {c: round(sum(v) / len(v)) for c, v in ctr[name].items()}, "dur_us %.1f" % (sum(dur[name]) / max(1, len(dur[name]))))
PY
find "$O" -name "*.csv" -size +2M -delete

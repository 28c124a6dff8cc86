#!/bin/bash
# PMC passes over one conv layer of tools/layer_bench.py, every GEMM-shaped kernel of the layer reported by name (forward / input-gradient /
# weight-gradient kernels of whatever kind the library picks), same call.  Separate --pmc passes combined with --kernel-trace only.
# Usage (on the GPU box): [LIB=libconvnet_hip_r05.so] bash tools/pmc_layer.sh conv2 gpurun_out/pmc_conv2
LAYER=${1:-conv2}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=${2:-$R/gpurun_out/pmc_$LAYER}
case $O in /*) ;; *) O=$R/$O;; esac
mkdir -p "$O"
[ -n "$LIB" ] && export CONVNET_HIP_LIB=$LIB
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set -d "$O/p$i" -o p --output-format csv -- python "$R/tools/layer_bench.py" --only $LAYER --reps 3 > "$O/p$i.log" 2>&1
  echo "set $i rc=$?"
done
python - "$O" <<'PY'
import csv, glob, sys, collections, os, re
O = sys.argv[1]
def short(k):
    k = k.replace("void ", "").replace("chip::", "")
    return re.sub(r"\(.*", "", k)
keep = ("gpw_kernel", "gpv_kernel", "gpp_kernel", "ggp_kernel", "gg_kernel", "wgw_kernel", "wg_kernel", "gfc_kernel")
for d in sorted(glob.glob(O + "/p*")):
    if not os.path.isdir(d): continue
    ctr = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k.startswith(keep): ctr[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k.startswith(keep): dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for name in ctr:
        print(os.path.basename(d), name, {c: round(sum(v) / len(v)) for c, v in ctr[name].items()}, "dur_us %.1f" % (sum(dur[name]) / max(1, len(dur[name]))))
PY
find "$O" -name "*.csv" -size +2M -delete; find "$O" -name "*.db" -delete

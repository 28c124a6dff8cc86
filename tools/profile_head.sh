#!/bin/bash
# The bench line and the rocprofv3 kernel statistics of the same command on the library of the current commit (a short form of
# tools/profile_round.sh for a last check after small changes).  Usage: bash tools/gpu.sh 900 'bash tools/profile_head.sh r05_head'
TAG=${1:-head}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"; cd "$R" || exit 1
Q="--no-cpu-baseline --no-ref-host --no-other-path --no-live-traffic"
timeout 400 python bench.py > "$O/bench_n1.json" 2> "$O/bench_n1.err"; echo "bench rc=$?"; cut -c60-175 "$O/bench_n1.json"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d "$O/prof" -o b --output-format csv -- python "$R/bench.py" --steps 10 --warmup 3 $Q > "$O/bench_under_rocprof.json" 2> "$O/rocprof.err"; echo "rocprof rc=$?"
find "$O/prof" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$O/bench_kernel_stats.csv"; rm -rf "$O/prof"
head -14 "$O/bench_kernel_stats.csv" | cut -c1-150

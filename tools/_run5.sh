cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_split_arithmetic_gpu.py tests/test_data_parallel_gpu.py "tests/test_full_geometry_gpu.py::test_vgg_224_training_pass_vs_cpu_oracle" "tests/test_full_geometry_gpu.py::test_vgg_first_conv_layers_at_full_size_n128" "tests/test_full_geometry_gpu.py::test_vgg_pool1_2x2_stride2_at_full_size_n128" -q -m gpu -s --durations=8 2>&1 | grep -v "^$" | grep -E "x 2\^-24|passed|failed|Error|assert|oracle forward|^[0-9.]+s |FAILED" | head -90
mkdir -p gpurun_out/dp1
F="--steps 12 --warmup 4 --no-cpu-baseline --no-ref-host --no-other-path"
for v in "plain:" "torch:--force-exchange" "abi:--force-exchange --transport abi"; do
  n=${v%%:*}; a=${v#*:}
  timeout 200 python bench.py $F $a > gpurun_out/dp1/$n.json 2> gpurun_out/dp1/$n.err; echo "$n rc=$? $(cut -c1-160 gpurun_out/dp1/$n.json)"
done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/dp1/prof -o b --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-ref-host --no-other-path --force-exchange --no-kernel-timers > /dev/null 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/dp1/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $GRAFT_REPO_ROOT/gpurun_out/dp1/kernel_stats_force_exchange.csv
grep -i -E "nccl|rccl|Name" $GRAFT_REPO_ROOT/gpurun_out/dp1/kernel_stats_force_exchange.csv | cut -c1-200 | head
rm -rf $GRAFT_REPO_ROOT/gpurun_out/dp1/prof

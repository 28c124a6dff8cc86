cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab6
F="--steps 20 --warmup 5 --no-cpu-baseline --no-ref-host --no-other-path"
for b in 256 32; do for o in before after; do
  CONVNET_WGRAD_ORDER=$o timeout 200 python bench.py $F --batch $b > gpurun_out/ab6/${o}_b$b.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab6/${o}_b$b.json").read().strip().splitlines()[-1])
print("$o b=$b", d["value"], d["ms_per_step"])
PY
done; done
for v in "torch:--force-exchange" "abi:--force-exchange --transport abi"; do
  n=${v%%:*}; a=${v#*:}
  timeout 200 python bench.py $F $a > gpurun_out/ab6/dp_$n.json 2>/dev/null; echo "$n $(cut -c1-140 gpurun_out/ab6/dp_$n.json)"
done
timeout 900 python -m pytest tests/test_split_arithmetic_gpu.py tests/test_data_parallel_gpu.py tests/test_net_gpu.py "tests/test_full_geometry_gpu.py::test_vgg_224_training_pass_vs_cpu_oracle" "tests/test_full_geometry_gpu.py::test_vgg_first_conv_layers_at_full_size_n128" "tests/test_full_geometry_gpu.py::test_vgg_pool1_2x2_stride2_at_full_size_n128" -q -m gpu -x 2>&1 | tail -15

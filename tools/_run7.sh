cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab7
F="--steps 16 --warmup 4 --no-cpu-baseline --no-ref-host --no-other-path"
GPU_MAX_HW_QUEUES=8 timeout 200 python bench.py $F --force-exchange > gpurun_out/ab7/dp_q8.json 2>/dev/null; echo "dp q8 $(cut -c60-150 gpurun_out/ab7/dp_q8.json)"
GPU_MAX_HW_QUEUES=2 timeout 200 python bench.py $F > gpurun_out/ab7/plain_q2.json 2>/dev/null; echo "plain q2 $(cut -c60-150 gpurun_out/ab7/plain_q2.json)"
GPU_MAX_HW_QUEUES=8 timeout 200 python bench.py $F > gpurun_out/ab7/plain_q8.json 2>/dev/null; echo "plain q8 $(cut -c60-150 gpurun_out/ab7/plain_q8.json)"
timeout 200 python bench.py $F --force-exchange --no-overlap-wgrad > gpurun_out/ab7/dp_nowg.json 2>/dev/null; echo "dp no-overlap-wgrad $(cut -c60-150 gpurun_out/ab7/dp_nowg.json)"
timeout 200 python bench.py $F --no-overlap-wgrad > gpurun_out/ab7/plain_nowg.json 2>/dev/null; echo "plain no-overlap-wgrad $(cut -c60-150 gpurun_out/ab7/plain_nowg.json)"
timeout 900 python -m pytest tests/test_net_gpu.py "tests/test_full_geometry_gpu.py::test_vgg_224_training_pass_vs_cpu_oracle" "tests/test_full_geometry_gpu.py::test_vgg_first_conv_layers_at_full_size_n128" "tests/test_full_geometry_gpu.py::test_vgg_pool1_2x2_stride2_at_full_size_n128" -q -m gpu 2>&1 | tail -8

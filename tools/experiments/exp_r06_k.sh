#!/bin/bash
# (run on the trees that still had the LDS-tile kernel behind VPT_CONV_FIRST_LDS_TILE=1 -- commit 53cb17f and the working tree after it; the switch and
# that kernel were removed once the A/B was recorded: profiles/r06_experiments.md section 4)
# Round 6, call K: vpt_conv_first_kernel without the conv tile in LDS (nine window positions per lane) -- parity, then A/B against the LDS-tile kernel.
mkdir -p gpurun_out/r06k
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp16_kernels.py -x -q -p no:cacheprovider -k "conv_first or pack" > gpurun_out/r06k/t_kernels.log 2>&1; tail -3 gpurun_out/r06k/t_kernels.log
timeout 900 python -m pytest tests/test_gpu_training.py -x -q -p no:cacheprovider -k "conv_first" > gpurun_out/r06k/t_training.log 2>&1; tail -3 gpurun_out/r06k/t_training.log
for i in 1 2; do
  timeout 300 python tools/conv_first_bench.py 1024 2>&1 | tail -1 | sed 's/^/new: /' | tee -a gpurun_out/r06k/ab.log
  VPT_CONV_FIRST_LDS_TILE=1 timeout 300 python tools/conv_first_bench.py 1024 2>&1 | tail -1 | sed 's/^/old: /' | tee -a gpurun_out/r06k/ab.log
done
timeout 1200 python -m pytest tests/test_gpu_policy.py tests/test_gpu_configs.py tests/test_gpu_dropin.py -x -q -p no:cacheprovider > gpurun_out/r06k/t_policy.log 2>&1; tail -3 gpurun_out/r06k/t_policy.log

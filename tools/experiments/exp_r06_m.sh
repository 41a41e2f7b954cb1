#!/bin/bash
# Round 6, call M: vpt_conv_first_kernel with 4-wave workgroups (both half-tiles in every wave, 4 workgroups per CU) against 8-wave ones.
mkdir -p gpurun_out/r06m
W4=$PWD/video-pre-training_amd/build/libvpt_cf_w4.so
VPT_HIP_LIB=$W4 timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -p no:cacheprovider -k "conv_first" 2>&1 | tail -2
for i in 1 2; do
  timeout 300 python tools/conv_first_bench.py 1024 2>&1 | tail -1 | sed 's/^/8 waves: /' | tee -a gpurun_out/r06m/ab.log
  VPT_HIP_LIB=$W4 timeout 300 python tools/conv_first_bench.py 1024 2>&1 | tail -1 | sed 's/^/4 waves: /' | tee -a gpurun_out/r06m/ab.log
done

#!/bin/bash
# round 4, run R: conv_first forward with per-channel sums on the conflict-free pool-read map vs the previous map (build/libvpt_ref.so)
cd "$(dirname "$0")/.."; out=gpurun_out/r04_r; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp16_kernels.py -m gpu -x -q -k "conv_first or n_folded or pool" 2>&1 | tail -2
for r in 1 2; do
  timeout 120 python tools/conv_first_bench.py 1024 2>&1 | grep "^conv_first" | tee -a $out/cf.log
  VPT_HIP_LIB=$PWD/video-pre-training_amd/build/libvpt_ref.so timeout 120 python tools/conv_first_bench.py 1024 2>&1 | grep "^conv_first" | tee -a $out/cf.log
done

#!/bin/bash
# round 4, run S: residual prefetch requested in the last-but-one step (behind its DMA, never waited for by a barrier) vs in the step before (reslate0)
cd "$(dirname "$0")/.."; out=gpurun_out/r04_s; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp16_kernels.py tests/test_gpu_training.py -m gpu -x -q -k "conv3x3 or n_folded or dgrad or conv_layer or backward" 2>&1 | tail -2
for r in 1 2 3; do
  for v in "" reslate0; do
    lib=""; [ -n "$v" ] && lib="$PWD/video-pre-training_amd/build/libvpt_$v.so"
    echo "== ${v:-new} round $r"
    VPT_HIP_LIB=$lib VPT_BENCH_POOL=0 timeout 300 python tools/conv_bench.py 2>&1 | grep "^\[bf16\]" | grep "block" | tee -a $out/cb.log
  done
done

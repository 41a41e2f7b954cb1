#!/bin/bash
# round 4, run Q: phase ablations of the scatter-free first-conv backward (wrong results, timing only)
cd "$(dirname "$0")/.."; out=gpurun_out/r04_q; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_training.py -m gpu -x -q -k "conv_first" 2>&1 | tail -2
for v in "" cfb1 cfb16 ref; do
  lib=""; [ -n "$v" ] && lib="$PWD/video-pre-training_amd/build/libvpt_$v.so"
  VPT_HIP_LIB=$lib timeout 120 python tools/conv_first_bwd_bench.py 1024 5 2>&1 | grep "^conv_first" | tee -a $out/cfb.log
done

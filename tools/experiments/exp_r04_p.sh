#!/bin/bash
# round 4, run P: scatter-free first-conv backward vs the round-3 kernel (build/libvpt_ref.so), same box
cd "$(dirname "$0")/.."; out=gpurun_out/r04_p; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_training.py -m gpu -x -q -k "conv_first" > $out/t1.log 2>&1; echo "t1 rc=$?"; tail -5 $out/t1.log | cut -c1-300
for r in 1 2; do
  timeout 120 python tools/conv_first_bwd_bench.py 1024 5 2>&1 | grep "^conv_first" | tee -a $out/cfb.log
  VPT_HIP_LIB=$PWD/video-pre-training_amd/build/libvpt_ref.so timeout 120 python tools/conv_first_bwd_bench.py 1024 5 2>&1 | grep "^conv_first" | tee -a $out/cfb.log
done

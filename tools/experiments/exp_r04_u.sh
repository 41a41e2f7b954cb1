#!/bin/bash
# round 4, run U: residual without a main-loop prefetch (all four subtiles requested at the start of the epilogue) vs shipped; correct results both
cd "$(dirname "$0")/.."; out=gpurun_out/r04_u; mkdir -p $out
VPT_HIP_LIB=$PWD/video-pre-training_amd/build/libvpt_nopf.so timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv3x3" 2>&1 | tail -1
for r in 1 2 3; do
  for v in "" nopf; do
    lib=""; [ -n "$v" ] && lib="$PWD/video-pre-training_amd/build/libvpt_$v.so"
    echo "== ${v:-shipped} round $r"
    VPT_HIP_LIB=$lib VPT_BENCH_POOL=0 timeout 300 python tools/conv_bench.py 2>&1 | grep "^\[bf16\]" | grep block | tee -a $out/cb.log
  done
done

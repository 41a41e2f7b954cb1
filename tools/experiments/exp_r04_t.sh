#!/bin/bash
# round 4, run T: what the residual costs a K = 1152 tile besides its loads (timing only; ablated builds give wrong results)
cd "$(dirname "$0")/.."; out=gpurun_out/r04_t; mkdir -p $out
export VPT_BENCH_SHAPES="s0.res,64,128,128,1;s0.nores,64,128,128,0"
for r in 1 2; do
  for v in "" epi1 epi3 epi7; do
    lib=""; [ -n "$v" ] && lib="$PWD/video-pre-training_amd/build/libvpt_$v.so"
    echo "== ${v:-shipped} round $r"
    VPT_HIP_LIB=$lib VPT_BENCH_POOL=0 timeout 300 python tools/conv_bench.py 2>&1 | grep "^\[bf16\]" | tee -a $out/cb.log
  done
done

#!/bin/bash
# halo-staging ablation on the real conv kernel (timing only): 1 = no zero-fill selects, 2 = no halo loads / LDS writes after block 0
out=$PWD/gpurun_out/r03_n; mkdir -p $out
B=$PWD/video-pre-training_amd/build
for r in 1 2; do
  for v in base halo1 halo2; do
    if [ $v = base ]; then unset VPT_HIP_LIB; else export VPT_HIP_LIB=$B/libvpt_$v.so; fi
    echo "== $v round $r"
    timeout 300 python tools/conv_bench.py 2>&1 | grep -E "TF/s" | cut -c1-110 | tee -a $out/conv_$v.log
  done
done

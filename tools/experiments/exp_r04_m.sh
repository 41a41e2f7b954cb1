#!/bin/bash
# round 4, run M: start-up de-phasing of the two co-resident workgroups of a CU (16-row tiles), same box
cd "$(dirname "$0")/.."; out=gpurun_out/r04_m; mkdir -p $out
for r in 1 2; do
  for v in base dph2 dph4 dph6; do
    echo "== $v round $r"
    lib=""; [ $v != base ] && lib="$PWD/video-pre-training_amd/build/libvpt_$v.so"
    VPT_HIP_LIB=$lib VPT_BENCH_TILING=throughput16 VPT_BENCH_POOL=0 timeout 300 python tools/conv_bench.py 2>&1 | grep "^\[" | tee $out/cb_${v}_$r.log
  done
done

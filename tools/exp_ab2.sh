#!/bin/bash
# A/B of library variants inside one call: exp_ab2.sh <tag> name1 name2 ... ("new" = working tree), two interleaved rounds
tag=$1; shift
out=gpurun_out/exp_$tag; mkdir -p $out
B=$PWD/video-pre-training_amd/build
run() { t=$1; shift; env "$@" timeout 300 python tools/conv_bench.py 512 5 > $out/$t.log 2>&1; echo "== $t"; cat $out/$t.log | grep -v "Warn\|amdgpu.ids"; }
for r in 1 2; do
  for n in "$@"; do
    if [ $n = new ]; then run new_$r; else run ${n}_$r VPT_HIP_LIB=$B/libvpt_$n.so; fi
  done
done

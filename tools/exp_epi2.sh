#!/bin/bash
out=gpurun_out/exp_epi2; mkdir -p $out
run() { tag=$1; shift; env "$@" timeout 300 python tools/conv_bench.py 512 5 > $out/$tag.log 2>&1; echo "== $tag"; cat $out/$tag.log | grep -v "Warn\|amdgpu.ids"; }
run base
for k in 1 2 3; do run abl$k VPT_HIP_LIB=$PWD/video-pre-training_amd/build/libvpt_abl$k.so; done
run nostats VPT_BENCH_NOSTATS=1
for k in 1 3; do VPT_HIP_LIB=$PWD/video-pre-training_amd/build/libvpt_abl$k.so python tools/conv_trace.py 2>&1 | grep -v amdgpu.ids > $out/trace_s0_abl$k.log; sed -n 1,5p $out/trace_s0_abl$k.log; grep "per-CU" $out/trace_s0_abl$k.log; done

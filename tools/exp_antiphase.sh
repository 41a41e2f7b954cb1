#!/bin/bash
out=gpurun_out/exp_ap; mkdir -p $out
run() { tag=$1; shift; env "$@" timeout 300 python tools/conv_bench.py 512 5 > $out/$tag.log 2>&1; echo "== $tag"; cat $out/$tag.log | grep -v "Warn\|amdgpu.ids"; }
run ap0 VPT_CONV_ANTIPHASE_US=0
for us in 6 10 14 18 24 30; do run ap$us VPT_CONV_ANTIPHASE_US=$us; done
run ap0_b VPT_CONV_ANTIPHASE_US=0
VPT_CONV_ANTIPHASE_US=14 python tools/conv_trace.py 2>&1 | grep -v amdgpu.ids > $out/trace_s0.log; head -24 $out/trace_s0.log
VPT_CONV_ANTIPHASE_US=24 python tools/conv_trace.py 32 256 256 1 2048 2>&1 | grep -v amdgpu.ids > $out/trace_s1.log; head -24 $out/trace_s1.log

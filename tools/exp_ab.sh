#!/bin/bash
# A/B inside one call: reference library (tools/build_ref_lib.sh) vs the working tree, interleaved
out=gpurun_out/exp_ab; mkdir -p $out
REF=$PWD/video-pre-training_amd/build/libvpt_ref.so
run() { tag=$1; shift; env "$@" timeout 300 python tools/conv_bench.py 512 5 > $out/$tag.log 2>&1; echo "== $tag"; cat $out/$tag.log | grep -v "Warn\|amdgpu.ids"; }
run ref_1 VPT_HIP_LIB=$REF
run new_1
run ref_2 VPT_HIP_LIB=$REF
run new_2

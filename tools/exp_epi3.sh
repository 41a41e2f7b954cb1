#!/bin/bash
out=gpurun_out/exp_epi3; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp16_kernels.py tests/test_gpu_training.py -q -x -k "conv3x3 or dgrad or conv_layer or bc_gradients" > $out/test.log 2>&1; echo "test rc=$?" >> $out/test.log; tail -3 $out/test.log
run() { tag=$1; shift; env "$@" timeout 300 python tools/conv_bench.py 512 5 > $out/$tag.log 2>&1; echo "== $tag"; cat $out/$tag.log | grep -v "Warn\|amdgpu.ids"; }
run base
run abl1 VPT_HIP_LIB=$PWD/video-pre-training_amd/build/libvpt_abl1.so
run base_b

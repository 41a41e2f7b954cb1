#!/bin/bash
out=gpurun_out/exp_epi; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp16_kernels.py tests/test_gpu_training.py -q -x -k "conv3x3 or dgrad or conv_layer or bc_gradients" > $out/test.log 2>&1; echo "test rc=$?" >> $out/test.log; tail -3 $out/test.log
run() { tag=$1; shift; env "$@" timeout 300 python tools/conv_bench.py 512 5 > $out/$tag.log 2>&1; echo "== $tag"; cat $out/$tag.log | grep -v "Warn\|amdgpu.ids"; }
run base
run ap10 VPT_CONV_ANTIPHASE_US=10
run ap20 VPT_CONV_ANTIPHASE_US=20
run base_b
python tools/conv_trace.py 2>&1 | grep -v amdgpu.ids > $out/trace_s0.log; head -24 $out/trace_s0.log
VPT_CONV_ANTIPHASE_US=14 python tools/conv_trace.py 2>&1 | grep -v amdgpu.ids > $out/trace_s0_ap.log; head -24 $out/trace_s0_ap.log
VPT_CONV_ANTIPHASE_US=24 python tools/conv_trace.py 32 256 256 1 2048 2>&1 | grep -v amdgpu.ids > $out/trace_s1.log; head -24 $out/trace_s1.log

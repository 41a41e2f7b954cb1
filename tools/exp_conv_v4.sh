#!/bin/bash
out=gpurun_out/exp_v4; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "conv3x3" -x > $out/test_plain.log 2>&1; echo "test rc=$?" >> $out/test_plain.log; tail -3 $out/test_plain.log
VPT_CONV_COUNTED=1 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_training.py -q -k "conv3x3 or dgrad" -x > $out/test_counted.log 2>&1; echo "test rc=$?" >> $out/test_counted.log; tail -3 $out/test_counted.log
run() { tag=$1; shift; env "$@" timeout 300 python tools/conv_bench.py 512 5 > $out/$tag.log 2>&1; echo "== $tag"; cat $out/$tag.log | grep -v "Warn\|amdgpu.ids"; }
run base
run counted VPT_CONV_COUNTED=1
run counted_prio VPT_CONV_COUNTED=1 VPT_CONV_PRIO=1
run one_wg VPT_CONV_EXTRA_LDS=8192 VPT_CONV_COUNTED=1
run noepi VPT_CONV_ABLATE=1 VPT_CONV_COUNTED=1
run one_wg_noepi VPT_CONV_EXTRA_LDS=8192 VPT_CONV_ABLATE=1 VPT_CONV_COUNTED=1
run nomain VPT_CONV_ABLATE=2
run zero VPT_BENCH_ZERO=1 VPT_CONV_COUNTED=1
run base_b

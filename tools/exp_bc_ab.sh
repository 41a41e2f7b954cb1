#!/bin/bash
# BC step A/B inside one call: reference library vs working tree (tools/build_ref_lib.sh), kernel tests first
out=gpurun_out/exp_bc_ab; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_kernels.py -q -x -k "${VPT_AB_TESTS:-wgrad or conv_layer or bc_gradients or conv_backward or conv_first}" > $out/test.log 2>&1; echo "test rc=$?" >> $out/test.log; grep "PARITY conv_first\|passed\|failed\|Error" $out/test.log | tail -6
REF=$PWD/video-pre-training_amd/build/libvpt_ref.so
for r in 1 2; do
  VPT_HIP_LIB=$REF timeout 300 python tools/bc_bench.py --steps 3 2>&1 | grep -v amdgpu.ids > $out/ref_$r.log; echo "== ref_$r"; grep "BC step\|${VPT_AB_GREP:-wgrad\|dgrad\|3x3_forward\|conv_first}" $out/ref_$r.log
  timeout 300 python tools/bc_bench.py --steps 3 2>&1 | grep -v amdgpu.ids > $out/new_$r.log; echo "== new_$r"; grep "BC step\|${VPT_AB_GREP:-wgrad\|dgrad\|3x3_forward\|conv_first}" $out/new_$r.log
done

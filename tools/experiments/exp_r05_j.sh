#!/bin/bash
# round 5, call j: whole-line stores in vpt_conv_bwd_prep_pooled_kernel (one lane exchange per row) vs the 3-wave kernel of call d (the 4-wave variant of call i measured slower: reverted)
# SGPRs: 141 / 160 -> 122 / 128 VGPRs) vs the library before it; training kernel tests first
out=gpurun_out/r05_j; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_training.py -q -m gpu -s -k "prepare_pooled or conv_layer or gated or bc_gradients_vs_oracle or chunking" > $out/test.log 2>&1; echo "test rc=$?" >> $out/test.log
grep -E "PARITY pooled|passed|failed|^FAILED|Error|rc=" $out/test.log | cut -c1-260 | tail -14
REF=$PWD/video-pre-training_amd/build/libvpt_ref.so
for r in 1 2; do
  VPT_HIP_LIB=$REF timeout 300 python tools/bc_bench.py --steps 3 --streams1 2>&1 | grep -v amdgpu.ids > $out/ref_$r.log; echo "== ref_$r"; grep -E "^BC step|backward_prepare  |work/call  3.758e\+09|work/call  9.395e\+08" $out/ref_$r.log
  timeout 300 python tools/bc_bench.py --steps 3 --streams1 2>&1 | grep -v amdgpu.ids > $out/new_$r.log; echo "== new_$r"; grep -E "^BC step|backward_prepare  |work/call  3.758e\+09|work/call  9.395e\+08" $out/new_$r.log
done

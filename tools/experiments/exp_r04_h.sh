#!/bin/bash
# round 4, call H: GEMM epilogue variants (parity + speed), attn_bwd without spills / bank conflicts, BC step; A/B vs the previous commit
out=gpurun_out/r04_h; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_training.py tests/test_gpu_fp16_kernels.py -q -x -k "linear or attention_backward or bc_gradients_vs_oracle or heads_logprob" > $out/t1.log 2>&1; echo "t1 rc=$?"; grep -E "passed|failed|Error|assert" $out/t1.log | cut -c1-300 | tail -8
REF=$PWD/video-pre-training_amd/build/libvpt_ref.so
for r in 1 2; do
  for n in new ref; do
    if [ $n = new ]; then L=""; else L="VPT_HIP_LIB=$REF"; fi
    env $L timeout 300 python tools/gemm_bench.py 8192,6304,2048 8192,2048,2048 8192,8192,2048 8192,2048,8192 8192,8764,2048 > $out/gemm_${n}_$r.log 2>&1; echo "== gemm $n $r"; grep "M=" $out/gemm_${n}_$r.log
  done
done
for r in 1 2; do
  for n in new ref; do
    if [ $n = new ]; then L=""; else L="VPT_HIP_LIB=$REF"; fi
    env $L timeout 600 python tools/bc_bench.py --steps 3 > $out/bc_${n}_$r.log 2>&1; echo "== bc $n $r"; grep -E "BC step|attention|linear|kernel time" $out/bc_${n}_$r.log
  done
done
timeout 600 python tools/latency_bench.py --steps 300 > $out/latency.log 2>&1; grep -E "wrapper" $out/latency.log

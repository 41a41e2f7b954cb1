#!/bin/bash
# round 5, call h: statistics of the residual modes (1, 5) from the stored pairs by v_dot2 (fewer hazard nops), dot2 statistics in the pool-fused phase 2: tests + A/B vs the library before
# inference epilogue (mode 4): kernel / policy / fold tests, then conv_bench and the forward bench against the library before (build/libvpt_ref.so)
out=gpurun_out/r05_h; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp16_kernels.py tests/test_gpu_policy.py tests/test_gpu_idm.py -q -m gpu -x > $out/test.log 2>&1; echo "test rc=$?" >> $out/test.log
grep -E "passed|failed|^FAILED|Error|rc=" $out/test.log | cut -c1-300 | tail -8
REF=$PWD/video-pre-training_amd/build/libvpt_ref.so
for r in 1 2; do
  echo "== ref conv_bench $r"; VPT_HIP_LIB=$REF timeout 300 python tools/conv_bench.py 512 2>&1 | grep -E "median|fused"
  echo "== new conv_bench $r"; timeout 300 python tools/conv_bench.py 512 2>&1 | grep -E "median|fused"
done
for r in 1 2 3; do
  echo "== ref forward $r"; VPT_HIP_LIB=$REF timeout 300 python bench.py --steps 8 --warmup 2 --bc-steps 0 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['by_mode'])"
  echo "== new forward $r"; timeout 300 python bench.py --steps 8 --warmup 2 --bc-steps 0 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['by_mode'])"
done

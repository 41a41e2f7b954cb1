#!/bin/bash
# round 4, call D: residual in the accumulators' layout (8-byte loads, no lane exchanges) vs the round-3 path
out=gpurun_out/r04_d; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp16_kernels.py tests/test_gpu_training.py -q -x -k "conv3x3 or conv_layer or conv_prepare or bc_gradients_vs_oracle" > $out/t1.log 2>&1; echo "t1 rc=$?"; grep -E "passed|failed|Error|assert" $out/t1.log | cut -c1-300 | tail -8
B=$PWD/video-pre-training_amd/build
export VPT_BENCH_POOL=0
for r in 1 2 3; do
  for n in new resold; do
    if [ $n = new ]; then timeout 300 python tools/conv_bench.py 512 5 > $out/cb_${n}_$r.log 2>&1; else VPT_HIP_LIB=$B/libvpt_$n.so timeout 300 python tools/conv_bench.py 512 5 > $out/cb_${n}_$r.log 2>&1; fi
    echo "== $n round $r"; grep -v "Warn\|amdgpu.ids" $out/cb_${n}_$r.log | cut -c1-120
  done
done
unset VPT_BENCH_POOL
for r in 1 2; do
  for n in new resold; do
    if [ $n = new ]; then L=""; else L="VPT_HIP_LIB=$B/libvpt_$n.so"; fi
    env $L timeout 300 python bench.py --steps 6 --warmup 2 --bc-steps 0 --no-cpu-baseline > $out/bench_${n}_$r.json 2> $out/bench_${n}_$r.err
    python - <<PY
import json
try:
    d=json.loads(open("$out/bench_${n}_$r.json").read().strip().splitlines()[-1])
    print("$n round $r:", d["value"], "frames/s", d["ms_per_step"], "ms; roofline", d["roofline"]["frac"], d["roofline"].get("by_mode"))
except Exception as e:
    print("bench $n failed", e); print(open("$out/bench_${n}_$r.err").read()[-1500:])
PY
  done
done

#!/bin/bash
# round 4, call A: new sampling path + auto graph, pool-fused conv (parity + A/B), autograd overflow guard
out=gpurun_out/r04_a; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sampling.py tests/test_native_abi.py -q -x -s -k "conv3x3 or sampling or philox or uniforms or draw or act_stochastic or abi or maxpool" > $out/t1.log 2>&1; echo "t1 rc=$?"; grep -E "passed|failed|Error|SAMPLING|assert" $out/t1.log | cut -c1-300 | tail -15
timeout 900 python -m pytest tests/test_gpu_policy.py tests/test_gpu_training.py -q -s -k "step_graph or act_uses or overflow or chunks_vs_golden or chunking" > $out/t2.log 2>&1; echo "t2 rc=$?"; grep -E "passed|failed|Error|assert" $out/t2.log | cut -c1-300 | tail -15
timeout 600 python -m pytest tests/test_gpu_dropin.py -q -s -k "minerl" > $out/t3.log 2>&1; echo "t3 rc=$?"; grep -E "passed|failed|Error|DROP-IN|assert" $out/t3.log | cut -c1-300 | tail -8
timeout 300 python tools/conv_bench.py 512 5 > $out/conv_bench.log 2>&1; grep -v "Warn\|amdgpu.ids" $out/conv_bench.log
for r in 1 2; do
  for fp in 0 1; do
    VPT_FUSE_POOL=$fp timeout 300 python bench.py --steps 6 --warmup 2 --bc-steps 0 --no-cpu-baseline > $out/bench_fp${fp}_$r.json 2> $out/bench_fp${fp}_$r.err
    python - <<PY
import json
try:
    d=json.loads(open("$out/bench_fp${fp}_$r.json").read().strip().splitlines()[-1])
    k=d["kernels"]
    print("fuse_pool=$fp round $r:", d["value"], "frames/s", d["ms_per_step"], "ms; roofline", d["roofline"]["frac"], "; conv", k["vpt_conv3x3_forward"]["ms"], "pool", k.get("vpt_maxpool_forward",{}).get("ms"), "affine", k["vpt_frame_affine_forward"]["ms"])
except Exception as e:
    print("bench fp=$fp failed", e); print(open("$out/bench_fp${fp}_$r.err").read()[-1500:])
PY
  done
done

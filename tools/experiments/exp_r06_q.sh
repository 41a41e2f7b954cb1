#!/bin/bash
# Round 6, call Q: the final tree -- all GPU tests, smoke, and the concurrency soak (3 processes) over the whole paths with the rewritten first-conv kernel
# in them, 2x width, both formats.
out=gpurun_out/r06q; mkdir -p $out
timeout 1800 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $out/gpu_tests.log 2>&1; tail -2 $out/gpu_tests.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $out/smoke.log
timeout 900 python tools/kernel_stress.py 3 4000 STRESS_PRECISION=bf16 STRESS_WIDTH=2x > $out/soak_bf16_2x.log 2>&1
timeout 900 python tools/kernel_stress.py 3 4000 STRESS_PRECISION=fp16 STRESS_WIDTH=1x > $out/soak_fp16_1x.log 2>&1
grep -h "mismatching" $out/soak_*.log | grep -vc " 0 mismatching"; grep -hc "mismatching" $out/soak_*.log

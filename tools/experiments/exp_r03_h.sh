#!/bin/bash
# 8-wave latency conv: kernel + policy tests, then the T = 1 step in both precisions
out=$PWD/gpurun_out/r03_h; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp16_kernels.py tests/test_gpu_policy.py -q > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $out/tests.log | cut -c1-300 | tail -6
timeout 300 python tools/latency_bench.py --steps 300 2>&1 | grep -E "eager|graph|conv3x3|total" | tee $out/latency_bf16.log
VPT_PRECISION=fp16 timeout 300 python tools/latency_bench.py --steps 300 2>&1 | grep -E "eager|graph|conv3x3|total" | tee $out/latency_fp16.log

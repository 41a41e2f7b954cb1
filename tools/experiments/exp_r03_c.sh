#!/bin/bash
out=gpurun_out/r03_c; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp16_kernels.py tests/test_gpu_policy.py -q > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $out/tests.log | cut -c1-300 | tail
( timeout 120 tools/ubench/wino_skeleton | tail -4; timeout 120 tools/ubench/wino_skeleton 1 | tail -4 ) > $out/ladder.log 2>&1; cat $out/ladder.log
timeout 300 python tools/latency_bench.py --steps 200 2>&1 | grep -v Warning | head -40 | tee $out/latency_bf16.log
VPT_PRECISION=fp16 timeout 300 python tools/latency_bench.py --steps 200 2>&1 | grep -E "eager|graph" | tee $out/latency_fp16.log

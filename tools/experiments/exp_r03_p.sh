#!/bin/bash
# acting step: in-place mask, one-launch act() tail, cached parameter list
out=$PWD/gpurun_out/r03_p; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_policy.py tests/test_gpu_dropin.py tests/test_gpu_fp16_kernels.py -q -x > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=|Error" $out/tests.log | cut -c1-300 | tail -8
timeout 300 python tools/latency_bench.py --steps 300 2>&1 | grep -v Warn | head -34 | cut -c1-160 | tee $out/latency_bf16.log
VPT_PRECISION=fp16 timeout 300 python tools/latency_bench.py --steps 300 2>&1 | grep -E "eager|graph|replay" | tee $out/latency_fp16.log

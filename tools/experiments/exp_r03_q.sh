#!/bin/bash
out=$PWD/gpurun_out/r03_q; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_policy.py tests/test_gpu_dropin.py "tests/test_gpu_kernels.py::test_conv_first_pool" "tests/test_gpu_training.py::test_conv_first_backward" -q > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=|Error" $out/tests.log | cut -c1-300 | tail -8
timeout 300 python tools/latency_bench.py --steps 300 2>&1 | grep -E "eager|graph|replay:" | tee $out/latency_bf16.log

#!/bin/bash
# one GPU visit: the fp16 / parity tests, then the whole GPU suite
out=gpurun_out/r02_tests; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_fp16_kernels.py tests/test_gpu_policy.py -x -q -s -m gpu > $out/parity.log 2>&1; echo "rc=$?" >> $out/parity.log
grep -E "PARITY|ACTIONS|passed|failed|Error|error|rc=" $out/parity.log | cut -c1-400 | tail -60

#!/bin/bash
out=$PWD/gpurun_out/r03_t; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_policy.py tests/test_gpu_idm.py -q > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=|Error|assert" $out/tests.log | cut -c1-400 | tail -12

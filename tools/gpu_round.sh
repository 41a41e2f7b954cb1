#!/bin/bash
out=gpurun_out/r02_tests; mkdir -p $out
timeout 2400 python -m pytest tests/test_gpu_training.py tests/test_gpu_distributed.py -x -q -s -m gpu --durations=5 > $out/training.log 2>&1; echo "rc=$?" >> $out/training.log
grep -E "PARITY|passed|failed|Error|error|rc=|s call" $out/training.log | cut -c1-330 | tail -40

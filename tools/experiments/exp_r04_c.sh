#!/bin/bash
out=gpurun_out/r04_c; mkdir -p $out
tools/ubench/permlane > $out/permlane.log 2>&1; cat $out/permlane.log
timeout 600 python -m pytest tests/test_gpu_sampling.py -q -s > $out/t1.log 2>&1; echo "t1 rc=$?"; grep -E "passed|failed|Error|SAMPLING|assert" $out/t1.log | cut -c1-300 | tail -12

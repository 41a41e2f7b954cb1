#!/bin/bash
out=gpurun_out/r02_tests; mkdir -p $out
timeout 2400 python -m pytest tests/test_gpu_configs.py tests/test_gpu_idm.py -x -q -s -m gpu --durations=10 > $out/configs.log 2>&1; echo "rc=$?" >> $out/configs.log
grep -E "PARITY|ACTIONS|passed|failed|Error|error|rc=|s call" $out/configs.log | cut -c1-330 | tail -50
timeout 600 python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -6 $out/smoke.log | cut -c1-300

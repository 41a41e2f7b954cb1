#!/bin/bash
out=gpurun_out/r02_full; mkdir -p $out
timeout 3000 python -m pytest tests/ -x -q -m gpu --durations=8 > $out/gpu_tests.log 2>&1; echo "rc=$?" >> $out/gpu_tests.log
grep -E "passed|failed|Error|rc=|s call|assert" $out/gpu_tests.log | cut -c1-300 | tail -24
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -c 6000 $out/bench.json; tail -5 $out/bench.err

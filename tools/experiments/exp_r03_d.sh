#!/bin/bash
out=gpurun_out/r03_d; mkdir -p $out
timeout 2400 python -m pytest tests/ -q -m gpu --durations=8 > $out/gpu_tests.log 2>&1; echo "rc=$?" >> $out/gpu_tests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $out/gpu_tests.log | cut -c1-400 | tail -20
timeout 300 python tools/latency_bench.py --steps 300 2>&1 | grep -v Warning | head -30 | tee $out/latency_bf16.log
VPT_PRECISION=fp16 timeout 300 python tools/latency_bench.py --steps 300 2>&1 | grep -E "eager|graph" | tee $out/latency_fp16.log

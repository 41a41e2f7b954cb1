#!/bin/bash
# Round 6, call J: the concurrency soak in the OTHER format and width (fp16 library = a separate binary; 2x = the benchmark's width).
# 3 processes each; every output compared bit for bit with the process's first result.
mkdir -p gpurun_out/r06j
timeout 900 python tools/kernel_stress.py 3 6000 STRESS_PRECISION=fp16 STRESS_WIDTH=1x > gpurun_out/r06j/soak_fp16_1x.log 2>&1
timeout 900 python tools/kernel_stress.py 3 4000 STRESS_PRECISION=bf16 STRESS_WIDTH=2x > gpurun_out/r06j/soak_bf16_2x.log 2>&1
timeout 900 python tools/kernel_stress.py 3 4000 STRESS_PRECISION=fp16 STRESS_WIDTH=2x > gpurun_out/r06j/soak_fp16_2x.log 2>&1
grep -h "===\|mismatching\|skipped" gpurun_out/r06j/*.log | awk '{print}' | tail -80

#!/bin/bash
# round 4, call G: full bench line (configs block, cpu bc baseline), competitive-heads test, latency through the wrapper, changed suites
out=gpurun_out/r04_g; mkdir -p $out
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -c 6000 $out/bench.json; tail -5 $out/bench.err
timeout 600 python tools/latency_bench.py --steps 300 > $out/latency.log 2>&1; grep -E "eager|graph|wrapper|replay" $out/latency.log
VPT_PRECISION=fp16 timeout 600 python tools/latency_bench.py --steps 300 > $out/latency_fp16.log 2>&1; grep -E "eager|graph|wrapper|replay" $out/latency_fp16.log
timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_dropin.py tests/test_gpu_distributed.py -q -s -k "competitive or prior_dominated or minerl or idm_agent or two_rank" > $out/t1.log 2>&1; echo "t1 rc=$?"; grep -E "passed|failed|Error|ACTIONS|DROP-IN|assert" $out/t1.log | cut -c1-420 | tail -24

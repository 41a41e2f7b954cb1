#!/bin/bash
out=$PWD/gpurun_out/r03_f; mkdir -p $out
timeout 300 python tools/gemv_bench.py 2>&1 | grep -v Warn | tee $out/gemv_bench_graph.log
timeout 300 python tools/latency_bench.py --steps 300 2>&1 | grep -E "eager|graph" | tee $out/latency_bf16.log

#!/bin/bash
out=gpurun_out/r03_b; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_training.py tests/test_gpu_distributed.py -q -s > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $out/tests.log | cut -c1-300 | tail
( timeout 120 tools/ubench/wino_skeleton; timeout 120 tools/ubench/wino_skeleton 1 ) > $out/wino_skeleton.log 2>&1; cat $out/wino_skeleton.log
for s in 1 3 3 1; do VPT_BC_STREAMS=$s timeout 300 python tools/bc_bench.py --steps 4 2>&1 | grep -E "BC step|instrumented" | sed "s/^/streams=$s /"; done | tee $out/bc_streams.log
VPT_PRECISION=bf16 timeout 200 python tools/conv_bench.py 512 5 2>&1 | grep TF > $out/conv_bf16.log; VPT_PRECISION=fp16 timeout 200 python tools/conv_bench.py 512 5 2>&1 | grep TF > $out/conv_fp16.log; cat $out/conv_bf16.log $out/conv_fp16.log

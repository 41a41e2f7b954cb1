#!/bin/bash
out=$PWD/gpurun_out/r03_i; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_kernels.py -q -k "conv_first or bc_ or chunking or checkpoint" -s > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=|PARITY conv_first" $out/tests.log | cut -c1-300 | tail -14
for s in 1 2; do timeout 300 python tools/bc_bench.py --steps 4 2>&1 | grep -E "BC step|conv_first" ; done | tee $out/bc.log

#!/bin/bash
# round 4, run N: the 256 x 256 / eight-wave LDS-DMA GEMM vs the 256 x 128 kernel
cd "$(dirname "$0")/.."; out=gpurun_out/r04_n; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "linear" > $out/t1.log 2>&1; echo "t1 rc=$?"; tail -5 $out/t1.log | cut -c1-300
for r in 1 2; do
  echo "== round $r"
  timeout 300 python tools/gemm_bench.py 8192,6304,2048 8192,2048,2048 8192,8192,2048 8192,2048,8192 8192,8764,2048 2>&1 | grep "^M=" | tee $out/gemm_$r.log
done

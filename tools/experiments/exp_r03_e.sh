#!/bin/bash
out=$PWD/gpurun_out/r03_e; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp16_kernels.py tests/test_gpu_policy.py tests/test_gpu_idm.py tests/test_gpu_distributed.py "tests/test_gpu_training.py::test_bc_gradients_independent_of_cnn_chunking" -q > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $out/tests.log | cut -c1-300 | tail
timeout 300 python tools/latency_bench.py --steps 300 2>&1 | grep -v Warning | head -30 | tee $out/latency_bf16.log
VPT_PRECISION=fp16 timeout 300 python tools/latency_bench.py --steps 300 2>&1 | grep -v Warning | head -14 | tee $out/latency_fp16.log
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t1prof -- python $GRAFT_REPO_ROOT/tools/latency_bench.py --steps 200 > $out/t1prof.log 2>&1
f=$(find $out/t1prof -name "*kernel_stats.csv" | head -1); head -30 "$f" | cut -c1-160

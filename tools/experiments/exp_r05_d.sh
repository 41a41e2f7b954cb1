#!/bin/bash
# round 5, call d: pool-fused TRAINING forward with arg-max masks (vpt_conv3x3_kernel mode 7 + seam masks) and its backward from the pooled
# tensors (vpt_conv_bwd_prep_pooled_kernel): kernel tests, BC gradient tests, BC step A/B (VPT_BC_FUSED_POOL = 0 / 1) in one call
out=gpurun_out/r05_d; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_training.py -q -m gpu -s \
  -k "pool_argmax or prepare_pooled or pool_fused or bc_gradients or bc_step or chunking or reference_bc_loop or fused_pool" > $out/test.log 2>&1; echo "test rc=$?" >> $out/test.log
grep -E "PARITY pooled|PARITY BC gradients|mean rel-L2|passed|failed|^FAILED|Error|rc=" $out/test.log | cut -c1-330 | tail -30
for r in 1 2; do
  VPT_BC_FUSED_POOL=0 timeout 300 python tools/bc_bench.py --steps 3 --streams1 2>&1 | grep -v amdgpu.ids > $out/old_$r.log; echo "== conv->pool_$r"; grep -B1 -A14 "kernel time" $out/old_$r.log | grep -v "^--"
  timeout 300 python tools/bc_bench.py --steps 3 --streams1 2>&1 | grep -v amdgpu.ids > $out/new_$r.log; echo "== fused_$r"; grep -B1 -A14 "kernel time" $out/new_$r.log | grep -v "^--"
done
grep "peak mem" $out/old_1.log $out/new_1.log
grep "vpt_conv_backward_prepare " $out/new_1.log | head -12
timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_distributed.py -q -m gpu -x -k "config5 or bc_3x or pre_lstm or distributed or shard" > $out/test2.log 2>&1; echo "test2 rc=$?" >> $out/test2.log
grep -E "passed|failed|^FAILED|rc=" $out/test2.log | cut -c1-300 | tail -6

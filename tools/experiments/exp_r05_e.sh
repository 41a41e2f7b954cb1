#!/bin/bash
# round 5, call e: GroupNorm-n backward apply folded into the pooled prepare (stacks 1, 2); is the fp16 gradient-vs-oracle figure of the fused-pool
# path noise (same test under VPT_BC_FUSED_POOL = 0 / 1, twice)?; BC A/B fold on / off; ingest leg (steady state, deeper raw-frame pipeline)
out=gpurun_out/r05_e; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_training.py -q -m gpu -s -k "prepare_pooled or chunking or bc_step" > $out/test.log 2>&1; echo "test rc=$?" >> $out/test.log
grep -E "PARITY pooled|PARITY BC gradients|passed|failed|^FAILED|Error|rc=" $out/test.log | cut -c1-330 | tail -24
for r in 1 2; do for fp in 0 1; do
  echo "== bc_gradients_vs_oracle fused_pool=$fp run $r"; VPT_BC_FUSED_POOL=$fp timeout 300 python -m pytest tests/test_gpu_training.py -q -m gpu -s -k "bc_gradients_vs_oracle and True" 2>&1 | grep -E "mean rel-L2|passed|failed" | cut -c1-220
done; done
for r in 1 2; do
  VPT_BC_FOLD_N_BWD=0 timeout 300 python tools/bc_bench.py --steps 3 --streams1 2>&1 | grep -v amdgpu.ids > $out/old_$r.log; echo "== two-pass_$r"; grep -E "^BC step|affine_backward|backward_prepare  " $out/old_$r.log
  timeout 300 python tools/bc_bench.py --steps 3 --streams1 2>&1 | grep -v amdgpu.ids > $out/new_$r.log; echo "== folded_$r"; grep -E "^BC step|affine_backward|backward_prepare  " $out/new_$r.log
done
timeout 300 python bench.py --steps 4 --warmup 2 --ingest-only 2>/dev/null | tail -1

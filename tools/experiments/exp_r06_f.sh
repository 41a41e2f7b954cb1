#!/bin/bash
# round 6, call F: stack 0's n backward inside the first conv's backward kernel -- tests, then the BC step A/B in one call
O=gpurun_out/r06_f; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_training.py -x -q -p no:cacheprovider -s -k "conv_first or bitwise or vs_oracle or reference_bc_loop or chunking" > $O/tests.log 2>&1; echo "tests rc $?"; grep "folded in\|passed\|failed" $O/tests.log | cut -c1-200
for rep in 1 2; do
  for v in 0 1; do
    VPT_BC_FOLD_N_BWD0=$v timeout 600 python tools/bc_bench.py --steps 6 > $O/bc_fold${v}_$rep.log 2>&1
    echo "fold_n_bwd0=$v rep $rep: $(grep -E "^BC step" $O/bc_fold${v}_$rep.log | head -1 | tr '\n' ' ')"
  done
done
grep -E "vpt_frame_affine_backward|vpt_conv_first_backward" $O/bc_fold0_2.log $O/bc_fold1_2.log | head

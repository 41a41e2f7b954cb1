#!/bin/bash
# round 6, call B: the deterministic (slab) reductions against the per-kernel tests; then the 2-rank diagnosis with bitwise comparisons
O=gpurun_out/r06_b; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_training.py -x -q -p no:cacheprovider > $O/test_training.log 2>&1; echo "training tests rc $?"; tail -n 3 $O/test_training.log
for m in "pga 6 40" "sync 4 40" "pgraw 4 40" "nopg 2 40"; do
  set -- $m
  timeout 700 python tools/diag_r06.py $1 $2 $3 > $O/$1.log 2>&1
  echo "== $1 rc $?"; grep -c "tensors differ" $O/$1.log; grep "calls differ" $O/$1.log | sort | uniq -c
done
timeout 300 python tools/diag_r06.py poisonlds 1 2 > $O/poisonlds.log 2>&1; grep "poison(" $O/poisonlds.log | cut -c1-300

#!/bin/bash
# round 6, call E: after the fix (vpt_backward.hip without SLP vectorisation) -- stress, trainer diagnosis, the new tests; and: does a busy neighbour
# INSIDE the process trigger the old build's fault too?
O=gpurun_out/r06_e; mkdir -p $O
export TMPDIR=/tmp
B=/root/repo/video-pre-training_amd/build
timeout 900 python tools/kernel_stress.py 3 2000 > $O/stress_fixed_3proc.log 2>&1
echo "== fixed build, 3 processes: pairs with mismatches: $(grep -h 'mismatching' $O/stress_fixed_3proc.log | grep -vc ' 0 mismatching')"; grep -h "skipped\|mismatching" $O/stress_fixed_3proc.log | grep -v " 0 mismatching" | head
timeout 600 python tools/kernel_stress.py 1 4000 VPT_HIP_LIB=$B/libvpt_lnb_nopk.so STRESS_SIDE_STREAM=1 STRESS_PATHS=0 > $O/stress_old_sidestream_1proc.log 2>&1
echo "== OLD build, 1 process + busy side stream:"; grep -h 'mismatching' $O/stress_old_sidestream_1proc.log | grep -v ' 0 mismatching'
timeout 600 python tools/kernel_stress.py 2 3000 VPT_HIP_LIB=$B/libvpt_lnb_nopk.so STRESS_PATHS=0 > $O/stress_old_2proc.log 2>&1
echo "== OLD build, 2 processes (positive control):"; grep -h 'mismatching' $O/stress_old_2proc.log | grep -v ' 0 mismatching'
for m in "sync 3 60" "pga 3 60"; do
  set -- $m
  timeout 700 python tools/diag_r06.py $1 $2 $3 > $O/$1.log 2>&1
  echo "== diag $1 rc $?: events $(grep -c 'tensors differ' $O/$1.log)"; grep "calls differ" $O/$1.log | sort | uniq -c
done
timeout 1500 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_concurrency.py tests/test_gpu_training.py tests/test_gpu_policy.py -x -q -p no:cacheprovider -s > $O/tests.log 2>&1; echo "tests rc $?"; grep "PARITY 2-rank\|PARITY concurrency\|passed\|failed" $O/tests.log | cut -c1-330

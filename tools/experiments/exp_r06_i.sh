#!/bin/bash
# round 6, call I: the long soak of the final tree -- every path beside two other processes, and the two-rank trainer comparisons, ~10x the suite's counts
O=gpurun_out/r06_i; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python tools/kernel_stress.py 3 20000 > $O/stress_3proc_20000.log 2>&1
echo "== stress 3 x 20000: launches $(grep -h mismatching $O/stress_3proc_20000.log | awk '{s+=$NF==\"launches\"?$(NF-1):0} END {print s}'), wrong $(grep -h mismatching $O/stress_3proc_20000.log | awk '{s+=$(NF-5)} END {print s}')"
timeout 1500 python tools/kernel_stress.py 4 8000 > $O/stress_4proc_8000.log 2>&1
echo "== stress 4 x 8000: wrong $(grep -h mismatching $O/stress_4proc_8000.log | awk '{s+=$(NF-5)} END {print s}')"
for m in "sync 3 300" "pga 3 300"; do
  set -- $m
  timeout 1200 python tools/diag_r06.py $1 $2 $3 > $O/$1.log 2>&1
  echo "== diag $1: events $(grep -c 'tensors differ' $O/$1.log)"; grep "calls differ" $O/$1.log | sort | uniq -c
done

#!/bin/bash
# round 6, call D: the one-row shift of vpt_ln_bwd_kernel's dx under a second process -- which build of the kernel shows it?
O=gpurun_out/r06_d; mkdir -p $O
export TMPDIR=/tmp
B=/root/repo/video-pre-training_amd/build
for v in default lnb_nopk lnb_noslp lnb_O1; do
  if [ $v = default ]; then e=""; else e="VPT_HIP_LIB=$B/libvpt_$v.so"; fi
  for rep in 1 2; do
    timeout 600 python tools/kernel_stress.py 3 3000 $e > $O/stress_${v}_$rep.log 2>&1
  done
  echo "== $v: $(grep -h 'mismatching' $O/stress_${v}_*.log | grep -v ' 0 mismatching' | wc -l) kernel/rank pairs with mismatches; total $(grep -h 'mismatching' $O/stress_${v}_*.log | awk '{s+=$(NF-5)} END {print s}')"
  grep -h 'mismatching' $O/stress_${v}_*.log | grep -v ' 0 mismatching' | sed 's/^ *rank . //' | sort | uniq -c
done

#!/bin/bash
# round 6, call C: single kernels under a second process (tools/kernel_stress.py)
O=gpurun_out/r06_c; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/kernel_stress.py 1 1500 > $O/stress_1proc.log 2>&1; grep -c "mismatching" $O/stress_1proc.log
timeout 900 python tools/kernel_stress.py 2 3000 > $O/stress_2proc.log 2>&1
timeout 900 python tools/kernel_stress.py 2 3000 AMD_SERIALIZE_KERNEL=3 > $O/stress_2proc_serialize.log 2>&1
timeout 900 python tools/kernel_stress.py 2 3000 HSA_ENABLE_SDMA=0 > $O/stress_2proc_nosdma.log 2>&1
timeout 900 python tools/kernel_stress.py 3 2000 > $O/stress_3proc.log 2>&1
grep -h "mismatching\|rank .: .* s$\|===" $O/stress_*.log | grep -v " 0 mismatching"
grep -h -A3 "elements differ" $O/stress_2proc.log | cut -c1-400 | head -40

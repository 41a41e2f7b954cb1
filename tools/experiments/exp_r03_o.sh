#!/bin/bash
out=$PWD/gpurun_out/r03_o; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
( cd $root && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python tools/latency_bench.py --steps 300 > $out/trace.log 2>&1 )
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python $root/tools/t1_timeline.py $f 100 | tee $out/timeline.txt
grep -E "eager|graph" $out/trace.log
rm -rf $out/trace

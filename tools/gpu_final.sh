#!/bin/bash
# End-of-round evidence in one call: all GPU tests, smoke, bench, (optionally) profiles.   tools/gpu_final.sh [profile]
bash tools/gpu_round.sh r03_final
if [ "$1" = "profile" ]; then bash tools/profile_round.sh r03 > gpurun_out/r03_final/profile_round.log 2>&1; tail -3 gpurun_out/r03_final/profile_round.log; fi
out=$PWD/gpurun_out/r03_final
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t1prof -- python $GRAFT_REPO_ROOT/tools/latency_bench.py --steps 200 > $out/t1prof.log 2>&1
grep -E "eager|graph" $out/t1prof.log
find $out/t1prof -name "*kernel_trace.csv" -delete

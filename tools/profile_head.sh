#!/bin/bash
# the parts of tools/profile_round.sh that depend on the launch structure, re-run at the round's last commit: single-stream kernel stats (both formats) + conv traffic
R=${1:-r03}
out=$PWD/gpurun_out/prof_$R; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py"
VPT_CNN_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/fwd1 -- $B --steps 3 --warmup 1 --bc-steps 0 --no-cpu-baseline > $out/fwd1_bench.json 2> $out/fwd1.err
VPT_CNN_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/fwd1_fp16 -- $B --precision fp16 --steps 3 --warmup 1 --bc-steps 0 --no-cpu-baseline > $out/fwd1_fp16_bench.json 2> $out/fwd1_fp16.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B --steps 1 --warmup 0 --bc-steps 0 --no-cpu-baseline > $out/pmc_fetch.json 2> $out/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $B --steps 1 --warmup 0 --bc-steps 0 --no-cpu-baseline > $out/pmc_write.json 2> $out/pmc_write.err
cd $GRAFT_REPO_ROOT
find $out -name "*kernel_trace.csv" -size +4M -delete
du -sh $out

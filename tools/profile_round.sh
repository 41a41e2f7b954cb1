#!/bin/bash
# rocprofv3 evidence for the round (run on the GPU box): kernel-trace stats of the bench, PMC passes of the conv kernel.
R=${1:-r02}
out=$PWD/gpurun_out/prof_$R; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
# (1) forward only, single CNN stream: per-launch durations of vpt_conv3x3_kernel comparable with bench.py's HIP events
VPT_CNN_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/fwd1 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --bc-steps 0 --no-cpu-baseline > $out/fwd1_bench.json 2> $out/fwd1.err
# (2) default forward (3 streams) + BC steps
rocprofv3 --kernel-trace --stats --output-format csv -d $out/fwdbc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --bc-steps 2 --bc-warmup 1 --no-cpu-baseline > $out/fwdbc_bench.json 2> $out/fwdbc.err
if [ -z "$VPT_PROF_SKIP_PMC" ]; then   # (VPT_PROF_SKIP_PMC=1: kernel-trace passes only, when vpt_conv3x3_kernel itself did not change)
# (3) PMC passes on the conv micro-benchmark (separate passes: SQ counters, GRBM, FETCH, WRITE)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/pmc_sq -- python $GRAFT_REPO_ROOT/tools/conv_bench.py 256 2 > $out/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $out/pmc_grbm -- python $GRAFT_REPO_ROOT/tools/conv_bench.py 256 2 > $out/pmc_grbm.log 2>&1
# (4) HBM traffic of the conv kernel over one bench step
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --bc-steps 0 --no-cpu-baseline > $out/pmc_fetch.json 2> $out/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --bc-steps 0 --no-cpu-baseline > $out/pmc_write.json 2> $out/pmc_write.err
fi
cd $GRAFT_REPO_ROOT
find $out -name "*.csv" | head -40
du -sh $out

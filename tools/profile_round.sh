#!/bin/bash
# rocprofv3 evidence for the round (run on the GPU box): kernel-trace stats of the bench in both operand formats, PMC passes of the
# conv kernel.  tools/profile_round.sh [R=r03]; tools/profile_summarize.py turns the raw output into the files under profiles/.
R=${1:-r05}
out=$PWD/gpurun_out/prof_$R; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py"
# (1) forward only, single CNN stream, per format: per-launch durations of vpt_conv3x3_kernel comparable with bench.py's HIP events
VPT_CNN_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/fwd1 -- $B --steps 3 --warmup 1 --bc-steps 0 --no-cpu-baseline > $out/fwd1_bench.json 2> $out/fwd1.err
VPT_CNN_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/fwd1_fp16 -- $B --precision fp16 --steps 3 --warmup 1 --bc-steps 0 --no-cpu-baseline > $out/fwd1_fp16_bench.json 2> $out/fwd1_fp16.err
# (2) default forward (3 streams) + BC steps
rocprofv3 --kernel-trace --stats --output-format csv -d $out/fwdbc -- $B --steps 2 --warmup 1 --bc-steps 2 --bc-warmup 1 --no-cpu-baseline > $out/fwdbc_bench.json 2> $out/fwdbc.err
if [ -z "$VPT_PROF_SKIP_PMC" ]; then
# (3) PMC passes on the conv micro-benchmark, both formats (separate passes: SQ counters, GRBM)
for p in bf16 fp16; do
VPT_BENCH_POOL=0 VPT_PRECISION=$p rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/pmc_sq_$p -- python $GRAFT_REPO_ROOT/tools/conv_bench.py 256 2 > $out/pmc_sq_$p.log 2>&1
VPT_BENCH_POOL=0 VPT_PRECISION=$p rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $out/pmc_grbm_$p -- python $GRAFT_REPO_ROOT/tools/conv_bench.py 256 2 > $out/pmc_grbm_$p.log 2>&1
done
if [ -z "$VPT_PROF_SKIP_T32" ]; then
# (3b) the same SQ / GRBM passes on the 32-row / eight-wave tile variant (tiling 3), bf16
VPT_BENCH_TILING=throughput32 VPT_BENCH_POOL=0 VPT_PRECISION=bf16 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/pmc_sq_bf16_t32 -- python $GRAFT_REPO_ROOT/tools/conv_bench.py 256 2 > $out/pmc_sq_bf16_t32.log 2>&1
VPT_BENCH_TILING=throughput32 VPT_BENCH_POOL=0 VPT_PRECISION=bf16 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $out/pmc_grbm_bf16_t32 -- python $GRAFT_REPO_ROOT/tools/conv_bench.py 256 2 > $out/pmc_grbm_bf16_t32.log 2>&1
fi
# (4) HBM traffic of the conv kernel over one bench step (FETCH_SIZE x2 on gfx950, MI355X_MICROARCH.md)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B --steps 1 --warmup 0 --bc-steps 0 --no-cpu-baseline > $out/pmc_fetch.json 2> $out/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $B --steps 1 --warmup 0 --bc-steps 0 --no-cpu-baseline > $out/pmc_write.json 2> $out/pmc_write.err
# ... and in the parity mode (fp16 operands: same bytes by construction; measured, not assumed)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_fp16 -- $B --precision fp16 --steps 1 --warmup 0 --bc-steps 0 --no-cpu-baseline > $out/pmc_fetch_fp16.json 2> $out/pmc_fetch_fp16.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_fp16 -- $B --precision fp16 --steps 1 --warmup 0 --bc-steps 0 --no-cpu-baseline > $out/pmc_write_fp16.json 2> $out/pmc_write_fp16.err
fi
# (5) the BC step on ONE stream (per-kernel durations without cross-stream overlap: the table DESIGN.md section 5 quotes), round 5
VPT_BC_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bc1 -- python $GRAFT_REPO_ROOT/tools/bc_bench.py --steps 2 --streams1 > $out/bc1.log 2> $out/bc1.err
cd $GRAFT_REPO_ROOT
# (6) acting-step latency (auto-captured graph, fresh state per step since round 5)
timeout 300 python tools/latency_bench.py > $out/latency.log 2>&1
# the merged-back output is capped at 64 MiB: keep the stats and counter tables, drop the per-dispatch kernel traces of the long runs
find $out -name "*kernel_trace.csv" -size +4M -delete
find $out -name "*.csv" | head -40
du -sh $out

#!/bin/bash
# PMC survey of every kernel of a (small) BC step: two SQ passes + FETCH_SIZE + WRITE_SIZE, summed per kernel name.   tools/pmc_bc_step.sh [tag]
tag=${1:-r03_pmc_bc}
out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export VPT_BC_STREAMS=1
P1="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $out/pmc$i -- python $GRAFT_REPO_ROOT/tools/bc_bench.py --batch 16 --steps 1 > $out/pmc$i.log 2>&1
done
python $GRAFT_REPO_ROOT/tools/pmc_summarize.py $out > $out/summary.txt
cat $out/summary.txt
find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete

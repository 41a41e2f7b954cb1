#!/bin/bash
# PMC survey (two SQ passes + FETCH_SIZE + WRITE_SIZE) of every kernel of an arbitrary python command, summed per kernel name.
#   tools/pmc_cmd.sh <tag> <python args...>      e.g.  tools/pmc_cmd.sh r03_pmc_t1 tools/latency_bench.py --steps 100
tag=$1; shift
out=$PWD/gpurun_out/$tag; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES"
i=0
for P in "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd $root && timeout 400 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $out/pmc$i -- python "$@" > $out/pmc$i.log 2>&1 )
done
python $root/tools/pmc_summarize.py $out > $out/summary.txt
cut -c1-330 $out/summary.txt | head -${PMC_HEAD:-30}
find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete

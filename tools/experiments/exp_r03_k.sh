#!/bin/bash
# PMC counters of vpt_conv_first_kernel (two passes of eight SQ counters), summarised per launch
out=$PWD/gpurun_out/r03_k; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
P1="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_BUSY_CYCLES"
P3="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS_ATOMIC SQ_WAVES"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $out/pmc$i -- python $GRAFT_REPO_ROOT/tools/conv_first_bench.py 1024 > $out/pmc$i.log 2>&1
done
python - <<P
import csv, glob, collections
for d in sorted(glob.glob("$out/pmc*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:40]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in acc.items():
            if "conv_first" not in k: continue
            print(k, {n: round(sum(v) / len(v)) for n, v in c.items()}, "launches", len(next(iter(c.values()))))
P

#!/bin/bash
# conv_first with aligned 8-byte input records: tests, micro-benchmark vs build/libvpt_ref.so, PMC of the new kernel
out=$PWD/gpurun_out/r03_m; mkdir -p $out
REF=$PWD/video-pre-training_amd/build/libvpt_ref.so
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp16_kernels.py tests/test_gpu_training.py -q -x -k "conv_first or pack" > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $out/tests.log | cut -c1-300 | tail -6
for lib in ref new; do
  if [ $lib = ref ]; then export VPT_HIP_LIB=$REF; else unset VPT_HIP_LIB; fi
  echo "== $lib"
  timeout 300 python tools/conv_first_bench.py 1024 2>&1 | grep conv_first | tee $out/conv_first_$lib.log
done
unset VPT_HIP_LIB
cd /tmp && export TMPDIR=/tmp
P1="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_BUSY_CYCLES"
i=0
for P in "$P1" "$P2"; do
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

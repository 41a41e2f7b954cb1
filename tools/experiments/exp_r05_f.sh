#!/bin/bash
# round 5, call f: vpt_gemm_tn_kernel with the pinned fragment schedule (linear wgrad) vs the library before it (build/libvpt_ref.so = HEAD), the
# mask test against oracle/pool_mask.py, and the full default bench line (ingest leg in its own process)
out=gpurun_out/r05_f; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_kernels.py tests/test_gpu_fp16_kernels.py -q -m gpu -k "linear or pool_argmax or wgrad" > $out/test.log 2>&1; echo "test rc=$?" >> $out/test.log
grep -E "passed|failed|^FAILED|Error|rc=" $out/test.log | cut -c1-300 | tail -8
REF=$PWD/video-pre-training_amd/build/libvpt_ref.so
for r in 1 2; do
  VPT_HIP_LIB=$REF timeout 300 python tools/bc_bench.py --steps 3 --streams1 2>&1 | grep -v amdgpu.ids > $out/ref_$r.log; echo "== ref_$r"; grep -E "^BC step|linear_wgrad|linear_forward" $out/ref_$r.log
  timeout 300 python tools/bc_bench.py --steps 3 --streams1 2>&1 | grep -v amdgpu.ids > $out/new_$r.log; echo "== new_$r"; grep -E "^BC step|linear_wgrad|linear_forward" $out/new_$r.log
done
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; python - <<P
import json
d = json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["bc_step"]["ms_per_step"], d["bc_step"]["kernels_ms"])
print(json.dumps(d["ingest"])[:1500])
P
tail -3 $out/bench.err

#!/bin/bash
# round 4, run W: relative-position bias of the attention as an MFMA table (T = R . b_nd) vs the per-score 10-term sums (build/libvpt_ref.so)
cd "$(dirname "$0")/.."; out=gpurun_out/r04_w; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_policy.py tests/test_gpu_fp16_kernels.py -m gpu -x -q -k "attention or golden or chunk or ragged or batch_around" 2>&1 | tail -2
for r in 1 2; do
  for v in "" ref; do
    lib=""; [ -n "$v" ] && lib="$PWD/video-pre-training_amd/build/libvpt_$v.so"
    VPT_HIP_LIB=$lib timeout 600 python bench.py --steps 6 --warmup 2 --bc-steps 0 --no-cpu-baseline > $out/bench_${v:-new}_$r.json 2> $out/bench_${v:-new}_$r.err
    python - <<PY
import json
d=json.loads(open("$out/bench_${v:-new}_$r.json").read().strip().splitlines()[-1])
print("${v:-new} round $r:", d["value"], d["ms_per_step"], "attention", d["kernels"]["vpt_masked_attention_forward"]["ms"])
PY
  done
done

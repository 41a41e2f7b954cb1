#!/bin/bash
# round 4, run V: forward step with the residual prefetch in the last-but-one step (shipped) vs one step earlier (reslate0), same box
cd "$(dirname "$0")/.."; out=gpurun_out/r04_v; mkdir -p $out
for r in 1 2 3; do
  for v in "" reslate0; do
    lib=""; [ -n "$v" ] && lib="$PWD/video-pre-training_amd/build/libvpt_$v.so"
    VPT_HIP_LIB=$lib timeout 600 python bench.py --steps 8 --warmup 2 --bc-steps 0 --no-cpu-baseline > $out/bench_${v:-new}_$r.json 2> $out/bench_${v:-new}_$r.err
    python - <<PY
import json
d=json.loads(open("$out/bench_${v:-new}_$r.json").read().strip().splitlines()[-1])
print("${v:-new} round $r:", d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:v["ms"] for k,v in d["roofline"]["by_mode"].items()})
PY
  done
done

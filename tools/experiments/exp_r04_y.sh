#!/bin/bash
# round 4, run Y: stack 0's four block convolutions depth-first over sub-chunks of frames (Infinity Cache residency), same box
cd "$(dirname "$0")/.."; out=gpurun_out/r04_y; mkdir -p $out
run() { local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 6 --warmup 2 --bc-steps 0 --no-cpu-baseline > $out/b.json 2> $out/b.err
  python - <<PY
import json
d=json.loads(open("$out/b.json").read().strip().splitlines()[-1])
print("$label:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["by_mode"]["vpt_conv3x3_forward"])
PY
}
for r in 1 2; do
  run "whole chunk r$r" VPT_BLOCK_SUB=0
  run "sub 256 r$r" VPT_BLOCK_SUB=256
  run "sub 128 r$r" VPT_BLOCK_SUB=128
  run "sub 64 r$r" VPT_BLOCK_SUB=64
done

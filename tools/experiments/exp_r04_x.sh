#!/bin/bash
# round 4, run X: CNN chunk size / stream count of the forward engine with the fused path (defaults: 1024 frames, 3 streams), same box
cd "$(dirname "$0")/.."; out=gpurun_out/r04_x; mkdir -p $out
run() { # label env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 6 --warmup 2 --bc-steps 0 --no-cpu-baseline > $out/b.json 2> $out/b.err
  python - <<PY
import json
d=json.loads(open("$out/b.json").read().strip().splitlines()[-1])
print("$label:", d["value"], d["ms_per_step"])
PY
}
for r in 1 2; do
  run "chunk 1024 x 3 streams (default) r$r" VPT_X=0
  run "chunk 512 x 3 r$r" VPT_CNN_CHUNK=512
  run "chunk 2048 x 3 r$r" VPT_CNN_CHUNK=2048
  run "chunk 1024 x 2 r$r" VPT_CNN_STREAMS=2
  run "chunk 1024 x 4 r$r" VPT_CNN_STREAMS=4
  run "chunk 512 x 4 r$r" VPT_CNN_CHUNK=512 VPT_CNN_STREAMS=4
done

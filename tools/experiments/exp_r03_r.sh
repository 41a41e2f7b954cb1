#!/bin/bash
# conv + pool in sub-chunks of frames (pre-pool tensor pooled out of the Infinity Cache): forward sweep, two rounds
out=$PWD/gpurun_out/r03_r; mkdir -p $out
for r in 1 2; do
  for sub in 0 64 128 256; do
    VPT_POOL_SUBCHUNK=$sub timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --bc-steps 0 2>/dev/null | tail -1 > $out/sub${sub}_$r.json
  done
done
python - <<P
import json, glob
for t in sorted(glob.glob("$out/*.json")):
    d = json.load(open(t)); k = d.get("kernels", {})
    print(t.split("/")[-1], "value %.0f  ms/step %.2f  roofline %.4f  conv %.2f pool %.2f affine %.2f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], k["vpt_conv3x3_forward"]["ms"], k["vpt_maxpool_forward"]["ms"], k["vpt_frame_affine_forward"]["ms"]))
P

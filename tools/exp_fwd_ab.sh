#!/bin/bash
# headline forward bench A/B inside one call: reference library (build/libvpt_ref.so) vs working tree, two interleaved rounds
out=gpurun_out/exp_fwd_ab; mkdir -p $out
REF=$PWD/video-pre-training_amd/build/libvpt_ref.so
for r in 1 2; do
  VPT_HIP_LIB=$REF timeout 600 python bench.py --steps ${VPT_AB_STEPS:-10} --warmup 3 2>/dev/null | tail -1 > $out/ref_$r.json
  timeout 600 python bench.py --steps ${VPT_AB_STEPS:-10} --warmup 3 2>/dev/null | tail -1 > $out/new_$r.json
  python - <<P
import json
for t in ("ref_$r", "new_$r"):
    d = json.load(open("$out/%s.json" % t))
    print(t, "value %.0f  ms/step %.2f  roofline %.4f  bc %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("bc_step", {}).get("ms_per_step") if isinstance(d.get("bc_step"), dict) else d.get("bc_step")))
P
done

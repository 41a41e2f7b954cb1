#!/bin/bash
# round 4, run O: forward + BC step with / without the 256 x 256 GEMM kernel, same box
cd "$(dirname "$0")/.."; out=gpurun_out/r04_o; mkdir -p $out
for r in 1 2; do
  for v in 1 0; do
    VPT_LINEAR_256=$v timeout 600 python bench.py --steps 6 --warmup 2 --bc-steps 3 --bc-warmup 1 --no-cpu-baseline > $out/bench_${v}_$r.json 2> $out/bench_${v}_$r.err
    python - <<PY
import json
d=json.loads(open("$out/bench_${v}_$r.json").read().strip().splitlines()[-1])
print("gemm256=$v round $r:", d["value"], d["ms_per_step"], "linear", d["kernels"]["vpt_linear_forward"]["ms"], "bc", d["bc_step"]["ms_per_step"], {k:v for k,v in d["bc_step"]["kernels_ms"].items() if "linear" in k})
PY
  done
done

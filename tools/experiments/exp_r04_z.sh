#!/bin/bash
# round 4, run Z: split-K factor of the dense GEMM (K = 16384, M = 1024 per chunk, N = 512), same box
cd "$(dirname "$0")/.."; out=gpurun_out/r04_z; mkdir -p $out
for r in 1 2; do
  for sk in 16 32 8; do
    VPT_DENSE_SPLITK=$sk timeout 300 python bench.py --steps 6 --warmup 2 --bc-steps 0 --no-cpu-baseline > $out/b.json 2> $out/b.err
    python - <<PY
import json
d=json.loads(open("$out/b.json").read().strip().splitlines()[-1])
print("splitk $sk r$r:", d["value"], d["ms_per_step"], "linear", d["kernels"]["vpt_linear_forward"]["ms"], "fold epilogue", d["kernels"].get("vpt_dense_fold_epilogue",{}).get("ms"))
PY
  done
done
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1

#!/bin/bash
# Round 6, call R: split-K of the K = 65536 dense layer.  At 32 the GEMM launches 256 workgroups -- half of the 512 slots (2 per CU) -- and each runs 32 k-steps
# with ONE step of prefetch against HBM latency (single-stream: 186 us per 1024-frame chunk = 185 TF/s, 0.9 TB/s).  More splits = more loads in flight.
mkdir -p gpurun_out/r06r
for rep in 1 2; do for sk in 32 64 128; do
  VPT_DENSE_SPLITK=$sk timeout 600 python bench.py --steps 10 --warmup 3 --bc-steps 0 --no-cpu-baseline --no-ingest --value-blocks 0 --no-dp-probe > gpurun_out/r06r/fwd_sk${sk}_$rep.json 2> gpurun_out/r06r/fwd_sk${sk}_$rep.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r06r/fwd_sk${sk}_$rep.json").read().strip().splitlines()[-1])
k = d.get("kernels_ms") or {}
print("splitk $sk rep $rep:", d["value"], d["ms_per_step"], d["roofline"]["frac"], {n: v for n, v in k.items() if "linear" in n or "dense" in n} if isinstance(k, dict) else "")
PY
done; done 2>&1 | tee gpurun_out/r06r/summary.log

#!/bin/bash
# round 4, call E: GroupNorm n folded into block 0 (parity + A/B)
out=gpurun_out/r04_e; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -s -k "folded or pool_fused or conv_first_pool or test_conv3x3" > $out/t1.log 2>&1; echo "t1 rc=$?"; grep -E "passed|failed|Error|NFOLD|assert" $out/t1.log | cut -c1-300 | tail -24
timeout 900 python -m pytest tests/test_gpu_policy.py tests/test_gpu_configs.py -q -x -s -k "chunks_vs_golden or full_chunk or config2_sequence or idm_4x_forward" > $out/t2.log 2>&1; echo "t2 rc=$?"; grep -E "passed|failed|Error|PARITY|assert" $out/t2.log | cut -c1-260 | tail -24
for r in 1 2; do
  for fn in 0 1; do
    VPT_FOLD_N=$fn timeout 300 python bench.py --steps 6 --warmup 2 --bc-steps 0 --no-cpu-baseline > $out/bench_fn${fn}_$r.json 2> $out/bench_fn${fn}_$r.err
    python - <<PY
import json
try:
    d=json.loads(open("$out/bench_fn${fn}_$r.json").read().strip().splitlines()[-1])
    k=d["kernels"]
    print("fold_n=$fn round $r:", d["value"], "frames/s", d["ms_per_step"], "ms; roofline", d["roofline"]["frac"], d["roofline"].get("by_mode"), {kk: v["ms"] for kk, v in k.items() if kk not in ("vpt_conv3x3_forward", "vpt_conv3x3_pool_forward")})
except Exception as e:
    print("bench fn=$fn failed", e); print(open("$out/bench_fn${fn}_$r.err").read()[-1500:])
PY
  done
done

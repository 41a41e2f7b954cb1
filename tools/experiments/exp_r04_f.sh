#!/bin/bash
# round 4, call F: per-channel sums inside the producers (no channel-stats pass) -- parity + A/B
out=gpurun_out/r04_f; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -s -k "folded or pool_fused or conv_first_pool" > $out/t1.log 2>&1; echo "t1 rc=$?"; grep -E "passed|failed|Error|assert" $out/t1.log | cut -c1-300 | tail -8
timeout 900 python -m pytest tests/test_gpu_policy.py tests/test_gpu_configs.py -q -x -s -k "chunks_vs_golden or full_chunk or config2_sequence or idm_4x_forward or wide_models" > $out/t2.log 2>&1; echo "t2 rc=$?"; grep -E "passed|failed|Error|assert" $out/t2.log | cut -c1-260 | tail -8
for r in 1 2; do
  for fs in 0 1; do
    VPT_FOLD_STATS=$fs timeout 300 python bench.py --steps 6 --warmup 2 --bc-steps 0 --no-cpu-baseline > $out/bench_fs${fs}_$r.json 2> $out/bench_fs${fs}_$r.err
    python - <<PY
import json
try:
    d=json.loads(open("$out/bench_fs${fs}_$r.json").read().strip().splitlines()[-1])
    k=d["kernels"]
    print("fold_stats=$fs round $r:", d["value"], "frames/s", d["ms_per_step"], "ms; roofline", d["roofline"]["frac"], d["roofline"].get("by_mode"), {kk: v["ms"] for kk, v in k.items() if kk not in ("vpt_conv3x3_forward", "vpt_conv3x3_pool_forward")})
except Exception as e:
    print("bench fs=$fs failed", e); print(open("$out/bench_fs${fs}_$r.err").read()[-1500:])
PY
  done
done

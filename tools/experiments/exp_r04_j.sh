#!/bin/bash
out=gpurun_out/r04_j; mkdir -p $out
for cfg in "VPT_FOLD_N=0 VPT_FUSE_POOL=0 VPT_FOLD_DENSE=0" "VPT_FOLD_N=0 VPT_FUSE_POOL=1 VPT_FOLD_DENSE=0" "VPT_FOLD_N=1 VPT_FOLD_STATS=0 VPT_FOLD_DENSE=0" "VPT_FOLD_N=1 VPT_FOLD_STATS=1 VPT_FOLD_DENSE=0" "VPT_FOLD_N=0 VPT_FUSE_POOL=0 VPT_FOLD_DENSE=1"; do
  env $cfg timeout 300 python -m pytest tests/test_gpu_policy.py -q -x -k "batch_around" > $out/t.log 2>&1; echo "$cfg -> rc=$?"; grep -E "passed|failed" $out/t.log | tail -1
done
timeout 900 python -m pytest tests/test_gpu_policy.py tests/test_gpu_configs.py -q -x -k "chunks_vs_golden or full_chunk or idm_4x_forward or wide_models" > $out/t2.log 2>&1; echo "t2 rc=$?"; grep -E "passed|failed|Error|assert" $out/t2.log | cut -c1-300 | tail -5
for r in 1 2; do for fd in 0 1; do
    VPT_FOLD_DENSE=$fd timeout 300 python bench.py --steps 6 --warmup 2 --bc-steps 0 --no-cpu-baseline > $out/bench_$fd_$r.json 2> $out/bench_$r.err
    python - <<PY
import json
d=json.loads(open("$out/bench_$fd_$r.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("fold_dense=$fd round $r:", d["value"], "frames/s", d["ms_per_step"], "ms; roofline", d["roofline"]["frac"], {kk: v["ms"] for kk, v in k.items() if "conv3x3" not in kk})
PY
done; done

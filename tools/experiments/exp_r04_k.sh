#!/bin/bash
out=gpurun_out/r04_k; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_policy.py -q -x -k "batch_around or chunks_vs_golden or step_graph or ragged or first_resets" > $out/t1.log 2>&1; echo "t1 rc=$?"; grep -E "passed|failed|Error|assert" $out/t1.log | cut -c1-400 | tail -6
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_kernels.py -q -x -k "idm_4x_forward or wide_models or test_linear" > $out/t2.log 2>&1; echo "t2 rc=$?"; grep -E "passed|failed|Error|assert" $out/t2.log | cut -c1-300 | tail -5
for r in 1 2; do for fd in 0 1; do
    VPT_FOLD_DENSE=$fd timeout 300 python bench.py --steps 6 --warmup 2 --bc-steps 0 --no-cpu-baseline > $out/bench_${fd}_$r.json 2> $out/bench_${fd}_$r.err
    python - <<PY
import json
d=json.loads(open("$out/bench_${fd}_$r.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("fold_dense=$fd round $r:", d["value"], "frames/s", d["ms_per_step"], "ms; roofline", d["roofline"]["frac"], {kk: v["ms"] for kk, v in k.items() if "conv3x3" not in kk})
PY
done; done

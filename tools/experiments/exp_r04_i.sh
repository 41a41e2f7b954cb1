#!/bin/bash
out=gpurun_out/r04_i; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_policy.py tests/test_gpu_kernels.py -q -x -k "batch_around or conv_first_pool or folded or step_graph or act_uses" > $out/t1.log 2>&1; echo "t1 rc=$?"; grep -E "passed|failed|Error|assert" $out/t1.log | cut -c1-300 | tail -8
for r in 1 2; do
    timeout 300 python bench.py --steps 6 --warmup 2 --bc-steps 0 --no-cpu-baseline > $out/bench_$r.json 2> $out/bench_$r.err
    python - <<PY
import json
d=json.loads(open("$out/bench_$r.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("round $r:", d["value"], "frames/s", d["ms_per_step"], "ms; roofline", d["roofline"]["frac"], {kk: v["ms"] for kk, v in k.items()})
PY
done

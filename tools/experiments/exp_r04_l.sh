#!/bin/bash
# round 4, run L: 32-row conv tiles (8 waves, one weight fetch per 512 pixels) vs the 16-row tiles, same box
cd "$(dirname "$0")/.."; out=gpurun_out/r04_l; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv3x3 or n_folded or dgrad or backward" > $out/t1.log 2>&1; echo "t1 rc=$?"; tail -3 $out/t1.log
for r in 1 2 3; do
  for t in throughput32 throughput; do
    echo "== $t round $r"
    VPT_BENCH_TILING=$t VPT_BENCH_POOL=0 timeout 300 python tools/conv_bench.py 2>&1 | grep -v "^$" | tee $out/cb_${t}_$r.log
  done
done
timeout 600 python bench.py --steps 6 --warmup 2 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_l/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("by_mode"))
PY

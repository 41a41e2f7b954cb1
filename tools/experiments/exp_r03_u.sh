#!/bin/bash
# launch-structure sweep with the conv + pool sub-chunks on: CNN streams x chunk size, forward only
out=$PWD/gpurun_out/r03_u; mkdir -p $out
for r in 1 2; do
for c in 1024 2048; do for s in 2 3 4; do
  VPT_CNN_CHUNK=$c VPT_CNN_STREAMS=$s timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --bc-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('chunk $c streams $s round $r: %.0f frames/s  %.2f ms' % (d['value'], d['ms_per_step']))"
done; done; done

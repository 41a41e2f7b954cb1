#!/bin/bash
# round 5, call l: cross-step overlap (PolicyEngine.overlap_steps): the CNN of forward call i + 1 beside the transformer of call i.  Bit-identity test, then the timed forward A/B.
out=gpurun_out/r05_l; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_policy.py -q -m gpu -x -k "overlapped or batch_around or row_count or full_chunk" > $out/test.log 2>&1; echo "test rc=$?" >> $out/test.log
grep -E "passed|failed|^FAILED|Error|rc=" $out/test.log | cut -c1-300 | tail -8
for r in 1 2 3; do
  for ov in 0 1; do
    echo "== overlap $ov round $r"; timeout 300 python bench.py --steps 10 --warmup 2 --bc-steps 0 --no-cpu-baseline --no-ingest --step-overlap $ov 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
  done
done
echo "== fp16"; for ov in 0 1; do timeout 300 python bench.py --precision fp16 --steps 10 --warmup 2 --bc-steps 0 --no-cpu-baseline --no-ingest --step-overlap $ov 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; done

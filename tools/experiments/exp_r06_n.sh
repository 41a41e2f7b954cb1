#!/bin/bash
# (run on the trees that still had the LDS-tile kernel behind VPT_CONV_FIRST_LDS_TILE=1 -- commit 53cb17f and the working tree after it; the switch and
# that kernel were removed once the A/B was recorded: profiles/r06_experiments.md section 4)
# Round 6, call N: final vpt_conv_first_kernel (4-wave workgroups, zero records for the pool's padding) -- parity of everything that runs through it, then
# A/B against the LDS-tile kernel, then the forward bench.
mkdir -p gpurun_out/r06n
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp16_kernels.py tests/test_gpu_training.py -x -q -p no:cacheprovider -k "conv_first or pack" 2>&1 | tail -2
for i in 1 2; do
  timeout 300 python tools/conv_first_bench.py 1024 2>&1 | tail -1 | sed 's/^/new: /' | tee -a gpurun_out/r06n/ab.log
  VPT_CONV_FIRST_LDS_TILE=1 timeout 300 python tools/conv_first_bench.py 1024 2>&1 | tail -1 | sed 's/^/old: /' | tee -a gpurun_out/r06n/ab.log
done
timeout 1500 python -m pytest tests/test_gpu_policy.py tests/test_gpu_configs.py tests/test_gpu_dropin.py tests/test_gpu_idm.py -x -q -p no:cacheprovider 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-dp-probe > gpurun_out/r06n/bench.json 2> gpurun_out/r06n/bench.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06n/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('box'))
PY

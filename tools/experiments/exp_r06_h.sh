#!/bin/bash
# round 6, call H: what does building vpt_conv_first.hip without SLP vectorisation (no cross-half op_sel packed adds in its statistics) cost?
O=gpurun_out/r06_h; mkdir -p $O
export TMPDIR=/tmp
B=/root/repo/video-pre-training_amd/build
for rep in 1 2 3; do
  python tools/conv_first_bench.py 1024 2>&1 | grep -v amdgpu | tail -4 | tr '\n' ' '; echo " [shipped]"
  VPT_HIP_LIB=$B/libvpt_cf_noslp.so python tools/conv_first_bench.py 1024 2>&1 | grep -v amdgpu | tail -4 | tr '\n' ' '; echo " [no SLP]"
done
for rep in 1 2; do
  python bench.py --steps 10 --warmup 3 --bc-steps 0 --no-cpu-baseline --value-blocks 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shipped', d['value'], d['ms_per_step'], d['kernels']['vpt_conv_first_forward'])"
  VPT_HIP_LIB=$B/libvpt_cf_noslp.so python bench.py --steps 10 --warmup 3 --bc-steps 0 --no-cpu-baseline --value-blocks 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no SLP ', d['value'], d['ms_per_step'], d['kernels']['vpt_conv_first_forward'])"
done

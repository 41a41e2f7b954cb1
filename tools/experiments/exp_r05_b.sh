#!/bin/bash
# round 5, call b: gated dgrad (vpt_conv3x3_kernel mode 6) + vpt_conv_backward_reduce -- kernel tests, BC A/B in one call, conv bench of the mode-5
# layers (spill removed) vs the round-4 library, ingest probe, single-stream rocprof kernel stats of the BC step
out=gpurun_out/r05_b; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_kernels.py tests/test_gpu_fp16_kernels.py -q -x -m gpu -s \
  -k "gated or conv_layer or bc_gradients or bc_step or chunking or conv3x3 or group_norm_n or reference_bc_loop" > $out/test.log 2>&1; echo "test rc=$?" >> $out/test.log
grep -E "PARITY gated|passed|failed|Error|rc=" $out/test.log | cut -c1-400 | tail -20
for r in 1 2; do
  VPT_BC_GATED_DGRAD=0 timeout 300 python tools/bc_bench.py --steps 3 --streams1 2>&1 | grep -v amdgpu.ids > $out/old_$r.log; echo "== two-step_$r"; grep "BC step\|wgrad\|dgrad\|3x3_forward\|prepare\|kernel time" $out/old_$r.log
  timeout 300 python tools/bc_bench.py --steps 3 --streams1 2>&1 | grep -v amdgpu.ids > $out/new_$r.log; echo "== gated_$r"; grep "BC step\|wgrad\|dgrad\|3x3_forward\|prepare\|kernel time" $out/new_$r.log
done
R04=$PWD/video-pre-training_amd/build/libvpt_r04.so
for r in 1 2; do
  echo "== r04 forward round $r"; VPT_HIP_LIB=$R04 timeout 300 python bench.py --steps 8 --warmup 2 --bc-steps 0 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['by_mode'], {k: v['ms'] for k, v in d['kernels'].items() if v['ms'] > 0.25})"
  echo "== new forward round $r"; timeout 300 python bench.py --steps 8 --warmup 2 --bc-steps 0 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['by_mode'], {k: v['ms'] for k, v in d['kernels'].items() if v['ms'] > 0.25})"
done
timeout 300 python tools/ingest_probe.py 2>&1 | grep -v amdgpu.ids | tail -8
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/ingest_probe.py 2>&1 | grep -v amdgpu.ids | tail -4
cd /tmp && export TMPDIR=/tmp
VPT_BC_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof_bc -o bc -- python $GRAFT_REPO_ROOT/tools/bc_bench.py --steps 2 > $GRAFT_REPO_ROOT/$out/prof_bc.log 2>&1
cd $GRAFT_REPO_ROOT; find $out/prof_bc -name "*kernel_stats.csv" | head -2; f=$(find $out/prof_bc -name "*kernel_stats.csv" | head -1); head -45 "$f" | cut -c1-160
find $out/prof_bc -name "*.db" -delete; find $out/prof_bc -name "*kernel_trace.csv" -delete

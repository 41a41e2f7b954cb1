#!/bin/bash
# round 5, call c: the gated-dgrad tests that call b's -x cut off, the mode-5 / folded-inference tests, the arg-max-mask probe of the pool-fused
# epilogue (what would emitting the pool's arg-max cost mode 4?), the ingest leg with the copy stream created first, per-shape table of the BC
# step's convolution passes, single-stream rocprof kernel stats of the BC step
out=gpurun_out/r05_c; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_kernels.py tests/test_gpu_fp16_kernels.py -q -m gpu -s \
  -k "gated or conv_layer or chunking or conv3x3 or group_norm_n or reference_bc_loop or pool" > $out/test.log 2>&1; echo "test rc=$?" >> $out/test.log
grep -E "PARITY gated|PARITY BC gradients|passed|failed|^FAILED|rc=" $out/test.log | cut -c1-300 | tail -24
timeout 600 python -m pytest tests/test_gpu_policy.py tests/test_gpu_configs.py -q -m gpu -x -k "chunks_vs_golden or batch_around or row_count or config2 or bc_3x or config5" > $out/test2.log 2>&1; echo "test2 rc=$?" >> $out/test2.log
grep -E "passed|failed|^FAILED|rc=" $out/test2.log | cut -c1-300 | tail -6
PROBE=$PWD/video-pre-training_amd/build/libvpt_argmask.so
export VPT_BENCH_SHAPES="s1.first,64,128,256,0;s2.first,32,256,256,0"
for r in 1 2 3; do
  echo "== shipped round $r"; VPT_BENCH_POOL_PROBE=1 timeout 300 python tools/conv_bench.py 1024 2>&1 | grep -E "fused|conv\+pool" 
  echo "== argmask round $r"; VPT_BENCH_POOL_PROBE=1 VPT_HIP_LIB=$PROBE timeout 300 python tools/conv_bench.py 1024 2>&1 | grep -E "fused|conv\+pool"
done
unset VPT_BENCH_SHAPES
for r in 1 2; do timeout 300 python bench.py --steps 4 --warmup 2 --ingest-only 2>/dev/null | tail -1; done
timeout 300 python tools/bc_bench.py --steps 3 --streams1 2>&1 | grep -v amdgpu.ids > $out/bc_shapes.log; grep -A60 "BC step" $out/bc_shapes.log | cut -c1-150
cd /tmp && export TMPDIR=/tmp
VPT_BC_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_bc -o bc -- python $GRAFT_REPO_ROOT/tools/bc_bench.py --steps 2 > $GRAFT_REPO_ROOT/$out/prof_bc.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $out/prof_bc -name "*kernel_stats.csv" | head -1); echo "stats file: $f"; head -50 "$f" | cut -c1-170
find $out/prof_bc -name "*kernel_trace.csv" -delete; find $out/prof_bc -name "*.db" -delete

#!/bin/bash
# round 4, call B: pool-fused conv on all shapes, sampling tests, epilogue ablations on s0, fused sub-chunk sweep
out=gpurun_out/r04_b; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sampling.py -q -s -k "pool_fused or sampling or uniforms or draw or act_stochastic" > $out/t1.log 2>&1; echo "t1 rc=$?"; grep -E "passed|failed|Error|SAMPLING|assert" $out/t1.log | cut -c1-300 | tail -15
B=$PWD/video-pre-training_amd/build
export VPT_BENCH_SHAPES="s0.res,64,128,128,1;s0.nores,64,128,128,0;s1.res,32,256,256,1;s1.nores,32,256,256,0" VPT_BENCH_POOL=0
for r in 1 2; do
  for n in new epi1 epi2 epi3; do
    if [ $n = new ]; then timeout 300 python tools/conv_bench.py 512 5 > $out/cb_${n}_$r.log 2>&1; else VPT_HIP_LIB=$B/libvpt_$n.so timeout 300 python tools/conv_bench.py 512 5 > $out/cb_${n}_$r.log 2>&1; fi
    echo "== $n round $r"; grep -v "Warn\|amdgpu.ids" $out/cb_${n}_$r.log | cut -c1-120
  done
done
unset VPT_BENCH_SHAPES VPT_BENCH_POOL
for r in 1 2; do
  for sub in 0 256 512; do
    VPT_FUSE_POOL_SUB=$sub timeout 300 python bench.py --steps 6 --warmup 2 --bc-steps 0 --no-cpu-baseline > $out/bench_sub${sub}_$r.json 2> $out/bench_sub${sub}_$r.err
    python - <<PY
import json
try:
    d=json.loads(open("$out/bench_sub${sub}_$r.json").read().strip().splitlines()[-1])
    k=d["kernels"]
    print("fuse sub=$sub round $r:", d["value"], "frames/s", d["ms_per_step"], "ms; roofline", d["roofline"]["frac"], d["roofline"].get("by_mode"), "; seam", k.get("vpt_pool_seam",{}).get("ms"), "affine", k["vpt_frame_affine_forward"]["ms"], "first", k["vpt_conv_first_forward"]["ms"], "gemm", k["vpt_linear_forward"]["ms"])
except Exception as e:
    print("bench sub=$sub failed", e); print(open("$out/bench_sub${sub}_$r.err").read()[-1500:])
PY
  done
done

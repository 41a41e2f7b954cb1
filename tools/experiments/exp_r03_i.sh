#!/bin/bash
# conv_first with 16-bit input tile + slot-ordered K, gemm with clamped A rows: tests, micro-benchmarks and forward A/B against build/libvpt_ref.so
out=$PWD/gpurun_out/r03_i; mkdir -p $out
REF=$PWD/video-pre-training_amd/build/libvpt_ref.so
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp16_kernels.py tests/test_gpu_training.py tests/test_gpu_policy.py -q -x > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log
grep -E "passed|failed|^FAILED|^ERROR|rc=" $out/tests.log | cut -c1-300 | tail -6
for lib in ref new; do
  if [ $lib = ref ]; then export VPT_HIP_LIB=$REF; else unset VPT_HIP_LIB; fi
  echo "== $lib"
  timeout 300 python tools/conv_first_bench.py 1024 2>&1 | grep conv_first | tee $out/conv_first_$lib.log
  timeout 300 python tools/gemm_bench.py 8192,2048,2048 8192,8192,2048 8192,2048,8192 8192,6304,2048 2>&1 | grep "M=" | tee $out/gemm_$lib.log
done
unset VPT_HIP_LIB
for r in 1 2; do
  VPT_HIP_LIB=$REF timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --bc-steps 4 2>/dev/null | tail -1 > $out/ref_$r.json
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --bc-steps 4 2>/dev/null | tail -1 > $out/new_$r.json
done
for c in 512 2048; do
  VPT_CNN_CHUNK=$c timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --bc-steps 0 2>/dev/null | tail -1 > $out/chunk_$c.json
done
python - <<P
import json, glob
for t in sorted(glob.glob("$out/*.json")):
    try:
        d = json.load(open(t))
    except Exception as e:
        print(t, "unreadable", e); continue
    bc = d.get("bc_step"); bc = bc.get("ms_per_step") if isinstance(bc, dict) else bc
    k = d.get("kernels", {})
    print(t.split("/")[-1], "value %.0f  ms/step %.2f  roofline %.4f  bc %s  conv_first %.2f linear %.2f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], bc, k.get("vpt_conv_first_forward", {}).get("ms", -1), k.get("vpt_linear_forward", {}).get("ms", -1)))
P

#!/bin/bash
# round 5, call k: residual modes (1, 5) with the ReLU through the clamp modifier of a 2^-40-scaled v_pk_fma_f32 (192 fewer vector instructions per wave and tile):
# kernel tests, bit-identity of outputs against the library before (build/libvpt_ref.so = HEAD), conv_bench / forward / BC A/B inside this one call
out=gpurun_out/r05_k; mkdir -p $out
REF=$PWD/video-pre-training_amd/build/libvpt_ref.so
timeout 300 python tools/conv_hash.py > $out/hash_new.txt 2>&1; VPT_HIP_LIB=$REF timeout 300 python tools/conv_hash.py > $out/hash_ref.txt 2>&1
if diff -q $out/hash_new.txt $out/hash_ref.txt > /dev/null; then echo "HASH identical ($(wc -l < $out/hash_new.txt) lines)"; else echo "HASH DIFFERS"; diff $out/hash_new.txt $out/hash_ref.txt | head -20; fi
grep -c "out<res 0.000 finite True" $out/hash_new.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp16_kernels.py -q -m gpu -x > $out/test.log 2>&1; echo "test rc=$?" >> $out/test.log
grep -E "passed|failed|^FAILED|Error|rc=" $out/test.log | cut -c1-300 | tail -8
for r in 1 2; do
  echo "== ref conv_bench $r"; VPT_HIP_LIB=$REF timeout 300 python tools/conv_bench.py 512 2>&1 | grep -E "median|fused"
  echo "== new conv_bench $r"; timeout 300 python tools/conv_bench.py 512 2>&1 | grep -E "median|fused"
done
for r in 1 2; do
  echo "== ref forward $r"; VPT_HIP_LIB=$REF timeout 300 python bench.py --steps 8 --warmup 2 --bc-steps 0 --no-cpu-baseline --no-ingest 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['by_mode'])"
  echo "== new forward $r"; timeout 300 python bench.py --steps 8 --warmup 2 --bc-steps 0 --no-cpu-baseline --no-ingest 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['by_mode'])"
done
echo "== ref bc"; VPT_HIP_LIB=$REF timeout 300 python tools/bc_bench.py --steps 4 2>&1 | grep -E "^BC step"
echo "== new bc"; timeout 300 python tools/bc_bench.py --steps 4 2>&1 | grep -E "^BC step"

#!/bin/bash
# round 6, call G: ISA-level bisect of the LayerNorm-backward fault INSIDE the stress that reproduces it (the kernel alone in every process, and beside
# a GEMM-only process, never failed: tools/ubench/pk_hazard/run.py, run2.py)
O=gpurun_out/r06_g; mkdir -p $O
export TMPDIR=/tmp
H=/root/repo/video-pre-training_amd/build/pk_hazard
B=/root/repo/video-pre-training_amd/build
timeout 300 python tools/kernel_stress.py 3 3000 VPT_HIP_LIB=$B/libvpt_lnb_nopk.so STRESS_PATHS=0 > $O/oldlib.log 2>&1
echo "== old SLP library (positive control): $(grep -h mismatching $O/oldlib.log | awk '{s+=$(NF-5)} END {print s}') wrong launches"
for v in base after_all_pk after_pk_fma after_pk_add after_pk_mul after_fmac after_mov_b64 after_opsel before_bpermute base; do
  timeout 300 python tools/kernel_stress.py 3 3000 STRESS_LN_HSACO=$H/$v.hsaco STRESS_PATHS=0 > $O/$v.log 2>&1
  echo "== $v: $(grep -h mismatching $O/$v.log | awk '{s+=$(NF-5)} END {print s}') wrong launches  [$(grep -h mismatching $O/$v.log | grep -v ' 0 mism' | sed 's/^ *rank . //;s/ mismatching.*//' | sort | uniq -c | tr '\n' ';' | cut -c1-300)]"
done

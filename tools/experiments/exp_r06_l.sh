#!/bin/bash
# Round 6, call L: what bounds vpt_conv_first_kernel (nine positions per lane)?  Timing-only builds (-DVPT_CF_ABLATE=bits; wrong results):
# 1 no output stores, 2 one window position instead of nine, 4 no statistics / gain / channel sums, 8 no raw -> operand conversion.
mkdir -p gpurun_out/r06l
L=video-pre-training_amd/build
for ab in 0 1 2 3 15; do
  if [ $ab = 0 ]; then lib=""; else lib=$PWD/$L/libvpt_cf_ab$ab.so; fi
  VPT_HIP_LIB=$lib timeout 300 python tools/conv_first_bench.py 1024 2>&1 | tail -1 | sed "s/^/ablate $ab: /" | tee -a gpurun_out/r06l/ablate.log
done

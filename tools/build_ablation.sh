#!/bin/bash
# Profiling builds of the conv kernel: libvpt_abl<k>.so = the bf16 library with vpt_conv3x3.hip compiled -DVPT_EPI_ABLATE=<k>.
# Use with VPT_HIP_LIB=video-pre-training_amd/build/libvpt_abl<k>.so (see _native.py).
set -e
cd "$(dirname "$0")/../video-pre-training_amd"
for k in "$@"; do
  mkdir -p build/abl$k
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-gpu-rdc -I../include $EXTRA_DEFS -DVPT_EPI_ABLATE=$k -c csrc/vpt_conv3x3.hip -o build/abl$k/vpt_conv3x3.o 2>/dev/null
  objs=$(ls build/bf16/*.o | grep -v vpt_conv3x3.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libvpt_abl$k.so $objs build/abl$k/vpt_conv3x3.o
  echo "built build/libvpt_abl$k.so"
done

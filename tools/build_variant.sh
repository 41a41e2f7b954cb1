#!/bin/bash
# Experiment builds of the conv kernel: build_variant.sh <name> <defines...>  ->  build/libvpt_<name>.so (bf16 library with
# vpt_conv3x3.hip compiled with the given -D flags).  Run with VPT_HIP_LIB=... (see _native.py).
set -e
cd "$(dirname "$0")/../video-pre-training_amd"
name=$1; shift
mkdir -p build/var_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-gpu-rdc -I../include "$@" -c csrc/vpt_conv3x3.hip -o build/var_$name/vpt_conv3x3.o 2>/dev/null
objs=$(ls build/bf16/*.o | grep -v vpt_conv3x3.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libvpt_$name.so $objs build/var_$name/vpt_conv3x3.o
echo "built build/libvpt_$name.so"

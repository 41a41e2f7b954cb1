#!/bin/bash
# build_variant_file.sh <name> <source.hip> <defines...> -> build/libvpt_<name>.so: the bf16 library with ONE source recompiled with extra -D flags
set -e
cd "$(dirname "$0")/../video-pre-training_amd"
name=$1; src=$2; shift 2
mkdir -p build/var_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-gpu-rdc -I../include "$@" -c csrc/$src.hip -o build/var_$name/$src.o 2>/dev/null
objs=$(ls build/bf16/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libvpt_$name.so $objs build/var_$name/$src.o
echo "built build/libvpt_$name.so"

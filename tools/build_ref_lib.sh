#!/bin/bash
# A/B baseline: build the bf16 library from the sources of a git revision (default HEAD) into build/libvpt_ref.so.
# Clock / power differ by a few percent from one GPU box to the next, so kernel variants are only comparable inside ONE gpurun
# call:  VPT_HIP_LIB=$PWD/video-pre-training_amd/build/libvpt_ref.so python tools/conv_bench.py ...   (see _native.py)
set -e
rev=${1:-HEAD}
root="$(cd "$(dirname "$0")/.." && pwd)"
tmp=$(mktemp -d)
git -C "$root" archive "$rev" video-pre-training_amd/csrc include | tar -x -C "$tmp"
out="$root/video-pre-training_amd/build/ref"; mkdir -p "$out"
pids=()
for src in "$tmp"/video-pre-training_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-gpu-rdc -I"$tmp/include" -c "$src" -o "$out/$(basename "${src%.hip}").o" 2>/dev/null &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/video-pre-training_amd/build/libvpt_ref.so" "$out"/*.o
rm -rf "$tmp"
echo "built video-pre-training_amd/build/libvpt_ref.so from $rev"

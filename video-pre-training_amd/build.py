"""Build libvpt_hip.so for gfx950 with hipcc, in-tree (video-pre-training_amd/csrc/ -> video-pre-training_amd/).

hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels to the GPU box.
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["vpt_conv3x3.hip", "vpt_conv_first.hip", "vpt_conv3d.hip", "vpt_elementwise.hip", "vpt_gemm.hip", "vpt_gemv.hip", "vpt_action_codec.hip",
           "vpt_transformer.hip", "vpt_optim.hip", "vpt_backward.hip", "vpt_cnn_backward.hip", "vpt_conv_wgrad.hip", "vpt_conv_first_bwd.hip", "vpt_capi.hip"]
LIB = os.path.join(HERE, "libvpt_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-gpu-rdc"]


def _fingerprint():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode())
            h.update(f.read())
    with open(os.path.join(HERE, "..", "include", "vpt_hip.h"), "rb") as f:
        h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    stamp = LIB + ".stamp"
    fp = _fingerprint()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == fp:
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        if os.path.exists(LIB):
            return LIB  # GPU box without a toolchain: use the prebuilt library that travelled with the tree
        raise RuntimeError("hipcc not found and no prebuilt libvpt_hip.so")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(fp)
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) // 1024} KiB)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

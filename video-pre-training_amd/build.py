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
SOURCES = ["vpt_conv3x3.hip", "vpt_conv_first.hip", "vpt_conv3d.hip", "vpt_elementwise.hip", "vpt_gemm.hip", "vpt_gemv.hip", "vpt_action_codec.hip", "vpt_clip.hip",
           "vpt_transformer.hip", "vpt_optim.hip", "vpt_backward.hip", "vpt_cnn_backward.hip", "vpt_conv_wgrad.hip", "vpt_conv_first_bwd.hip", "vpt_pack.hip", "vpt_reduce.hip", "vpt_capi.hip"]
LIB = os.path.join(HERE, "libvpt_hip.so")
LIB_F16 = os.path.join(HERE, "libvpt_hip_f16.so")   # same sources, 16-bit operands = IEEE half (precision="fp16")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-gpu-rdc"]
VARIANTS = [(LIB, "bf16", []), (LIB_F16, "f16", ["-DVPT_OPERAND_F16"])]


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
    """Build both libraries (bf16 default + fp16 parity mode); returns the path of the default one."""
    stamp = LIB + ".stamp"
    fp = _fingerprint()
    have = all(os.path.exists(lib) for lib, _, _ in VARIANTS)
    if not force and have and os.path.exists(stamp) and open(stamp).read().strip() == fp:
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        if have:
            return LIB  # GPU box without a toolchain: use the prebuilt libraries that travelled with the tree
        raise RuntimeError("hipcc not found and no prebuilt libvpt_hip.so")
    procs, link = [], []
    for lib, tag, extra in VARIANTS:
        objdir = os.path.join(HERE, "build", tag)
        os.makedirs(objdir, exist_ok=True)
        objs = []
        for src in SOURCES:
            obj = os.path.join(objdir, src.replace(".hip", ".o"))
            cmd = [hipcc] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            objs.append(obj)
        link.append((lib, objs))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {src}")
    for lib, objs in link:
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
        if verbose:
            print(f"built {lib} ({os.path.getsize(lib) // 1024} KiB)")
    with open(stamp, "w") as f:
        f.write(fp)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

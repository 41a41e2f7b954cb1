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
# every compile also reports each kernel's registers / spills / scratch / LDS (stderr remarks, saved next to the object as <source>.resources.txt):
# tests/test_kernel_resources_cpu.py holds the convolution kernels to zero spills and zero scratch
REMARKS = ["-Rpass-analysis=kernel-resource-usage"]
# Per-source flags: NO SLP vectorisation where the compiler (ROCm 7.2) would otherwise emit a packed fp32 add whose LOW lane reads the HIGH register of a
# source pair (`v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]`, its way of adding two neighbouring scalars).  In vpt_ln_bwd_kernel that instruction -- inside the
# packed update of the two running row sums (s1, s2) -- occasionally returns, in lanes 48..63 only (the wave's last 16-lane pass), the sum WITHOUT that
# operand when another process's waves share the SIMD: one term dy_i g_i of 16 lanes is missing from s1, one row of dx comes out shifted by a constant.
# This was the "2-rank deviation" of round 5.  Found with an instrumented build of the kernel (per-lane partial sums of every row:
# tools/kernel_stress.py STRESS_LN_HSACO / STRESS_LN_DEBUG, tools/ubench/pk_hazard/): 11 of 11 captured events are lanes 48..63, term k = 1 or 3 of a float4
# (the two terms added by that instruction), misfit 1e-7.  `s_nop 7` behind every packed instruction does not remove it (not a missing wait state);
# -fno-slp-vectorize and -O1 do (0 wrong launches in 54 000 against ~85).  vpt_conv3d.hip and vpt_conv_first.hip carry the same instruction in their
# frame-statistics sums (8 each, compiler-generated); they never failed in the stress, and are built without it as a precaution (+0.006 ms per 1024 frames on the
# first conv).
EXTRA_FLAGS = {"vpt_backward.hip": ["-fno-slp-vectorize"], "vpt_conv3d.hip": ["-fno-slp-vectorize"], "vpt_conv_first.hip": ["-fno-slp-vectorize"]}
VARIANTS = [(LIB, "bf16", []), (LIB_F16, "f16", ["-DVPT_OPERAND_F16"])]
# Test-only A/B library (never loaded by the product: tests/test_gpu_conv_clamp.py names it through VPT_HIP_LIB): the bf16 library with vpt_conv3x3.hip's
# residual epilogues built WITHOUT the inline-assembly clamp FMA (the fp32 v_max ReLU of rounds 1-4) -- the two must agree bit for bit.
LIB_NOCLAMP = os.path.join(HERE, "build", "libvpt_noclamp.so")


def resource_log(tag: str, src: str) -> str:
    return os.path.join(HERE, "build", tag, src.replace(".hip", ".resources.txt"))


def kernel_resources(tag: str = "bf16") -> dict:
    """{kernel symbol: {vgprs, agprs, sgpr_spill, vgpr_spill, scratch, occupancy, lds}} of the last build of one variant (the compiler's
    -Rpass-analysis=kernel-resource-usage remarks)."""
    import re
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "ScratchSize [bytes/lane]": "scratch",
            "Occupancy [waves/SIMD]": "occupancy", "LDS Size [bytes/block]": "lds"}
    out, cur = {}, None
    for src in SOURCES:
        path = resource_log(tag, src)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: run build(force=True) where hipcc is available")
        for line in open(path, errors="replace"):
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = out.setdefault(m.group(1), dict(source=src))
                continue
            m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
            if m and cur is not None and m.group(1).strip() in keys:
                cur[keys[m.group(1).strip()]] = int(m.group(2))
    return out


def _fingerprint():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode())
            h.update(f.read())
    with open(os.path.join(HERE, "..", "include", "vpt_hip.h"), "rb") as f:
        h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    """Build both libraries (bf16 default + fp16 parity mode); returns the path of the default one."""
    stamp = LIB + ".stamp"
    fp = _fingerprint()
    have = all(os.path.exists(lib) for lib, _, _ in VARIANTS) and (os.path.exists(LIB_NOCLAMP) or not os.path.exists(shutil.which("hipcc") or "/opt/rocm/bin/hipcc"))
    have_logs = all(os.path.exists(resource_log(tag, src)) for _, tag, _ in VARIANTS for src in SOURCES)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not force and have and os.path.exists(stamp) and open(stamp).read().strip() == fp and (have_logs or not os.path.exists(hipcc)):
        return LIB
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
            cmd = [hipcc] + FLAGS + REMARKS + EXTRA_FLAGS.get(src, []) + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
            procs.append((src, tag, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            objs.append(obj)
        link.append((lib, objs))
    for src, tag, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {src}")
        with open(resource_log(tag, src), "wb") as f:
            f.write(out)
    for lib, objs in link:
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
        if verbose:
            print(f"built {lib} ({os.path.getsize(lib) // 1024} KiB)")
    objdir = os.path.join(HERE, "build", "var_noclamp")
    os.makedirs(objdir, exist_ok=True)
    obj = os.path.join(objdir, "vpt_conv3x3.o")
    subprocess.check_call([hipcc] + FLAGS + ["-DVPT_EPI_NO_CLAMP_RELU=1", "-c", os.path.join(CSRC, "vpt_conv3x3.hip"), "-o", obj], stderr=subprocess.DEVNULL)
    others = [o for o in link[0][1] if not o.endswith("vpt_conv3x3.o")]
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_NOCLAMP] + others + [obj])
    with open(stamp, "w") as f:
        f.write(fp)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

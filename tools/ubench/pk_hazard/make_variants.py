"""ISA-level bisect of the vpt_ln_bwd_kernel fault: build the kernel WITH the compiler's SLP-packed row sums (the failing code),
and variants of its assembly with `s_nop 7` (8 wait states) inserted after one class of instructions each.  A variant that stops failing names the
instruction class whose result is consumed too early; if none does, the fault is not a missing wait state.

    python tools/ubench/pk_hazard/make_variants.py            ->  video-pre-training_amd/build/pk_hazard/<variant>.hsaco  (hipcc cross-compiles: no GPU needed)
    python tools/ubench/pk_hazard/run.py [procs] [iters]      ->  wrong launches per variant, beside procs - 1 other processes (needs an MI355X)"""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
CSRC = os.path.join(ROOT, "video-pre-training_amd", "csrc")
OUT = os.path.join(ROOT, "video-pre-training_amd", "build", "pk_hazard")
CLANG = "/opt/rocm/lib/llvm/bin/clang"
KERNEL = "_Z17vpt_ln_bwd_kernelILi4EEv12VptLnBwdArgs"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-gpu-rdc"]      # build.py's FLAGS, WITHOUT -fno-slp-vectorize

# variant -> regex of the instructions that get an `s_nop 7` behind them (inside the kernel only)
VARIANTS = {
    "base": None,
    "after_pk_fma": r"^\s+v_pk_fma_f32\b",
    "after_pk_add": r"^\s+v_pk_add_f32\b",
    "after_pk_mul": r"^\s+v_pk_mul_f32\b",
    "after_fmac": r"^\s+v_fmac_f32",
    "after_mov_b64": r"^\s+v_mov_b64",
    "after_opsel": r"^\s+v_pk_\w+_f32\b.*op_sel:\[0,1\]",
    "after_all_pk": r"^\s+v_pk_(fma|add|mul)_f32\b",
    "before_bpermute": None,     # handled below: s_nop in FRONT of every ds_bpermute_b32
}


def main():
    os.makedirs(OUT, exist_ok=True)
    asm = os.path.join(OUT, "vpt_backward_slp.s")
    subprocess.check_call(["hipcc"] + FLAGS + ["-S", "--cuda-device-only", os.path.join(CSRC, "vpt_backward.hip"), "-o", asm], stderr=subprocess.DEVNULL)
    lines = open(asm).read().splitlines()
    start = next(i for i, ln in enumerate(lines) if ln.startswith(KERNEL + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip() == "s_endpgm")
    body = lines[start:end + 1]
    n_pk = sum(bool(re.search(r"v_pk_(fma|add|mul)_f32", ln)) for ln in body)
    print(f"{KERNEL}: {len(body)} lines, {n_pk} packed fp32 instructions")
    for name, rx in VARIANTS.items():
        out, n = [], 0
        for i, ln in enumerate(lines):
            inside = start <= i <= end
            if inside and name == "before_bpermute" and re.match(r"^\s+ds_bpermute_b32", ln):
                out.append("\ts_nop 7")
                n += 1
            out.append(ln)
            if inside and rx and re.search(rx, ln):
                out.append("\ts_nop 7")
                n += 1
        src = os.path.join(OUT, f"{name}.s")
        open(src, "w").write("\n".join(out) + "\n")
        hsaco = os.path.join(OUT, f"{name}.hsaco")
        subprocess.check_call([CLANG, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", src, "-o", hsaco])
        print(f"  {name}: {n} s_nop inserted -> {hsaco} ({os.path.getsize(hsaco)} B)")
    # the SLP build with the diagnostics stores (per-lane partial sums and reduced sums of every row): debug.hsaco
    dbg = os.path.join(OUT, "debug.s")
    subprocess.check_call(["hipcc"] + FLAGS + ["-DVPT_LN_DEBUG", "-S", "--cuda-device-only", os.path.join(CSRC, "vpt_backward.hip"), "-o", dbg], stderr=subprocess.DEVNULL)
    subprocess.check_call([CLANG, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", dbg, "-o", os.path.join(OUT, "debug.hsaco")])
    n_pk = 0
    inside = False
    for ln in open(dbg):
        if ln.startswith(KERNEL + ":"):
            inside = True
        if inside and re.search(r"v_pk_(fma|add|mul)_f32", ln):
            n_pk += 1
        if inside and ln.strip() == "s_endpgm":
            break
    print(f"  debug: {n_pk} packed fp32 instructions in the instrumented kernel -> {os.path.join(OUT, 'debug.hsaco')}")


if __name__ == "__main__":
    main()

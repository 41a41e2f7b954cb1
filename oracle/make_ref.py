"""Package the UNMODIFIED reference for the CPU-baseline leg of bench.py  --  TEST / MEASUREMENT INFRASTRUCTURE.

    python oracle/make_ref.py        (build container only: needs /root/reference; __graft_entry__.build() calls it)

The reference is pure Python, so there is nothing to compile: its `lib/` package and the two agent modules are zipped,
byte for byte, into oracle/_ref/vpt_reference.zip (git-ignored -- reference sources never enter the repository's
history -- but not gpurun-ignored, so the archive travels to the GPU box like the built .so files).  bench.py puts the
archive (zipimport) and oracle/ref_stubs on sys.path and times the reference's own MinecraftAgentPolicy on the host cores:
`cpu_baseline.kind = "reference"`.  Nothing in the product path imports it.
"""
import os
import sys
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "vpt_reference.zip")
FILES = ["agent.py", "inverse_dynamics_model.py"]


def make(verbose=True):
    if not os.path.isdir(os.path.join(REF, "lib")):
        return None
    os.makedirs(OUT_DIR, exist_ok=True)
    names = list(FILES) + ["lib/" + f for f in sorted(os.listdir(os.path.join(REF, "lib"))) if f.endswith(".py")]
    newest = max(os.path.getmtime(os.path.join(REF, n)) for n in names)
    if os.path.exists(OUT) and os.path.getmtime(OUT) >= newest:
        return OUT
    with zipfile.ZipFile(OUT, "w", zipfile.ZIP_DEFLATED) as z:
        for n in names:
            z.write(os.path.join(REF, n), n)
    if verbose:
        print(f"packaged {len(names)} reference modules -> {OUT}")
    return OUT


def reference_path():
    """The archive if present (else None): sys.path entry for `import lib.policy` of the unmodified reference."""
    return OUT if os.path.exists(OUT) else None


if __name__ == "__main__":
    p = make()
    print(p or "no /root/reference here: nothing to package")
    sys.exit(0)

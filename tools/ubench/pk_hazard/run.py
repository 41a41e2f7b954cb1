"""Stress the variants built by make_variants.py: every process loads each .hsaco with the HIP module API, launches vpt_ln_bwd_kernel<4> (M = 10 rows,
D = 1024: the 2-rank test's shape) `iters` times on fixed inputs and counts the launches whose dx differs from the first one in any bit.
    python tools/ubench/pk_hazard/run.py [procs=3] [iters=4000]"""
import ctypes
import os
import sys
import tempfile

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
OUT = os.path.join(ROOT, "video-pre-training_amd", "build", "pk_hazard")
KERNEL = b"_Z17vpt_ln_bwd_kernelILi4EEv12VptLnBwdArgs"
FINISH = b"_Z24vpt_ln_bwd_finish_kernelPKfiiPfS1_"


class Args(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("x", "gain", "dy", "dx_add", "dx", "dgain", "dbias", "partials")] + [(n, ctypes.c_int) for n in ("M", "D", "relu_in")]


def _barrier(d, tag, rank, world):
    open(os.path.join(d, f"bar_{tag}_{rank}"), "w").close()
    while not all(os.path.exists(os.path.join(d, f"bar_{tag}_{r}")) for r in range(world)):
        pass


def noise(d):
    """The last process: a foreign MATRIX-CORE workload (bf16 GEMMs on every CU) until the others are done.  The fault never showed with LayerNorm kernels
    alone in every process; in tools/kernel_stress.py and in the trainer the neighbours run MFMA kernels."""
    a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
    n = 0
    while not os.path.exists(os.path.join(d, "stop")):
        for _ in range(50):
            c = a @ b
        torch.cuda.synchronize()
        n += 50
    print(f"  noise process: {n} GEMMs", flush=True)


def worker(rank, world, d, iters, variants):
    if os.environ.get("PK_NOISE", "1") == "1" and rank == world - 1:
        return noise(d)
    if os.environ.get("PK_NOISE", "1") == "1":
        world -= 1
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    m, dd = 10, 1024
    # the configuration that fails most often in tools/kernel_stress.py: ReLU on the input, a skip tensor added, two fill kernels in front of every launch
    x = torch.randn(m, dd, generator=g).to(dev)
    dy = (torch.randn(m, dd, generator=g) * 1e-3).to(dev)
    gain = (1 + 0.1 * torch.randn(dd, generator=g)).to(dev)
    dxa = (torch.randn(m, dd, generator=g) * 1e-3).to(dev)
    dx = torch.empty_like(x)
    dg, db = torch.zeros(dd, device=dev), torch.zeros(dd, device=dev)
    part = torch.empty(4 * ((m + 31) // 32) * 2 * dd, device=dev)
    a = Args(x.data_ptr(), gain.data_ptr(), dy.data_ptr(), dxa.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), part.data_ptr(), m, dd, 1)
    size = ctypes.c_size_t(ctypes.sizeof(a))
    extra = (ctypes.c_void_p * 5)(1, ctypes.cast(ctypes.pointer(a), ctypes.c_void_p), 2, ctypes.cast(ctypes.pointer(size), ctypes.c_void_p), 3)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    class FinArgs(ctypes.Structure):
        _fields_ = [("partials", ctypes.c_void_p), ("nblocks", ctypes.c_int), ("D", ctypes.c_int), ("dgain", ctypes.c_void_p), ("dbias", ctypes.c_void_p)]
    fa = FinArgs(part.data_ptr(), 4 * ((m + 31) // 32), dd, dg.data_ptr(), db.data_ptr())
    fsize = ctypes.c_size_t(ctypes.sizeof(fa))
    extra_f = (ctypes.c_void_p * 5)(1, ctypes.cast(ctypes.pointer(fa), ctypes.c_void_p), 2, ctypes.cast(ctypes.pointer(fsize), ctypes.c_void_p), 3)
    lines = []
    for vi, name in enumerate(variants):
        mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
        rc = hip.hipModuleLoad(ctypes.byref(mod), os.path.join(OUT, name + ".hsaco").encode())
        rc = rc or hip.hipModuleGetFunction(ctypes.byref(fn), mod, KERNEL)
        fin = ctypes.c_void_p()
        rc = rc or hip.hipModuleGetFunction(ctypes.byref(fin), mod, FINISH)
        if rc:
            lines.append(f"  rank {rank} {name}: module load failed ({rc})")
            continue

        def launch():      # what ops.layernorm_backward enqueues: two fills, the main kernel, the finish kernel
            dg.zero_(); db.zero_()
            r = hip.hipModuleLaunchKernel(fn, (m + 31) // 32, 1, 1, 256, 1, 1, 0, stream, None, extra)
            assert r == 0, r
            r = hip.hipModuleLaunchKernel(fin, (2 * dd + 15) // 16, 1, 1, 256, 1, 1, 0, stream, None, extra_f)
            assert r == 0, r
        launch()
        torch.cuda.synchronize()
        ref = dx.clone()
        _barrier(d, f"{vi}_{name}", rank, world)
        bad = torch.zeros(1, dtype=torch.int64, device=dev)
        for _ in range(iters):
            launch()
            bad += (dx != ref).any().to(torch.int64)
        torch.cuda.synchronize()
        lines.append(f"  rank {rank} {name}: {int(bad.item())} wrong launches of {iters}")
        hip.hipModuleUnload(mod)
    with open(os.path.join(d, f"out{rank}.txt"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    _barrier(d, "end", rank, world)
    if rank == 0:
        open(os.path.join(d, "stop"), "w").close()


def main():
    import torch.multiprocessing as mp
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    variants = sorted(f[:-6] for f in os.listdir(OUT) if f.endswith(".hsaco"))
    variants = ["base"] + [v for v in variants if v != "base"] + ["base"]          # the failing build first AND last: the box must still show the fault at the end
    print(f"=== pk_hazard: {procs} processes, {iters} launches per variant: {variants}", flush=True)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(worker, args=(procs, d, iters, variants), nprocs=procs, join=True)
        for r in range(procs):
            if os.path.exists(os.path.join(d, f"out{r}.txt")):
                print(open(os.path.join(d, f"out{r}.txt")).read().rstrip(), flush=True)


if __name__ == "__main__":
    main()

"""Where does the fault need to be launched from?  The same LayerNorm-backward configuration (M = 10, D = 1024, ReLU input, skip tensor) through
  H1  the SLP build loaded as a separate code object (hipModuleLaunchKernel; make_variants.py's base.hsaco),
  H2  the C entry point of an SLP-built library (VPT_OLD_LIB), static buffers,
  H3  the same entry point with a fresh torch.empty dx / workspace per call (what ops.layernorm_backward does),
  H4  H3 through the shipped (fixed) library,
each `iters` times in `procs` concurrent processes, dx compared with the first result bit for bit.
    python tools/ubench/pk_hazard/run2.py [procs=3] [iters=6000]     (VPT_OLD_LIB=path of a library whose vpt_backward.hip was built with SLP vectorisation)"""
import ctypes
import os
import sys
import tempfile

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from run import Args, KERNEL, OUT, _barrier  # noqa: E402

OLD = os.environ.get("VPT_OLD_LIB", os.path.join(ROOT, "video-pre-training_amd", "build", "libvpt_lnb_nopk.so"))
NEW = os.path.join(ROOT, "video-pre-training_amd", "libvpt_hip.so")
P, I = ctypes.c_void_p, ctypes.c_int


def worker(rank, world, d, iters):
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    m, dd = 10, 1024
    x = torch.randn(m, dd, generator=g).to(dev)
    dy = (torch.randn(m, dd, generator=g) * 1e-3).to(dev)
    gain = (1 + 0.1 * torch.randn(dd, generator=g)).to(dev)
    dxa = (torch.randn(m, dd, generator=g) * 1e-3).to(dev)
    dg, db = torch.zeros(dd, device=dev), torch.zeros(dd, device=dev)
    nfloats = 4 * ((m + 31) // 32) * 2 * dd
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    static_dx, static_part = torch.empty_like(x), torch.empty(nfloats, device=dev)

    def via_module():
        mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipModuleLoad(ctypes.byref(mod), os.path.join(OUT, "base.hsaco").encode()) == 0
        assert hip.hipModuleGetFunction(ctypes.byref(fn), mod, KERNEL) == 0
        a = Args(x.data_ptr(), gain.data_ptr(), dy.data_ptr(), dxa.data_ptr(), static_dx.data_ptr(), dg.data_ptr(), db.data_ptr(), static_part.data_ptr(), m, dd, 1)
        size = ctypes.c_size_t(ctypes.sizeof(a))
        extra = (ctypes.c_void_p * 5)(1, ctypes.cast(ctypes.pointer(a), ctypes.c_void_p), 2, ctypes.cast(ctypes.pointer(size), ctypes.c_void_p), 3)

        def f():
            dg.zero_(); db.zero_()
            assert hip.hipModuleLaunchKernel(fn, 1, 1, 1, 256, 1, 1, 0, stream, None, extra) == 0
            return static_dx
        f.keep = (a, size, extra, mod)
        return f

    def via_lib(path, fresh):
        lib = ctypes.CDLL(path)
        fn = lib.vpt_layernorm_backward
        fn.argtypes, fn.restype = [P] * 8 + [I, I, I, P], I

        def f():
            dg.zero_(); db.zero_()
            dx = torch.empty_like(x) if fresh else static_dx
            part = torch.empty(nfloats, device=dev) if fresh else static_part
            assert fn(x.data_ptr(), gain.data_ptr(), dy.data_ptr(), dxa.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), part.data_ptr(), m, dd, 1, stream) == 0
            return dx
        return f

    cases = {"H1 separate code object": via_module(), "H2 SLP library, static buffers": via_lib(OLD, False), "H3 SLP library, fresh buffers": via_lib(OLD, True),
             "H4 shipped library, fresh buffers": via_lib(NEW, True), "H2 again": via_lib(OLD, False)}
    lines = []
    for ci, (name, f) in enumerate(cases.items()):
        ref = f().clone()
        torch.cuda.synchronize()
        _barrier(d, f"c{ci}", rank, world)
        bad = torch.zeros(1, dtype=torch.int64, device=dev)
        for _ in range(iters):
            bad += (f() != ref).any().to(torch.int64)
        torch.cuda.synchronize()
        lines.append(f"  rank {rank} {name}: {int(bad.item())} wrong launches of {iters}")
    with open(os.path.join(d, f"out{rank}.txt"), "w") as fh:
        fh.write("\n".join(lines) + "\n")


def main():
    import torch.multiprocessing as mp
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
    print(f"=== pk_hazard run2: {procs} processes, {iters} launches per case; SLP library {OLD}", flush=True)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(worker, args=(procs, d, iters), nprocs=procs, join=True)
        for r in range(procs):
            print(open(os.path.join(d, f"out{r}.txt")).read().rstrip(), flush=True)


if __name__ == "__main__":
    main()

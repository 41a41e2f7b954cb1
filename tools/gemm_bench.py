"""GEMM micro-benchmark (vpt_linear_forward): python tools/gemm_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from vpt_amd import ops, packing


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        e.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(e) / reps)
    return sorted(ts)[1]


shapes = [(8192, 8192, k) for k in (256, 1024, 2048, 4096, 8192)] + [(8192, 2048, 2048), (8192, 2048, 8192), (8192, 6304, 2048), (1024, 65536, 256), (65536, 256, 1024)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in s.split(",")) for s in sys.argv[1:]]
for m, n, k in shapes:
    A = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    W = packing.pack_linear(torch.randn(n, k, device="cuda") / k ** 0.5)
    for mode in ("f32", "bf16"):
        for tiling in ("throughput256", "throughput"):   # 256 x 256 / eight-wave LDS-DMA kernel where the grid fills the chip, vs the shipped 256 x 128 tiles
            t = timeit(lambda: ops.linear(A, W, n, out_f32=(mode == "f32"), out_bf16=(mode == "bf16"), tiling=tiling))
            print(f"M={m:6d} N={n:6d} K={k:6d} out={mode:4s} {tiling:13s}: {t:7.3f} ms  {2.0 * m * n * k / t / 1e9:7.1f} TF/s")

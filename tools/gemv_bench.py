"""The weight-streaming linears of the T = 1 acting step, one shape at a time: time per launch and effective HBM rate, for every
rows-per-workgroup choice (vpt_gemv_set_rows: profiling hook) and with / without the fused LayerNorm prologue.  Each launch reads a
DIFFERENT copy of the weights (16 copies, > 256 MB in total for the large shapes) so nothing is served from the Infinity Cache, as
in the real step (482 MB of trunk weights per step).   python tools/gemv_bench.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from vpt_amd import ops, _native

DT = {"bf16": torch.bfloat16, "fp16": torch.float16}[os.environ.get("VPT_PRECISION", "bf16")]
lib = _native.load("bf16" if DT == torch.bfloat16 else "fp16")
dev = "cuda"
shapes = [("qkvr", 6304, 2048, True), ("mlp0", 8192, 2048, True), ("heads", 8763, 2048, True), ("proj", 2048, 2048, False), ("mlp1", 2048, 8192, False),
          ("last", 2048, 2048, True), ("img.linear", 2048, 256, True)]
g = torch.Generator().manual_seed(0)
for name, n, k, ln in shapes:
    copies = [ops.pack_linear((torch.randn(n, k, generator=g) / k ** 0.5).to(dev), dtype=DT) for _ in range(16)]
    x = torch.randn(1, k, generator=g).to(dev)
    gain, bias = torch.ones(k, device=dev), torch.zeros(k, device=dev)
    x16 = x.to(DT)
    line = f"{name:10s} N={n:5d} K={k:5d} {'LN' if ln else '  '} {n * k * 2 / 1e6:6.1f} MB:"
    for rows in (0, 2, 4, 8, 16):
        lib.vpt_gemv_set_rows(rows)
        def run(w):
            if ln:
                ops.layernorm_linear(x, gain, bias, w, n, dtype=DT)
            else:
                ops.linear(x16, w, n)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for w in copies[:4]:
                run(w)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()          # one hipGraph of 64 dependent launches: no host launch cost in the measurement
        with torch.cuda.graph(graph):
            for r in range(4):
                for w in copies:
                    run(w)
        graph.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        graph.replay()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 64
        del graph
        line += f"  rows={rows if rows else 'auto':>4}: {us:6.2f} us {n * k * 2 / us / 1e6:5.2f} TB/s"
    lib.vpt_gemv_set_rows(0)
    print(line)

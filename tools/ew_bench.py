"""Streaming-kernel bandwidth probe: frame_affine / maxpool / conv_backward_prepare on stack-0-sized tensors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from vpt_amd import ops

f, c, h = 1024, 128, 64
x = torch.randn(f, c // 32, h, h, 32, device="cuda").to(torch.bfloat16)
dy = torch.randn_like(x)
res = torch.randn_like(x)
g = torch.ones(c, device="cuda"); b = torch.zeros(c, device="cuda")
xf = x.float().reshape(f, -1).double()
st = torch.stack([xf.sum(1), (xf * xf).sum(1)], 1).contiguous()
so = torch.zeros(f, 2, dtype=torch.float64, device="cuda")
sa = torch.zeros(9, 128, device="cuda"); sg = torch.zeros(9, 128, device="cuda")
nbytes = x.numel() * 2


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        e.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(e) / reps)
    return sorted(ts)[2]


out = torch.empty_like(x)
t = timeit(lambda: ops.frame_affine(x, g, b, st, stats_out=so, out=out))
print(f"frame_affine      : {t:.3f} ms  {2 * nbytes / t / 1e9:.2f} TB/s (read + write)")
t = timeit(lambda: ops.maxpool(x, stats_out=so))
print(f"maxpool           : {t:.3f} ms  {1.25 * nbytes / t / 1e9:.2f} TB/s")
t = timeit(lambda: ops.conv_backward_prepare(dy, x, None, st, sa, sg, c))
print(f"prepare (no res)  : {t:.3f} ms  {3 * nbytes / t / 1e9:.2f} TB/s")
t = timeit(lambda: ops.conv_backward_prepare(dy, x, res, st, sa, sg, c))
print(f"prepare (res)     : {t:.3f} ms  {4 * nbytes / t / 1e9:.2f} TB/s")
dgn, dbn = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
t = timeit(lambda: ops.frame_affine_backward(x, dy, g, st, dgn, dbn))
print(f"affine backward   : {t:.3f} ms  {5 * nbytes / t / 1e9:.2f} TB/s (2 passes: 2 reads, then 2 reads + 1 write)")
t = timeit(lambda: out.copy_(x))
print(f"torch copy_       : {t:.3f} ms  {2 * nbytes / t / 1e9:.2f} TB/s")

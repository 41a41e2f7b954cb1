"""Does running the two passes of vpt_frame_affine_backward over blocks of frames that fit the 256 MB MALL save the second pass's HBM reads?  (round 4 probe)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpt_amd  # noqa: F401
from vpt_amd import ops

dev = torch.device("cuda:0")
for (f, c, h) in ((1024, 128, 64), (1024, 256, 32), (1024, 256, 16)):
    x = torch.randn(f, c // 32, h, h, 32, device=dev).to(torch.bfloat16)
    dy = torch.randn(f, c // 32, h, h, 32, device=dev).to(torch.bfloat16)
    gain = torch.ones(c, device=dev)
    st = torch.stack([x.float().reshape(f, -1).sum(1).double(), (x.float() ** 2).reshape(f, -1).sum(1).double()], 1).contiguous()
    dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    for blk in (f, 256, 128, 64, 32):
        def run():
            for i in range(0, f, blk):
                ops.frame_affine_backward(x[i:i + blk], dy[i:i + blk], gain, st[i:i + blk], dg, db)
        run(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(); e.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(e))
        ts.sort()
        mb = 2.0 * x.numel() * blk / f / 1e6 * 2
        print(f"frames={f} C={c} {h}x{h}: block {blk:5d} frames ({mb:7.1f} MB of x + dy per block): {ts[2]:7.3f} ms   {10.0 * x.numel() / ts[2] / 1e9:6.2f} TB/s (5 passes of the tensor)")

"""Micro-benchmark of vpt_conv_first_backward (HIP events).  Usage: python tools/conv_first_bwd_bench.py [frames] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from vpt_amd import ops, packing
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev, cout = "cuda", 128
g = torch.Generator().manual_seed(0)
W = torch.randn(cout, 3, 3, 3, generator=g) * 0.3
b = 0.1 * torch.randn(cout, generator=g)
img = torch.randint(0, 256, (frames, 128, 128, 3), generator=g, dtype=torch.uint8).to(dev)
dP = torch.randn(frames, cout // 32, 64, 64, 32, generator=g).to(torch.bfloat16).to(dev)
wf = packing.pack_conv_first(W.to(dev), b.to(dev))
out = ops.conv_first_backward(img, wf, dP, cout)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        out = ops.conv_first_backward(img, wf, dP, cout, out=out)
    e.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(e) / reps)
ts.sort()
print(f"conv_first_backward frames={frames}: median {ts[2]:.3f} ms  best {ts[0]:.3f} ms  {os.environ.get('VPT_HIP_LIB', '').split('/')[-1]}")

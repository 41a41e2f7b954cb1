"""Throughput of the clip data path kernel (vpt_clip_frames): decoded 640 x 360 BGR frames in HBM -> 128 x 128 RGB.
Usage: python tools/clip_bench.py [frames=4096] [height=360] [width=640]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as ge
ge.build()
import vpt_amd  # noqa: F401
from vpt_amd import clip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
h = int(sys.argv[2]) if len(sys.argv) > 2 else 360
w = int(sys.argv[3]) if len(sys.argv) > 3 else 640
g = torch.Generator().manual_seed(0)
frames = torch.randint(0, 256, (n, h, w, 3), generator=g, dtype=torch.uint8).cuda()
state = torch.zeros(n, 3, dtype=torch.int32)
state[:, 0] = (torch.rand(n, generator=g) < 0.3).int()
state[:, 1] = torch.randint(0, w, (n,), generator=g).int()
state[:, 2] = torch.randint(0, h, (n,), generator=g).int()
state = state.cuda()
cursor = np.random.default_rng(0).integers(0, 256, (16, 16, 4), dtype=np.uint8)
proc = clip.ClipFrameProcessor(cursor)
out = proc(frames, state)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        proc(frames, state, out=out)
    e.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(e) / 5)
ts.sort()
ms = ts[2]
gb = n * (h * w * 3 + 128 * 128 * 3) / 1e9
print(f"clip_frames {n} x {w}x{h} -> 128x128: median {ms:.3f} ms = {n / ms * 1e3 / 1e6:.2f} M frames/s, {gb / ms * 1e3:.0f} GB/s of frame bytes (whole source frames counted)")

"""Phase timeline of vpt_conv3x3_kernel workgroups (profiling): per CU, how much of the memory-bound prologue/epilogue of one
resident workgroup overlaps the MFMA main loop of the other.  Usage: python tools/conv_trace.py [hw cin cout res frames]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import __graft_entry__ as ge

ge.build()
from vpt_amd import _native, ops, packing  # noqa: E402

hw, cin, cout, use_res, f = 64, 128, 128, 1, 512
if len(sys.argv) > 5:
    hw, cin, cout, use_res, f = (int(v) for v in sys.argv[1:6])
dev = "cuda"
g = torch.Generator().manual_seed(0)
W = torch.randn(cout, cin, 3, 3, generator=g) * (1.6 / (cin * 9) ** 0.5)
wpk, sa, sg = packing.pack_conv3x3(W.to(dev), torch.ones(cin, device=dev), torch.zeros(cin, device=dev))
x = torch.relu(torch.randn(f, cin // 32, hw, hw, 32, device=dev)).to(torch.bfloat16)
res = torch.randn(f, cout // 32, hw, hw, 32, device=dev).to(torch.bfloat16) if use_res else None
xf = x.float().reshape(f, -1).double()
st_in = torch.stack([xf.sum(1), (xf * xf).sum(1)], 1).contiguous()
st_out = torch.zeros(f, 2, dtype=torch.float64, device=dev)
out = torch.empty(f, cout // 32, hw, hw, 32, dtype=torch.bfloat16, device=dev)
grid = f * (hw // 16) ** 2 * ((cout + 127) // 128)
trace = torch.zeros(grid, 12, dtype=torch.int64, device=dev)
lib = _native.load()
lib.vpt_conv3x3_set_trace.argtypes = [ctypes.c_void_p]
lib.vpt_conv3x3_set_trace.restype = None
for _ in range(2):
    ops.conv3x3(x, wpk, sa, sg, st_in, cout, res=res, stats_out=st_out, out=out)
torch.cuda.synchronize()
lib.vpt_conv3x3_set_trace(ctypes.c_void_p(trace.data_ptr()))
ops.conv3x3(x, wpk, sa, sg, st_in, cout, res=res, stats_out=st_out, out=out)
torch.cuda.synchronize()
lib.vpt_conv3x3_set_trace(None)
t = trace.cpu().numpy()
t0 = t[:, 0].min()
T = (t[:, :4] - t0) / 100.0  # us
key, hp = t[:, 4], t[:, 5]
print(f"shape {hw}x{hw} {cin}->{cout} res={use_res} frames={f}: grid {grid}, kernel span {T[:, 3].max():.1f} us, CUs seen {len(np.unique(key))}, high-prio WGs {int(hp.sum())}")
pro, main, epi = T[:, 1] - T[:, 0], T[:, 2] - T[:, 1], T[:, 3] - T[:, 2]
for name, v in (("prologue", pro), ("main loop", main), ("epilogue", epi), ("tile total", T[:, 3] - T[:, 0])):
    print(f"  {name:10s} us: mean {v.mean():6.2f}  p10 {np.percentile(v, 10):6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}")
E = (t[:, 6:11] - t0) / 100.0
names = ["epi start -> subtile 0 begins", "subtile 0", "subtile 1", "subtile 2", "subtile 3", "stats + barrier + atomics"]
segs = [E[:, 0] - T[:, 2], E[:, 1] - E[:, 0], E[:, 2] - E[:, 1], E[:, 3] - E[:, 2], E[:, 4] - E[:, 3], T[:, 3] - E[:, 4]]
print("  epilogue anatomy of wave 0 (us, mean / median / p90):")
for nme, v in zip(names, segs):
    print(f"    {nme:32s} {v.mean():6.2f} {np.median(v):6.2f} {np.percentile(v, 90):6.2f}")
# per CU: fraction of time with 0 / 1 / 2 workgroups in their main loop
span = T[:, 3].max()
n_main = {0: 0.0, 1: 0.0, 2: 0.0}
resident = 0.0
for k in np.unique(key):
    idx = np.where(key == k)[0]
    ev = []
    for i in idx:
        ev.append((T[i, 1], +1)); ev.append((T[i, 2], -1))
    ev.sort()
    cur, last = 0, 0.0
    for tt, d in ev:
        n_main[min(cur, 2)] += tt - last
        cur += d; last = tt
    n_main[0] += span - last
tot = sum(n_main.values())
print("  per-CU time share with N workgroups inside the main loop: " + ", ".join(f"N={n}: {v / tot:.3f}" for n, v in n_main.items()))
# a few CUs' timelines
for k in np.unique(key)[:2]:
    idx = np.where(key == k)[0]
    idx = idx[np.argsort(T[idx, 0])][:10]
    print(f"  CU key {k:#x}:")
    for i in idx:
        print(f"    blk {i:6d} hp={hp[i]} start {T[i,0]:7.2f} main {T[i,1]:7.2f}..{T[i,2]:7.2f} end {T[i,3]:7.2f}")

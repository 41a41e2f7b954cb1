"""Micro-benchmark of vpt_conv_first_forward / backward (2x: 3 -> 128 channels, 128x128 frames).  python tools/conv_first_bench.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from vpt_amd import ops, packing

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cout = 128
g = torch.Generator().manual_seed(0)
W = torch.randn(cout, 3, 3, 3, generator=g) * 0.3
b = 0.1 * torch.randn(cout, generator=g)
wfrag = packing.pack_conv_first(W.cuda(), b.cuda())
img = torch.randint(0, 256, (frames, 128, 128, 3), generator=g, dtype=torch.uint8).cuda()
st = torch.zeros(frames, 2, dtype=torch.float64, device="cuda")
dp = torch.randn(frames, cout // 32, 64, 64, 32, device="cuda").to(torch.bfloat16)


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


t1 = timeit(lambda: ops.conv_first(img, wfrag, cout, stats_out=st))
t2 = timeit(lambda: ops.conv_first(img, wfrag, cout, stats_out=None))
t3 = timeit(lambda: ops.conv_first_backward(img, wfrag, dp, cout))
gain = (1 + 0.1 * torch.randn(cout, generator=g)).cuda()
chs = torch.zeros(frames, cout, 2, dtype=torch.float64, device="cuda")
t4 = timeit(lambda: ops.conv_first(img, wfrag, cout, stats_out=st, out_gain=gain, chs_out=chs))   # the inference engine's call: gain of GroupNorm `n` folded in, per-channel sums
gb = frames * (128 * 128 * 3 + 64 * 64 * cout * 2) / 1e9
print(f"conv_first forward {frames} frames: {t1:.3f} ms with stats, {t2:.3f} ms without  ({gb / t1 * 1e3:.0f} GB/s algorithmic)   backward {t3:.3f} ms;  with out_gain + per-channel sums {t4:.3f} ms  {os.environ.get('VPT_HIP_LIB', '').split('/')[-1]}")

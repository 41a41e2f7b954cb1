"""Micro-benchmark of vpt_conv3x3_forward at the five shapes of the 2x IMPALA CNN (HIP events, per shape).
Usage: python tools/conv_bench.py [frames] [reps]      (run under rocprofv3 --pmc ... for counters)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import __graft_entry__ as ge

ge.build()
from vpt_amd import ops, packing  # noqa: E402

DT = {"bf16": torch.bfloat16, "fp16": torch.float16}[os.environ.get("VPT_PRECISION", "bf16")]   # operand format (library) under test
TILING = os.environ.get("VPT_BENCH_TILING", "throughput")   # "throughput32": the 32-row / eight-wave tiles
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = "cuda"
shapes = [("s0.block", 64, 128, 128, True), ("s1.first", 64, 128, 256, False), ("s1.block", 32, 256, 256, True),
          ("s2.first", 32, 256, 256, False), ("s2.block", 16, 256, 256, True)]
if os.environ.get("VPT_BENCH_SHAPES"):   # "name,hw,cin,cout,res;..." overrides the default list
    shapes = [(n, int(h), int(ci), int(co), r == "1") for n, h, ci, co, r in (x.split(",") for x in os.environ["VPT_BENCH_SHAPES"].split(";"))]
g = torch.Generator(device="cpu").manual_seed(0)
for name, hw, cin, cout, use_res in shapes:
    f = frames * (64 * 64) // (hw * hw) if hw < 64 else frames
    f = min(f, frames * 4)
    W = torch.randn(cout, cin, 3, 3, generator=g) * (1.6 / (cin * 9) ** 0.5)
    wpk, sa, sg = packing.pack_conv3x3(W.to(dev), torch.ones(cin, device=dev), torch.zeros(cin, device=dev), dtype=DT)
    x = torch.relu(torch.randn(f, cin // 32, hw, hw, 32, device=dev)).to(DT)
    res = torch.randn(f, cout // 32, hw, hw, 32, device=dev).to(DT) if use_res else None
    if os.environ.get("VPT_BENCH_ZERO") == "1":  # DVFS probe: same instruction stream, no operand toggling (MI355X_MICROARCH.md "DVFS give-back")
        x.zero_(); wpk.zero_()
        if res is not None:
            res.zero_()
    xf = x.float().reshape(f, -1).double()
    st_in = torch.stack([xf.sum(1), (xf * xf).sum(1)], 1).contiguous()
    st_out = None if os.environ.get("VPT_BENCH_NOSTATS") == "1" else torch.zeros(f, 2, dtype=torch.float64, device=dev)
    out = torch.empty(f, cout // 32, hw, hw, 32, dtype=DT, device=dev)
    for _ in range(2):
        ops.conv3x3(x, wpk, sa, sg, st_in, cout, res=res, stats_out=st_out, out=out, tiling=TILING)
    torch.cuda.synchronize()
    times = []
    for _ in range(5):  # 5 rounds of `reps` launches; report median and best round
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            ops.conv3x3(x, wpk, sa, sg, st_in, cout, res=res, stats_out=st_out, out=out, tiling=TILING)
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b) / reps)
    times.sort()
    ms, best = times[2], times[0]
    flops = 2.0 * f * hw * hw * cout * 9 * cin
    tag = " ".join(f"{k[9:].lower()}={v}" for k, v in os.environ.items() if k.startswith("VPT_CONV_"))
    if name.endswith(".first") and os.environ.get("VPT_BENCH_POOL", "1") == "1":
        # the stack's firstconv -> max-pool pair: two kernels (the pre-pool tensor through HBM) vs the pool-fused convolution + seam kernel
        stp = torch.zeros(f, 2, dtype=torch.float64, device=dev)
        pooled = torch.empty(f, cout // 32, hw // 2, hw // 2, 32, dtype=DT, device=dev)

        def pair():
            ops.conv3x3(x, wpk, sa, sg, st_in, cout, out=out)
            ops.maxpool(out, stats_out=stp, out=pooled)

        def fused():
            ops.conv3x3_pool(x, wpk, sa, sg, st_in, cout, stats_out=stp, out=pooled)

        def fused_masks():      # the training forward (round 5): the same pass + the arg-max masks (vpt_conv3x3_kernel mode 7)
            ops.conv3x3_pool_argmax(x, wpk, sa, sg, st_in, cout, stats_out=stp)

        for label, fn in (("conv+pool", pair), ("fused", fused), ("fused+masks", fused_masks), ("conv+pool", pair), ("fused", fused), ("fused+masks", fused_masks)):
            fn(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            t = a.elapsed_time(b) / reps
            print(f"[{os.environ.get('VPT_PRECISION', 'bf16')}] {name:9s} {label:11s}: {t:7.3f} ms per layer incl. pool ({flops / t / 1e9:7.1f} TF/s conv-equivalent)")
    print(f"[{os.environ.get('VPT_PRECISION', 'bf16')}] {name:9s} frames={f:5d} {hw}x{hw} {cin}->{cout}: median {ms:7.3f} ms {flops / ms / 1e9:7.1f} TF/s | best {flops / best / 1e9:7.1f} TF/s  {tag}")

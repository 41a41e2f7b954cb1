"""Bit-identity check ACROSS library builds: sha256 of the outputs of the residual convolution modes on fixed seeded inputs.

    python tools/conv_hash.py > a.txt;  VPT_HIP_LIB=$PWD/video-pre-training_amd/build/libvpt_ref.so python tools/conv_hash.py > b.txt;  diff a.txt b.txt

Covers vpt_conv3x3_kernel modes 0 (no residual), 1 (residual) and 5 (per-frame affine residual, folded table) on the policy's layer shapes and
on ragged ones, including inputs with exact zeros and negative pre-activations everywhere (the ReLU / clamp path), values near the 16-bit
formats' extremes, and the frame statistics each launch accumulates.  Needs an MI355X."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import vpt_amd  # noqa: F401,E402
from vpt_amd import ops  # noqa: E402

DEV = "cuda"


def h(*ts):
    m = hashlib.sha256()
    for t in ts:
        m.update(t.detach().contiguous().view(torch.uint8).cpu().numpy().tobytes())
    return m.hexdigest()[:16]


def blocked(x):
    f, c, hh, ww = x.shape
    return x.view(f, c // 32, 32, hh, ww).permute(0, 1, 3, 4, 2).contiguous().to(torch.bfloat16)


def stats_of(xb):
    xf = xb.float().double()
    return torch.stack([xf.sum(dim=(1, 2, 3, 4)), (xf * xf).sum(dim=(1, 2, 3, 4))], 1).contiguous()


def main():
    torch.manual_seed(0)
    # (frames, Cin, Cout, H, W, scale of the input and the residual, scale of the weights = of the convolution's output)
    cases = [(3, 128, 128, 64, 64, 1.0, 1.0), (2, 256, 256, 32, 32, 1.0, 1.0), (2, 256, 256, 16, 16, 1.0, 1.0), (2, 64, 96, 16, 48, 1.0, 1.0), (1, 32, 160, 48, 32, 1.0, 1.0),
             (2, 128, 128, 32, 32, 1e4, 1.0), (2, 128, 128, 32, 32, 1e-4, 1.0), (2, 128, 128, 32, 32, 1.0, 1e6), (2, 128, 128, 32, 32, 1e-18, 1e-20)]
    for f, cin, cout, hh, ww, scale, wscale in cases:
        g = torch.Generator().manual_seed(f * 1000 + cin + cout + hh)
        x = blocked((torch.randn(f, cin, hh, ww, generator=g) * scale).to(DEV))
        res = blocked((torch.randn(f, cout, hh, ww, generator=g) * scale).to(DEV))
        res.view(-1)[::7] = 0                                                    # exact zeros in the residual
        w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5 * wscale).to(DEV)
        gain = (1 + 0.2 * torch.randn(cin, generator=g)).to(DEV)
        bias = (0.1 * torch.randn(cin, generator=g) - 0.2).to(DEV)       # (x's own scale drops out in the GroupNorm; `scale` sizes the residual against the conv output)
        wpk, sa, sg = ops.pack_conv3x3(w, gain, bias)
        st = stats_of(x)
        tag = f"{f}x{cin}->{cout} {hh}x{ww} s={scale:g} w={wscale:g}"
        for tiling in ("throughput", "latency"):
            so0 = torch.zeros(f, 2, dtype=torch.float64, device=DEV)
            y0 = ops.conv3x3(x, wpk, sa, sg, st, cout, stats_out=so0, tiling=tiling)
            so1 = torch.zeros(f, 2, dtype=torch.float64, device=DEV)
            y1 = ops.conv3x3(x, wpk, sa, sg, st, cout, res=res, stats_out=so1, tiling=tiling)
            torch.cuda.synchronize()
            neg = float((y1.float() < res.float()).float().mean())               # must be 0: ReLU(conv) >= 0 was added
            print(f"{tag} {tiling}: mode0 {h(y0)} stats {h(so0)}  mode1 {h(y1)} stats {h(so1)}  out<res {neg:.3f} finite {bool(torch.isfinite(y1.float()).all())}")
        # mode 5: per-frame table + affine residual (the folded GroupNorm `n` path: conv3x3_folded)
        kk = (sa.unsqueeze(0) + 0.05 * wscale * torch.randn(f, 9, sa.shape[1], generator=g).to(DEV)).contiguous()
        rs = (1 + 0.1 * torch.rand(f, generator=g)).to(DEV)
        rsc = (0.5 + torch.rand(f, generator=g)).to(DEV)
        rb = (0.3 * scale * torch.randn(f, cout, generator=g)).to(DEV)
        so5 = torch.zeros(f, 2, dtype=torch.float64, device=DEV)
        y5 = ops.conv3x3_folded(x, wpk, sa, sg, st, cout, kk_frame=kk, rs_frame=rs, res=res, res_scale=rsc, res_bias=rb, stats_out=so5)
        so5b = torch.zeros(f, 2, dtype=torch.float64, device=DEV)
        y5b = ops.conv3x3_folded(x, wpk, sa, sg, st, cout, res=res, res_scale=rsc, res_bias=rb, stats_out=so5b)
        torch.cuda.synchronize()
        print(f"{tag} mode5: table {h(y5)} stats {h(so5)}  own-stats {h(y5b)} stats {h(so5b)}")


if __name__ == "__main__":
    main()

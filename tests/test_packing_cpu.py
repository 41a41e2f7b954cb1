"""Host logic (weight re-packing + GroupNorm fold algebra) checked on CPU against the oracle.
These emulate, in torch fp32, exactly what the kernels compute from the packed tensors, so a packing or
fold-algebra bug is caught without a GPU."""
import torch
import torch.nn.functional as F

import vpt_amd  # noqa: F401
from vpt_amd import packing
from oracle import vpt_oracle as O


def _edge_class_map(h, w):
    ey = torch.ones(h, dtype=torch.long); ey[0] = 0; ey[-1] = 2
    ex = torch.ones(w, dtype=torch.long); ex[0] = 0; ex[-1] = 2
    return ey.view(h, 1) * 3 + ex.view(1, w)


def test_conv3x3_pack_layout_and_fold():
    g = torch.Generator().manual_seed(3)
    cin, cout, h, w = 64, 160, 16, 16  # cout not a multiple of 128 -> padded tile
    W = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    gain = 1 + 0.2 * torch.randn(cin, generator=g)
    bias = 0.1 * torch.randn(cin, generator=g)
    x = torch.relu(torch.randn(2, cin, h, w, generator=g)) + 0.3
    wpk, sa, sg = packing.pack_conv3x3(W, gain, bias)
    nt = 2
    assert wpk.shape == (nt, cin // 32, 9, 128, 32) and sa.shape == (9, 256) and sg.shape == (9, 256)
    # layout: wpk[nt, cb, tap, n, ci] == bf16(W[nt*128+n, cb*32+ci, tap//3, tap%3] * gain)
    wg = (W * gain.view(1, -1, 1, 1)).to(torch.bfloat16)
    # undo the LDS-image swizzle (chunk c of row n lives at c ^ ((n >> 2) & 3)); it is an involution
    assert not torch.equal(packing.swizzle_rows64(wpk), wpk)
    rec = packing.swizzle_rows64(wpk).permute(0, 3, 1, 4, 2).reshape(nt * 128, cin, 3, 3)
    assert torch.equal(rec[:cout], wg) and rec[cout:].abs().max() == 0
    # emulate the kernel: raw bf16 activations through conv(W*g), then the epilogue fold
    xb = x.to(torch.bfloat16).float()
    mu = xb.mean(dim=(1, 2, 3)); var = xb.var(dim=(1, 2, 3), unbiased=False)
    rstd = torch.rsqrt(var + 1e-5)
    acc = F.conv2d(xb, rec[:cout].float(), padding=1)
    e = _edge_class_map(h, w)
    sa_map = sa[:, :cout][e].permute(2, 0, 1)  # [cout,h,w]
    sg_map = sg[:, :cout][e].permute(2, 0, 1)
    out = rstd.view(-1, 1, 1, 1) * acc + sa_map - (rstd * mu).view(-1, 1, 1, 1) * sg_map
    out = torch.relu(out)
    sd = {"norm.weight": gain, "norm.bias": bias, "layer.weight": W}
    ref = O._norm_conv_relu(sd, "", xb)
    err = (out - ref).abs().max() / ref.abs().max()
    assert err < 1e-2, err  # only the bf16 rounding of W*g separates the two


def test_conv_first_pack():
    g = torch.Generator().manual_seed(4)
    cout = 64
    W = torch.randn(cout, 3, 3, 3, generator=g) * 0.2
    b = 0.1 * torch.randn(cout, generator=g)
    frag = packing.pack_conv_first(W, b)
    assert frag.shape == (1, 4, 2, 64, 8)
    # undo the fragment order: lane -> (row = l&31, slot half = l>>5), slot -> k through packing.CONV_FIRST_SLOT_K
    wk = torch.zeros(128, 32)
    assert sorted(packing.CONV_FIRST_SLOT_K) == list(range(32))
    for cs in range(4):
        for ks in range(2):
            for lane in range(64):
                o = cs * 32 + (lane & 31)
                s0 = ks * 16 + (lane >> 5) * 8
                for e in range(8):
                    wk[o, packing.CONV_FIRST_SLOT_K[s0 + e]] = frag[0, cs, ks, lane, e].float()
    assert wk[cout:].abs().max() == 0
    img = torch.randint(0, 256, (1, 8, 8, 3), generator=g, dtype=torch.uint8)
    # emulate: acc = sum_k wk[o,k] * pix[k] with wk = W / 255 and pix the raw bytes, pix[27] = pix[28] = 1 (bias hi / lo), then relu
    xpad = F.pad(img.float().permute(0, 3, 1, 2), (1, 1, 1, 1))
    cols = F.unfold(xpad, 3).view(1, 3, 9, 64).permute(0, 2, 1, 3).reshape(1, 27, 64)  # k = tap*3 + ch
    pix = torch.cat([cols, torch.ones(1, 2, 64), torch.zeros(1, 3, 64)], dim=1)
    out = torch.relu(torch.einsum("ok,bkp->bop", wk[:cout], pix))
    ref = torch.relu(F.conv2d(img.float().permute(0, 3, 1, 2) / 255.0, W, b, padding=1)).reshape(1, cout, 64)
    assert (out - ref).abs().max() < 2e-2 * ref.abs().max()


def test_linear_pack_and_blocked_permutation():
    g = torch.Generator().manual_seed(5)
    W = torch.randn(200, 128, generator=g)
    wpk = packing.pack_linear(W)
    assert wpk.shape == (2, 4, 128, 32)
    rec = wpk.permute(0, 2, 1, 3).reshape(256, 128)
    assert torch.equal(rec[:200], W.to(torch.bfloat16)) and rec[200:].abs().max() == 0
    c, h, w = 64, 16, 16
    x = torch.randn(3, c, h, w, generator=g)
    Wd = torch.randn(8, c * h * w, generator=g)
    ref = x.reshape(3, -1) @ Wd.t()
    xb = packing.nchw_to_blocked(x, torch.float32).reshape(3, -1)
    out = xb @ packing.chw_to_blocked_columns(Wd, c, h, w).t()
    assert torch.allclose(out, ref, atol=1e-3)
    v = torch.randn(c * h * w, generator=g)
    vb = packing.chw_to_blocked_vector(v, c, h, w)
    assert torch.allclose((xb * vb).sum(1), (x.reshape(3, -1) * v).sum(1), atol=1e-2)
    assert torch.equal(packing.blocked_to_nchw(packing.nchw_to_blocked(x), c, h, w), x.to(torch.bfloat16).float())

"""Per-kernel parity of the HIP path (through the C ABI) against the CPU oracle.  Needs an MI355X.

Tolerances: the MFMA kernels round their operands to bf16 (fp32 accumulate), so results are compared with
an oracle fed the SAME bf16-rounded activations; what remains is the bf16 rounding of the weights and of
the stored outputs: 2e-2 of the tensor's max magnitude (typical observed error is ~3e-3).  fp32 kernels
(layernorm, attention, log-softmax) are held to 1e-4..2e-3 absolute as noted per test."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import vpt_amd  # noqa: E402,F401
from vpt_amd import ops, packing  # noqa: E402
from oracle import vpt_oracle as O  # noqa: E402

DEV = "cuda"


def _relerr(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp(min=1e-12)).item()


def _stats_of(x_nchw):
    f = x_nchw.shape[0]
    flat = x_nchw.reshape(f, -1).double()
    return torch.stack([flat.sum(1), (flat * flat).sum(1)], dim=1).contiguous()


@pytest.mark.parametrize("frames,h,w,cin,cout,use_res", [
    (2, 16, 16, 128, 128, True),
    (1, 32, 32, 64, 160, False),
    (3, 64, 64, 128, 256, False),
    (2, 32, 32, 256, 256, True),
    (3, 64, 64, 64, 64, True),      # 1x stack-0 block: two channel blocks only, residual
    (2, 64, 64, 128, 128, True),    # 2x stack-0 block
    (2, 32, 32, 128, 128, True),
    (5, 16, 16, 256, 256, True),
    (2, 64, 64, 64, 128, False),    # 1x stack-1 firstconv
    (2, 16, 16, 32, 64, True),      # a single 32-channel block: no peeled first block, the last one starts from zeroed accumulators
])
@pytest.mark.parametrize("tiling", ["throughput", "latency", "throughput32"])   # the workgroup tilings of the same convolution (vpt_conv3x3_forward_tiled)
def test_conv3x3(frames, h, w, cin, cout, use_res, tiling):
    g = torch.Generator().manual_seed(1)
    W = torch.randn(cout, cin, 3, 3, generator=g) * (1.6 / (cin * 9) ** 0.5)
    gain = 1 + 0.2 * torch.randn(cin, generator=g)
    bias = 0.1 * torch.randn(cin, generator=g)
    x = (torch.relu(torch.randn(frames, cin, h, w, generator=g)) + 0.2 * torch.randn(frames, cin, h, w, generator=g))
    xb = x.to(torch.bfloat16)
    res = torch.randn(frames, cout, h, w, generator=g).to(torch.bfloat16) if use_res else None
    sd = {"norm.weight": gain, "norm.bias": bias, "layer.weight": W}
    ref = O._norm_conv_relu(sd, "", xb.float())
    if use_res:
        ref = ref + res.float()
    wpk, sa, sg = packing.pack_conv3x3(W.to(DEV), gain.to(DEV), bias.to(DEV))
    st_in = _stats_of(xb.float()).to(DEV)
    st_out = torch.zeros(frames, 2, dtype=torch.float64, device=DEV)
    y = ops.conv3x3(packing.nchw_to_blocked(xb.float()).to(DEV), wpk, sa, sg, st_in, cout,
                    res=packing.nchw_to_blocked(res.float()).to(DEV) if use_res else None, stats_out=st_out, tiling=tiling)
    torch.cuda.synchronize()
    out = packing.blocked_to_nchw(y.cpu(), cout, h, w)
    err = _relerr(out, ref)
    assert err < 2e-2, f"conv3x3 rel err {err}"
    if tiling == "latency":     # same arithmetic and K order as the throughput kernel: equal up to the last 16-bit rounding of a few outputs
        y2 = ops.conv3x3(packing.nchw_to_blocked(xb.float()).to(DEV), wpk, sa, sg, st_in, cout,
                         res=packing.nchw_to_blocked(res.float()).to(DEV) if use_res else None, tiling="throughput")
        torch.cuda.synchronize()
        assert _relerr(y.float().cpu(), y2.float().cpu()) < 1e-3
    if tiling == "throughput32":   # 32-row / eight-wave tiles (where the image has whole 32-row bands) vs the shipped 16-row tiles: the same per-pixel program
        y2 = ops.conv3x3(packing.nchw_to_blocked(xb.float()).to(DEV), wpk, sa, sg, st_in, cout,
                         res=packing.nchw_to_blocked(res.float()).to(DEV) if use_res else None, tiling="throughput")
        torch.cuda.synchronize()
        assert torch.equal(y.view(torch.int16), y2.view(torch.int16))
    st_ref = _stats_of(ref)
    assert torch.allclose(st_out.cpu(), st_ref, rtol=5e-3, atol=1.0), (st_out.cpu(), st_ref)


@pytest.mark.parametrize("frames,h,w,cin,cout", [
    (3, 64, 64, 128, 256),     # 2x stack-1 firstconv: 4 x 4 tiles, two channel tiles
    (2, 32, 32, 256, 256),     # 2x stack-2 firstconv: 2 x 2 tiles
    (2, 64, 64, 64, 128),      # 1x stack-1 firstconv
    (5, 16, 16, 64, 96),       # one tile per frame: no seams at all; Cout / 32 odd (a wave with one valid channel block)
    (2, 48, 16, 32, 160),      # tiles in one direction only (row seams, no column seams), a channel tile with a single valid block
    (3, 16, 80, 96, 64),       # column seams only
    (1, 128, 128, 128, 128),   # the IDM's stack-0 firstconv shape: 8 x 8 tiles
])
@pytest.mark.parametrize("fmt", ["bf16", "fp16"])
def test_conv3x3_pool_fused(frames, h, w, cin, cout, fmt):
    """firstconv + ReLU + max_pool2d(3, 2, 1) in one pass (vpt_conv3x3_pool_forward: LDS-pooled tiles + the seam kernel) == the
    two-kernel path vpt_conv3x3_forward -> vpt_maxpool_forward BIT FOR BIT (same MFMA order, same rounding point, the maximum is
    exact), its statistics to the order of their fp64 additions; and close to CnnDownStack's first two lines in fp32
    (lib/impala_cnn.py:114-117) like every conv test."""
    dt = torch.bfloat16 if fmt == "bf16" else torch.float16
    g = torch.Generator().manual_seed(17)
    W = torch.randn(cout, cin, 3, 3, generator=g) * (1.6 / (cin * 9) ** 0.5)
    gain = 1 + 0.2 * torch.randn(cin, generator=g)
    bias = 0.1 * torch.randn(cin, generator=g)
    x = (torch.relu(torch.randn(frames, cin, h, w, generator=g)) + 0.2 * torch.randn(frames, cin, h, w, generator=g)).to(dt)
    wpk, sa, sg = ops.pack_conv3x3(W.to(DEV), gain.to(DEV), bias.to(DEV), dtype=dt)
    xb = packing.nchw_to_blocked(x.float(), dtype=dt).to(DEV)
    st_in = _stats_of(x.float()).to(DEV)
    pre = ops.conv3x3(xb, wpk, sa, sg, st_in, cout)
    st_a = torch.zeros(frames, 2, dtype=torch.float64, device=DEV)
    want = ops.maxpool(pre, stats_out=st_a)
    st_b = torch.zeros(frames, 2, dtype=torch.float64, device=DEV)
    got = ops.conv3x3_pool(xb, wpk, sa, sg, st_in, cout, stats_out=st_b)
    torch.cuda.synchronize()
    assert got.shape == want.shape == (frames, cout // 32, h // 2, w // 2, 32)
    neq = (got.view(torch.int16) != want.view(torch.int16))
    assert not bool(neq.any()), f"{int(neq.sum())} of {neq.numel()} pooled values differ; first at {neq.nonzero()[0].tolist()}"
    assert torch.allclose(st_a, st_b, rtol=1e-6, atol=1e-3), (st_a, st_b)     # fp32 partial sums in a different order, fp64 across tiles
    ref = F.max_pool2d(O._norm_conv_relu({"norm.weight": gain, "norm.bias": bias, "layer.weight": W}, "", x.float()), 3, 2, 1)
    assert _relerr(packing.blocked_to_nchw(got.cpu(), cout, h // 2, w // 2), ref) < 2e-2
    # out_gain: the pooled tensor stored times a per-channel gain (GroupNorm `n` folded), statistics still those of the unscaled tensor
    gain = (1 + 0.3 * torch.randn(cout, generator=g)).to(DEV)
    gain[1] = -0.5
    st_c = torch.zeros(frames, 2, dtype=torch.float64, device=DEV)
    scaled = ops.conv3x3_pool(xb, wpk, sa, sg, st_in, cout, stats_out=st_c, out_gain=gain)
    torch.cuda.synchronize()
    want_scaled = (want.float() * gain.view(1, cout // 32, 1, 1, 32)).to(dt)
    assert torch.equal(scaled.view(torch.int16), want_scaled.view(torch.int16))
    assert torch.allclose(st_c, st_b, rtol=1e-9, atol=1e-6)
    # chs_out: the per-channel sums of the stored tensor from the two launches themselves == a pass over the result
    chs = torch.zeros(frames, cout, 2, dtype=torch.float64, device=DEV)
    scaled2 = ops.conv3x3_pool(xb, wpk, sa, sg, st_in, cout, out_gain=gain, chs_out=chs)
    torch.cuda.synchronize()
    assert torch.equal(scaled2.view(torch.int16), want_scaled.view(torch.int16))
    assert torch.allclose(chs, ops.channel_stats(scaled2), rtol=1e-6, atol=1e-4), (chs - ops.channel_stats(scaled2)).abs().max()
    # into a slice of a larger batch's tensor (the engine's sub-chunk use)
    big = torch.zeros(frames + 2, cout // 32, h // 2, w // 2, 32, dtype=dt, device=DEV)
    ops.conv3x3_pool(xb, wpk, sa, sg, st_in, cout, out=big[1:1 + frames])
    torch.cuda.synchronize()
    assert torch.equal(big[1:1 + frames].view(torch.int16), want.view(torch.int16)) and not bool(big[0].any()) and not bool(big[-1].any())


@pytest.mark.parametrize("frames,h,w,cin,cout", [(2, 64, 64, 64, 96), (3, 32, 32, 32, 256), (1, 16, 16, 64, 32), (2, 48, 32, 32, 160), (3, 16, 80, 96, 64), (1, 128, 128, 32, 128)])
@pytest.mark.parametrize("fmt", ["bf16", "fp16"])
def test_conv3x3_pool_argmax_masks(frames, h, w, cin, cout, fmt):
    """Round 5, the TRAINING forward of firstconv -> max_pool2d (lib/impala_cnn.py:114-117): vpt_conv3x3_pool_argmax_forward gives the pooled
    tensor of the two-kernel path BIT FOR BIT plus, per pooled value, the mask of window positions that hold the maximum.  Decoded with
    torch's rule (first maximum in scan order = highest zero bit) the masks equal vpt_maxpool_forward's arg-max bytes wherever a gradient can
    flow (pooled > 0; an all-zero window passes none either way), on tiles with row seams, column seams, both and none; and a bit is 0 exactly
    where the pre-pool value equals the maximum (ties included), 1 outside the image."""
    dt = torch.bfloat16 if fmt == "bf16" else torch.float16
    g = torch.Generator().manual_seed(171)
    W = torch.randn(cout, cin, 3, 3, generator=g) * (1.6 / (cin * 9) ** 0.5)
    gain = 1 + 0.2 * torch.randn(cin, generator=g)
    bias = 0.1 * torch.randn(cin, generator=g) - 0.6          # (shifted: a good share of exact zeros after the ReLU -> ties and all-zero windows)
    x = (torch.relu(torch.randn(frames, cin, h, w, generator=g)) + 0.2 * torch.randn(frames, cin, h, w, generator=g)).to(dt)
    wpk, sa, sg = ops.pack_conv3x3(W.to(DEV), gain.to(DEV), bias.to(DEV), dtype=dt)
    xb = packing.nchw_to_blocked(x.float(), dtype=dt).to(DEV)
    st_in = _stats_of(x.float()).to(DEV)
    pre = ops.conv3x3(xb, wpk, sa, sg, st_in, cout)
    st_a = torch.zeros(frames, 2, dtype=torch.float64, device=DEV)
    want, am = ops.maxpool(pre, stats_out=st_a, want_argmax=True)
    st_b = torch.zeros(frames, 2, dtype=torch.float64, device=DEV)
    got, mask = ops.conv3x3_pool_argmax(xb, wpk, sa, sg, st_in, cout, stats_out=st_b)
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert torch.allclose(st_a, st_b, rtol=1e-6, atol=1e-3)
    m = mask.to(torch.int32) & 0xffff
    assert int(m.max()) <= 0x1ff
    inv = (~m) & 0x1ff
    assert bool((inv != 0).all())                                   # every window has a position that holds its maximum
    code = 8 - torch.floor(torch.log2(inv.float())).to(torch.int32)   # first (scan order) zero bit: bit 8 - k for position k
    live = want.float() > 0
    assert float(live.float().mean()) > 0.2 and float((~live).float().mean()) > 0.01, float(live.float().mean())
    assert torch.equal(code[live], am.to(torch.int32)[live]), f"{int((code[live] != am.to(torch.int32)[live]).sum())} arg-max positions differ"
    assert bool((am[~live] == 15).all())
    # every bit against the oracle's restatement of the mask format (oracle/pool_mask.py, pinned on the CPU against torch's max_pool2d indices incl.
    # ties: tests/test_backward_algebra_cpu.py) applied to the GPU's own pre-pool tensor: 0 <=> inside the image and equal to the window maximum
    from oracle import pool_mask
    pre_n = packing.blocked_to_nchw(pre.cpu(), cout, h, w).double().numpy()
    pooled_ref, masks_ref = pool_mask.pool_argmax_masks(pre_n)
    mask_n = packing.blocked_to_nchw(m.cpu().to(torch.float32), cout, h // 2, w // 2).to(torch.int32).numpy()
    assert np.array_equal(pooled_ref, packing.blocked_to_nchw(want.cpu(), cout, h // 2, w // 2).double().numpy())
    assert np.array_equal(mask_n, masks_ref.astype(np.int32)), f"{int((mask_n != masks_ref).sum())} masks differ"


@pytest.mark.parametrize("frames,cout,h,w", [(2, 128, 128, 128), (1, 64, 128, 128), (1, 192, 128, 128), (5, 64, 32, 80), (3, 128, 48, 16)])
def test_conv_first_pool(frames, cout, h, w):
    """(non-square frames: the persistent workgroups count tile coordinates up -- tile column, tile row, frame -- instead of decoding them)"""
    g = torch.Generator().manual_seed(2)
    W = torch.randn(cout, 3, 3, 3, generator=g) * 0.3
    b = 0.1 * torch.randn(cout, generator=g)
    img = torch.randint(0, 256, (frames, h, w, 3), generator=g, dtype=torch.uint8)
    ref = F.max_pool2d(torch.relu(F.conv2d(img.permute(0, 3, 1, 2).float() / 255.0, W, b, padding=1)), 3, 2, 1)
    st = torch.zeros(frames, 2, dtype=torch.float64, device=DEV)
    y = ops.conv_first(img.to(DEV), packing.pack_conv_first(W.to(DEV), b.to(DEV)), cout, stats_out=st)
    torch.cuda.synchronize()
    out = packing.blocked_to_nchw(y.cpu(), cout, h // 2, w // 2)
    err = _relerr(out, ref)
    assert err < 1.5e-2, f"conv_first rel err {err}"
    assert torch.allclose(st.cpu(), _stats_of(out), rtol=1e-4, atol=1e-2)
    gain = (1 + 0.3 * torch.randn(cout, generator=g)).to(DEV)       # out_gain: stored times a per-channel gain, statistics of the unscaled tensor
    st2 = torch.zeros(frames, 2, dtype=torch.float64, device=DEV)
    y2 = ops.conv_first(img.to(DEV), packing.pack_conv_first(W.to(DEV), b.to(DEV)), cout, stats_out=st2, out_gain=gain)
    torch.cuda.synchronize()
    assert torch.equal(y2.view(torch.int16), (y.float() * gain.view(1, cout // 32, 1, 1, 32)).to(torch.bfloat16).view(torch.int16))
    assert torch.allclose(st2, st, rtol=1e-9, atol=1e-6)
    if cout <= 128:        # chs_out: per-channel sums of the stored tensor, accumulated in registers across a frame's tiles
        chs = torch.zeros(frames, cout, 2, dtype=torch.float64, device=DEV)
        y3 = ops.conv_first(img.to(DEV), packing.pack_conv_first(W.to(DEV), b.to(DEV)), cout, out_gain=gain, chs_out=chs)
        torch.cuda.synchronize()
        assert torch.equal(y3.view(torch.int16), y2.view(torch.int16))
        assert torch.allclose(chs, ops.channel_stats(y3), rtol=1e-6, atol=1e-4), (chs - ops.channel_stats(y3)).abs().max()


def test_conv_first_does_not_depend_on_how_the_batch_is_cut():
    """The persistent workgroups of vpt_conv_first_kernel take contiguous tile ranges that depend on the number of frames in the launch; a frame's pooled tensor,
    its statistics and its per-channel sums must not (fp32 sums only inside a tile / a 16-tile group, fp64 across).  600 frames = three groups per
    workgroup, ranges that cross frame boundaries; against the same frames in six launches of 100 and in single-frame launches: every bit equal."""
    g = torch.Generator().manual_seed(11)
    cout, frames = 128, 600
    W = torch.randn(cout, 3, 3, 3, generator=g) * 0.3
    b = 0.1 * torch.randn(cout, generator=g)
    wf = packing.pack_conv_first(W.to(DEV), b.to(DEV))
    gain = (1 + 0.3 * torch.randn(cout, generator=g)).to(DEV)
    img = torch.randint(0, 256, (frames, 128, 128, 3), generator=g, dtype=torch.uint8).to(DEV)

    def run(lo, hi, with_chs):
        st = torch.zeros(hi - lo, 2, dtype=torch.float64, device=DEV)
        chs = torch.zeros(hi - lo, cout, 2, dtype=torch.float64, device=DEV) if with_chs else None
        y = ops.conv_first(img[lo:hi], wf, cout, stats_out=st, out_gain=gain if with_chs else None, chs_out=chs)
        return y, st, chs
    for with_chs in (True, False):
        y, st, chs = run(0, frames, with_chs)
        for lo in range(0, frames, 100):
            y2, st2, chs2 = run(lo, lo + 100, with_chs)
            assert torch.equal(y2.view(torch.int16), y[lo:lo + 100].view(torch.int16)) and torch.equal(st2, st[lo:lo + 100])
            assert not with_chs or torch.equal(chs2, chs[lo:lo + 100])
        for f in (0, 257, frames - 1):
            y1, st1, chs1 = run(f, f + 1, with_chs)
            assert torch.equal(y1.view(torch.int16), y[f:f + 1].view(torch.int16)) and torch.equal(st1, st[f:f + 1])
            assert not with_chs or torch.equal(chs1, chs[f:f + 1])


def test_maxpool_and_affine():
    g = torch.Generator().manual_seed(3)
    x = torch.relu(torch.randn(2, 64, 32, 32, generator=g)).to(torch.bfloat16)
    st = torch.zeros(2, 2, dtype=torch.float64, device=DEV)
    y = ops.maxpool(packing.nchw_to_blocked(x.float()).to(DEV), stats_out=st)
    torch.cuda.synchronize()
    ref = F.max_pool2d(x.float(), 3, 2, 1)
    out = packing.blocked_to_nchw(y.cpu(), 64, 16, 16)
    assert torch.equal(out, ref)
    assert torch.allclose(st.cpu(), _stats_of(ref), rtol=1e-5, atol=1e-3)
    gain = 1 + 0.2 * torch.randn(64, generator=g)
    bias = 0.1 * torch.randn(64, generator=g)
    st2 = torch.zeros(2, 2, dtype=torch.float64, device=DEV)
    z = ops.frame_affine(y, gain.to(DEV), bias.to(DEV), st, stats_out=st2)
    torch.cuda.synchronize()
    refz = O.group_norm_1(ref, gain, bias)
    outz = packing.blocked_to_nchw(z.cpu(), 64, 16, 16)
    assert _relerr(outz, refz) < 8e-3
    assert torch.allclose(st2.cpu(), _stats_of(refz), rtol=1e-3, atol=0.5)
    # per-element affine (the 65536-wide LayerNorm of ImpalaCNN.dense) in blocked order
    ge = 1 + 0.2 * torch.randn(64 * 256, generator=g)
    be = 0.1 * torch.randn(64 * 256, generator=g)
    z2 = ops.frame_affine(y, packing.chw_to_blocked_vector(ge, 64, 16, 16).to(DEV),
                          packing.chw_to_blocked_vector(be, 64, 16, 16).to(DEV), st, per_element=True)
    torch.cuda.synchronize()
    ref2 = O.layer_norm(ref.reshape(2, -1), ge, be).reshape(2, 64, 16, 16)
    assert _relerr(packing.blocked_to_nchw(z2.cpu(), 64, 16, 16), ref2) < 8e-3


@pytest.mark.parametrize("m,n,k,bias,relu,res,splitk", [
    (1, 121, 256, True, False, False, 1),
    (300, 8763, 2048, True, False, False, 1),
    (513, 2048, 8192, True, False, True, 1),
    (256, 4096, 1024, False, True, False, 1),
    (40, 256, 16384, False, False, False, 16),
    # 8 < M <= 512 with few N tiles: automatic split-K + epilogue kernel (one IDM window)
    (128, 4096, 4096, True, True, True, 1),
    (128, 640, 16384, True, False, False, 1),
    (300, 1000, 2048, False, True, False, 1),
    # M <= 8: the acting path's weight-streaming kernel (vpt_gemv.hip)
    (1, 8763, 2048, True, False, False, 1),
    (2, 2048, 8192, True, False, True, 1),
    (3, 4096, 1024, False, True, False, 1),
    (8, 300, 512, True, True, True, 1),
    (1, 256, 65536, False, False, False, 16),
    (5, 256, 16384, False, False, False, 7),
])
def test_linear(m, n, k, bias, relu, res, splitk):
    g = torch.Generator().manual_seed(4)
    A = torch.randn(m, k, generator=g).to(torch.bfloat16)
    W = torch.randn(n, k, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g) if bias else None
    r = torch.randn(m, n, generator=g) if res else None
    ref = A.float() @ W.to(torch.bfloat16).float().t()
    if bias:
        ref = ref + b
    if relu:
        ref = torch.relu(ref)
    if res:
        ref = ref + r
    o32, o16 = ops.linear(A.to(DEV), packing.pack_linear(W.to(DEV)), n, bias=b.to(DEV) if bias else None,
                          res=r.to(DEV) if res else None, relu=relu, out_f32=True, out_bf16=(splitk == 1), splitk=splitk)
    torch.cuda.synchronize()
    assert _relerr(o32.cpu(), ref) < 2e-3, _relerr(o32.cpu(), ref)
    if o16 is not None:
        assert _relerr(o16.cpu().float(), ref) < 1e-2


@pytest.mark.parametrize("bias,relu,res,mask,out_f32,out_bf16", [
    (True, False, False, False, True, False),     # qkvr / heads
    (True, False, True, False, True, False),      # proj / mlp1
    (False, True, False, False, False, True),     # mlp0
    (False, True, False, False, True, False),     # img linear / lastlayer (inference)
    (False, True, False, False, True, True),      # ... (training forward keeps both)
    (False, False, False, False, True, False),    # dgrad -> fp32
    (False, False, False, False, False, True),    # dense dgrad -> 16-bit
    (False, False, True, False, True, False),     # dgrad + skip
    (False, False, False, True, False, True),     # dgrad through a ReLU gate -> 16-bit
    (True, True, True, True, True, True),         # no dedicated instantiation: the generic epilogue
])
@pytest.mark.parametrize("m,n,k", [(300, 2048, 512), (777, 6304, 256), (256, 8764, 128)])
def test_linear_epilogue_variants(bias, relu, res, mask, out_f32, out_bf16, m, n, k):
    """Every compile-time epilogue of vpt_gemm_kernel (16-byte stores from the swapped-operand accumulator layout) and the generic one,
    on ragged M (rows beyond M are never written), an N with a partial last 128-tile and an N that is only a multiple of 4."""
    g = torch.Generator().manual_seed(40)
    A = torch.randn(m, k, generator=g).to(torch.bfloat16)
    W = torch.randn(n, k, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g) if bias else None
    r = torch.randn(m, n, generator=g) if res else None
    mk = (torch.randn(m, n, generator=g)).to(torch.bfloat16) if mask else None
    ref = A.float() @ W.to(torch.bfloat16).float().t()
    if bias:
        ref = ref + b
    if relu:
        ref = torch.relu(ref)
    if mask:
        ref = torch.where(mk.float() > 0, ref, torch.zeros_like(ref))
    if res:
        ref = ref + r
    o32, o16 = ops.linear(A.to(DEV), packing.pack_linear(W.to(DEV)), n, bias=b.to(DEV) if bias else None, res=r.to(DEV) if res else None,
                          relu=relu, mask=mk.to(DEV) if mask else None, out_f32=out_f32, out_bf16=out_bf16)
    torch.cuda.synchronize()
    if out_f32:
        assert tuple(o32.shape) == (m, n) and _relerr(o32.cpu(), ref) < 2e-3, _relerr(o32.cpu(), ref)
    if out_bf16:
        assert tuple(o16.shape) == (m, n) and _relerr(o16.cpu().float(), ref) < 1e-2


@pytest.mark.parametrize("m,n,k,variant", [
    (8192, 2048, 256, "bias_f32"),        # 32 x 8 tiles of 256 x 256: the trunk's proj shape (short K)
    (8100, 1604, 192, "relu_bf16"),       # ragged M, an odd number of 128-column tiles (the last 256-tile's second half does not exist), N % 128 != 0
    (4096, 3072, 128, "res_f32"),         # 16 x 12 = 192 tiles: the smallest grid that takes the 256 x 256 kernel; two k-steps
    (8192, 1664, 64, "mask_bf16"),        # one k-step: no DMA in the loop at all
])
def test_linear_256x256_tiles_equal_the_256x128_kernel(m, n, k, variant):
    """vpt_gemm256_kernel (eight waves, LDS-DMA operands, double-buffered) against vpt_gemm_kernel on the same call: the same K order and MFMA
    sequence per output element, so the outputs must be BIT-identical; and both against the fp32 reference."""
    g = torch.Generator().manual_seed(41)
    A = torch.randn(m, k, generator=g).to(torch.bfloat16)
    W = torch.randn(n, k, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g)
    r = torch.randn(m, n, generator=g)
    mk = torch.randn(m, n, generator=g).to(torch.bfloat16)
    kw = dict(bias_f32=dict(bias=b.to(DEV), out_f32=True, out_bf16=False), relu_bf16=dict(relu=True, out_f32=False, out_bf16=True),
              res_f32=dict(bias=b.to(DEV), res=r.to(DEV), out_f32=True, out_bf16=False), mask_bf16=dict(mask=mk.to(DEV), out_f32=False, out_bf16=True))[variant]
    wpk = packing.pack_linear(W.to(DEV))
    outs = {}
    for tiling in ("throughput256", "throughput"):
        o32, o16 = ops.linear(A.to(DEV), wpk, n, tiling=tiling, **kw)
        torch.cuda.synchronize()
        outs[tiling] = (o32 if o32 is not None else o16).cpu()
    ref = A.float() @ W.to(torch.bfloat16).float().t()
    if variant in ("bias_f32", "res_f32"):
        ref = ref + b
    if variant == "relu_bf16":
        ref = torch.relu(ref)
    if variant == "mask_bf16":
        ref = torch.where(mk.float() > 0, ref, torch.zeros_like(ref))
    if variant == "res_f32":
        ref = ref + r
    big, small = outs["throughput256"], outs["throughput"]
    assert tuple(big.shape) == (m, n)
    assert _relerr(big.float(), ref) < (2e-3 if big.dtype == torch.float32 else 1e-2)
    it = torch.int32 if big.dtype == torch.float32 else torch.int16
    assert torch.equal(big.view(it), small.view(it))


@pytest.mark.parametrize("m,d,relu_in", [(5, 256, True), (130, 2048, False), (7, 3072, False)])
def test_layernorm(m, d, relu_in):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(m, d, generator=g) * 2 + 0.5
    gain = 1 + 0.2 * torch.randn(d, generator=g)
    bias = 0.1 * torch.randn(d, generator=g)
    o32, o16 = ops.layernorm(x.to(DEV), gain.to(DEV), bias.to(DEV), relu_in=relu_in, out_f32=True, out_bf16=True)
    torch.cuda.synchronize()
    ref = O.layer_norm(torch.relu(x) if relu_in else x, gain, bias)
    assert (o32.cpu() - ref).abs().max() < 1e-4
    assert _relerr(o16.cpu().float(), ref) < 5e-3


@pytest.mark.parametrize("bsz,t,heads,first_flags", [(2, 128, 2, [False, True]), (3, 5, 2, [False, False, True]), (2, 1, 2, [False, False]), (1, 70, 2, [False]),
                                                    (32, 128, 16, [False] * 31 + [True])])   # the bench's grid shape: 2048 workgroups, several per CU (caught a missing barrier)
def test_masked_attention_and_kv_update(bsz, t, heads, first_flags):
    g = torch.Generator().manual_seed(6)
    hid, maxlen = heads * 128, 128
    ld = 3 * hid + 10 * heads
    qkvr = torch.randn(bsz * t, ld, generator=g)
    qkvr[:, :hid] *= 3.0  # spread the softmax
    kmem = torch.randn(bsz, maxlen, hid, generator=g)
    vmem = torch.randn(bsz, maxlen, hid, generator=g)
    state_mask = torch.rand(bsz, 1, maxlen, generator=g) > 0.3
    first_b = torch.tensor(first_flags)
    b_nd = 0.5 * torch.randn(10, maxlen, generator=g)
    # oracle on the same projections
    q = qkvr[:, :hid].reshape(bsz, t, heads, 128).permute(0, 2, 1, 3)
    k_full = torch.cat([kmem, qkvr[:, hid:2 * hid].reshape(bsz, t, hid)], 1)
    v_full = torch.cat([vmem, qkvr[:, 2 * hid:3 * hid].reshape(bsz, t, hid)], 1)
    kh = k_full.reshape(bsz, -1, heads, 128).permute(0, 2, 1, 3)
    vh = v_full.reshape(bsz, -1, heads, 128).permute(0, 2, 1, 3)
    logits = q @ kh.transpose(-1, -2) / 128.0
    vis, new_mask = O.band_visibility(t, maxlen, first_b, state_mask)
    logits = logits + (~vis).float().unsqueeze(1) * O.NEG_MASK
    logits = logits + O.rel_pos_bias(qkvr[:, 3 * hid:].reshape(bsz, t, heads, 10), b_nd, t, maxlen)
    ref = (torch.softmax(logits, -1) @ vh).permute(0, 2, 1, 3).reshape(bsz * t, hid)
    memvalid = (state_mask & ~first_b.view(bsz, 1, 1)).reshape(bsz, maxlen).to(torch.uint8)
    out = ops.masked_attention(qkvr.to(DEV), kmem.to(DEV), vmem.to(DEV), memvalid.to(DEV), b_nd.to(DEV), bsz, t, heads, hid)
    kout, vout = ops.kv_memory_update(qkvr.to(DEV), kmem.to(DEV), vmem.to(DEV), bsz, t, hid)
    torch.cuda.synchronize()
    err = (out.cpu().float() - ref).abs().max().item()
    assert err < 2e-2, f"attention abs err {err} (bf16 output of O(1) values)"
    assert torch.equal(kout.cpu(), k_full[:, -maxlen:])
    assert torch.equal(vout.cpu(), v_full[:, -maxlen:])


def test_log_softmax_cols():
    g = torch.Generator().manual_seed(7)
    z = torch.randn(9, 8763, generator=g) * 3
    for col0, n in [(0, 8641), (8641, 121)]:
        out = ops.log_softmax_cols(z.to(DEV), col0, n, 2.0)
        torch.cuda.synchronize()
        ref = torch.log_softmax(z[:, col0:col0 + n] / 2.0, -1)
        assert (out.cpu() - ref).abs().max() < 2e-5


@pytest.mark.parametrize("bsz,t,cout", [(1, 7, 128), (2, 3, 64)])
def test_conv3d_temporal(bsz, t, cout):
    g = torch.Generator().manual_seed(8)
    W = torch.randn(cout, 3, 5, 1, 1, generator=g) * 0.4
    b = 0.1 * torch.randn(cout, generator=g)
    img = torch.randint(0, 256, (bsz, t, 128, 128, 3), generator=g, dtype=torch.uint8)
    sd = {"net.conv3d_layer.layer.weight": W, "net.conv3d_layer.layer.bias": b}
    ref = O.conv3d_temporal(sd, img.float() / 255.0).reshape(bsz * t, 128, 128, cout).permute(0, 3, 1, 2)
    wfrag, bp = packing.pack_conv3d_t5(W.to(DEV), b.to(DEV))
    st = torch.zeros(bsz * t, 2, dtype=torch.float64, device=DEV)
    y = ops.conv3d_t5(img.reshape(bsz * t, 128, 128, 3).to(DEV), wfrag, bp, cout, t, stats_out=st)
    torch.cuda.synchronize()
    out = packing.blocked_to_nchw(y.cpu(), cout, 128, 128)
    err = _relerr(out, ref)
    assert err < 1.5e-2, f"conv3d rel err {err}"
    assert torch.allclose(st.cpu(), _stats_of(out), rtol=1e-4, atol=1e-1)


@pytest.mark.parametrize("bsz,t,heads", [(1, 128, 2), (2, 12, 2)])
def test_full_attention(bsz, t, heads):
    g = torch.Generator().manual_seed(9)
    hid = heads * 128
    qkv = torch.randn(bsz * t, 3 * hid, generator=g)
    qkv[:, :hid] *= 3.0
    sp = lambda z: z.reshape(bsz, t, heads, 128).permute(0, 2, 1, 3)
    q, k, v = sp(qkv[:, :hid]), sp(qkv[:, hid:2 * hid]), sp(qkv[:, 2 * hid:])
    ref = (torch.softmax(q @ k.transpose(-1, -2) / 128.0, -1) @ v).permute(0, 2, 1, 3).reshape(bsz * t, hid)
    out = ops.full_attention(qkv.to(DEV), bsz, t, heads, hid)
    torch.cuda.synchronize()
    assert (out.cpu().float() - ref).abs().max() < 2e-2


def test_adam_step_matches_torch():
    """vpt_adam_step vs torch.optim.Adam with the BC script's hyper-parameters (behavioural_cloning.py:38-40,63-67)."""
    g = torch.Generator().manual_seed(10)
    n = 100003  # not a multiple of 4: exercises the tail
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=0.000181, weight_decay=0.039428)
    p = p0.clone().to(DEV)
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    for step in range(1, 4):
        grad = torch.randn(n, generator=g) * 0.1
        ref.grad = grad.clone()
        opt.step()
        ops.adam_step_(p, grad.to(DEV), m, v, step, lr=0.000181, weight_decay=0.039428)
    torch.cuda.synchronize()
    assert (p.cpu() - ref.detach()).abs().max() < 2e-6
    # grad_scale = 1/world folds the data-parallel mean into the update
    p2, m2, v2 = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    p3, m3, v3 = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    gsum = torch.randn(n, generator=g).to(DEV)
    ops.adam_step_(p2, gsum, m2, v2, 1, lr=1e-3, grad_scale=0.125)
    ops.adam_step_(p3, gsum * 0.125, m3, v3, 1, lr=1e-3)
    torch.cuda.synchronize()
    assert torch.allclose(p2, p3, atol=1e-7)


@pytest.mark.parametrize("m,k,n,relu_in,relu,with_res", [(1, 2048, 2048, False, False, True), (1, 2048, 6304, False, False, False), (1, 256, 2048, True, True, False),
                                                        (3, 2048, 8192, False, True, False), (8, 3072, 520, True, False, True), (1, 2048, 8763, False, False, False)])
def test_layernorm_linear_fused_equals_two_kernels(m, k, n, relu_in, relu, with_res):
    """The acting path's fused LayerNorm -> linear launch = vpt_layernorm_forward + vpt_linear_forward, bit for bit."""
    g = torch.Generator().manual_seed(m * 7 + k + n)
    x = (torch.randn(m, k, generator=g) * 1.3 + 0.2).to(DEV)
    gain = (1 + 0.2 * torch.randn(k, generator=g)).to(DEV)
    lb = (0.1 * torch.randn(k, generator=g)).to(DEV)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    wpk = ops.pack_linear(w.to(DEV))
    bias = (0.1 * torch.randn(n, generator=g)).to(DEV)
    res = torch.randn(m, n, generator=g).to(DEV) if with_res else None
    ln32, ln16 = ops.layernorm(x, gain, lb, relu_in=relu_in, out_f32=True)
    o32, o16 = ops.linear(ln16, wpk, n, bias=bias, res=res, relu=relu, out_f32=True, out_bf16=True)
    f_ln32, f32_, f16_ = ops.layernorm_linear(x, gain, lb, wpk, n, bias=bias, res=res, relu=relu, relu_in=relu_in, ln_out_f32=True, out_f32=True, out_bf16=True)
    torch.cuda.synchronize()
    assert torch.equal(f_ln32, ln32) and torch.equal(f32_, o32) and torch.equal(f16_, o16)
    ref = torch.nn.functional.layer_norm(torch.relu(x) if relu_in else x, (k,), gain, lb).cpu() @ w.t() + bias.cpu()
    if relu:
        ref = torch.relu(ref)
    if with_res:
        ref = ref + res.cpu()
    assert float((f32_.cpu() - ref).norm() / ref.norm()) < 1e-2
    with pytest.raises(ValueError):
        ops.layernorm_linear(x.repeat(9, 1), gain, lb, wpk, n)


@pytest.mark.parametrize("batch,heads,maxlen,valid", [(1, 16, 128, "all"), (1, 16, 128, "none"), (3, 8, 128, "random"), (2, 4, 37, "random"), (1, 24, 128, "tail")])
def test_attention_step_equals_attention_plus_memory_update(batch, heads, maxlen, valid):
    """Acting step kernel (t = 1) vs vpt_masked_attention_forward + vpt_kv_memory_update: memory bit-identical, output to fp32 rounding."""
    g = torch.Generator().manual_seed(batch * 100 + heads + maxlen)
    hid = heads * 128
    ld = 3 * hid + 10 * heads
    qkvr = (torch.randn(batch, ld, generator=g) * 1.5).to(DEV)
    kmem = torch.randn(batch, maxlen, hid, generator=g).to(DEV)
    vmem = torch.randn(batch, maxlen, hid, generator=g).to(DEV)
    b_nd = (0.5 * torch.randn(10, maxlen, generator=g)).to(DEV)
    if valid == "all":
        mv = torch.ones(batch, maxlen, dtype=torch.uint8)
    elif valid == "none":
        mv = torch.zeros(batch, maxlen, dtype=torch.uint8)
    elif valid == "tail":
        mv = torch.zeros(batch, maxlen, dtype=torch.uint8); mv[:, -5:] = 1
    else:
        mv = (torch.rand(batch, maxlen, generator=g) < 0.6).to(torch.uint8)
    mv = mv.to(DEV)
    first = torch.zeros(batch, dtype=torch.bool); first[0] = valid == "random"       # an episode start ignores (and invalidates) the memory
    first = first.to(DEV)
    memvalid = (mv.bool() & ~first[:, None]).to(torch.uint8).contiguous()
    ref = ops.masked_attention(qkvr, kmem, vmem, memvalid, b_nd, batch, 1, heads, hid)
    kref, vref = ops.kv_memory_update(qkvr, kmem, vmem, batch, 1, hid)
    mref = torch.cat([memvalid[:, 1:], torch.ones(batch, 1, dtype=torch.uint8, device=DEV)], dim=1)
    out, kout, vout, mout = ops.masked_attention_step(qkvr, kmem, vmem, mv.bool(), first, b_nd, batch, heads, hid)
    torch.cuda.synchronize()
    assert torch.equal(kout, kref) and torch.equal(vout, vref) and torch.equal(mout, mref)
    o, r = out.float().cpu(), ref.float().cpu()
    assert float((o - r).abs().max()) <= 2.0 ** -7 * float(r.abs().max()) + 1e-6      # one bf16 ulp of the largest value
    assert float((o - r).norm() / r.norm()) < 2e-3
    # in place: the same results land in the input buffers
    k2, v2, m2 = kmem.clone(), vmem.clone(), mv.clone()
    out2, ka, va, ma = ops.masked_attention_step(qkvr, k2, v2, m2, first, b_nd, batch, heads, hid, inplace=True)
    torch.cuda.synchronize()
    assert ka.data_ptr() == k2.data_ptr() and torch.equal(k2, kref) and torch.equal(v2, vref) and torch.equal(out2, out)
    assert ma.data_ptr() != m2.data_ptr() and torch.equal(ma, mref) and torch.equal(m2, mv)     # without a counter the mask is a new tensor (cross-workgroup read/write)
    # everything in place (vpt_masked_attention_step_inplace): the workgroup that arrives last writes the mask and re-zeroes the counter;
    # two chained steps = two chained steps of the copying variant
    k3, v3, m3 = kmem.clone(), vmem.clone(), mv.clone()
    done = torch.zeros(64, dtype=torch.int32, device=DEV)
    out3, kb, vb, mb = ops.masked_attention_step(qkvr, k3, v3, m3, first, b_nd, batch, heads, hid, inplace=True, done=done)
    torch.cuda.synchronize()
    assert mb.data_ptr() == m3.data_ptr() and torch.equal(m3, mref) and torch.equal(k3, kref) and torch.equal(v3, vref) and torch.equal(out3, out)
    assert int(done.abs().sum()) == 0
    nofirst = torch.zeros_like(first)
    o4r, k4r, v4r, m4r = ops.masked_attention_step(qkvr, kref, vref, mref, nofirst, b_nd, batch, heads, hid)
    o4, _, _, _ = ops.masked_attention_step(qkvr, k3, v3, m3, nofirst, b_nd, batch, heads, hid, inplace=True, done=done)
    torch.cuda.synchronize()
    assert torch.equal(o4, o4r) and torch.equal(k3, k4r) and torch.equal(v3, v4r) and torch.equal(m3, m4r) and int(done.abs().sum()) == 0
    with pytest.raises(ValueError):
        ops.masked_attention_step(qkvr, k3, v3, m3, first, b_nd, batch, heads, hid, inplace=False, done=done)
    with pytest.raises(ValueError):
        ops.masked_attention_step(qkvr.repeat(2, 1), kmem, vmem, mv, first, b_nd, batch, heads, hid)


@pytest.mark.parametrize("m", [1, 5])
def test_act_epilogue(m):
    """vpt_act_epilogue: summed head log-probs, de-normalised value, NaN flag and the packed record (MinecraftAgentPolicy.act's tail)."""
    g = torch.Generator().manual_seed(m)
    ab = torch.randint(0, 8641, (m,), generator=g).to(DEV); ac = torch.randint(0, 121, (m,), generator=g).to(DEV)
    lb = -torch.rand(m, generator=g).to(DEV) * 9; lc = -torch.rand(m, generator=g).to(DEV) * 5
    logits = torch.randn(m, 8763, generator=g).to(DEV)
    scale, shift = 1.7320508, -0.25
    keep, flag = ops.act_epilogue(ab, ac, lb, lc, logits, 8762, scale, shift)
    b2, c2, lp, vd, v = ops.unpack_act_keep(keep)
    torch.cuda.synchronize()
    assert torch.equal(b2, ab) and torch.equal(c2, ac) and int(flag) == 0
    assert torch.equal(lp, lb + lc) and torch.equal(v, logits[:, 8762])
    ref = torch.addcmul(torch.full((m,), shift, device=DEV), logits[:, 8762], torch.full((m,), scale, device=DEV))
    assert torch.allclose(vd, ref, rtol=1e-6, atol=1e-6)
    lb[m - 1] = float("nan")
    _, flag = ops.act_epilogue(ab, ac, lb, lc, logits, 8762, scale, shift)
    assert int(flag) == 1


@pytest.mark.parametrize("m,n,k,accumulate", [(300, 256, 192, False), (1000, 8768, 2048, False), (64, 65536, 256, True), (8192, 2048, 2048, False)])
def test_linear_wgrad_tn(m, n, k, accumulate):
    """dW = dY^T X from the row-major activations (vpt_gemm_tn_kernel, LDS transpose reads) against torch."""
    g = torch.Generator().manual_seed(9)
    dy = torch.randn(m, n, generator=g).to(torch.bfloat16)
    x = torch.randn(m, k, generator=g).to(torch.bfloat16)
    ref = dy.float().t() @ x.float()
    base = torch.randn(n, k, generator=g) if accumulate else None
    out = ops.linear_wgrad(dy.to(DEV), x.to(DEV), n, out=base.to(DEV).clone() if accumulate else None)
    torch.cuda.synchronize()
    if accumulate:
        ref = ref + base
    assert _relerr(out.cpu(), ref) < 2e-3, _relerr(out.cpu(), ref)


@pytest.mark.parametrize("m,n,ld,col0,stochastic", [(7, 8641, 8763, 0, False), (7, 121, 8763, 8641, True), (40, 2, 2, 0, True), (3, 8641, 8641, 0, True)])
def test_action_head_mask_sample_logprob(m, n, ld, col0, stochastic):
    """vpt_action_head_forward: CategoricalActionHead.forward with a mask (LOG0) + sample (arg-max / Gumbel-max on the caller's
    uniforms, first maximum) + logprob in one kernel (lib/action_head.py:163-207)."""
    g = torch.Generator().manual_seed(9)
    temp = 2.0
    logits = torch.randn(m, ld, generator=g) * 3
    mask = torch.rand(m, n, generator=g) > 0.4
    mask[:, 0] = True
    u = torch.rand(m, n, generator=g)
    u[0, min(5, n - 1)] = 1.0                    # the reference's guard: u == 1 -> 0.999
    z = logits[:, col0:col0 + n] / temp
    ref = torch.log_softmax(torch.where(mask, z, torch.full_like(z, -100.0)), -1)
    uu = u.clone(); uu[uu == 1.0] = 0.999
    want = torch.argmax(ref - torch.log(-torch.log(uu)), -1) if stochastic else torch.argmax(ref, -1)
    lp, ac, alp = ops.log_softmax_cols(logits.to(DEV), col0, n, temp, mask=mask.to(torch.uint8).to(DEV),
                                       noise=u.to(DEV) if stochastic else None, want_action=True)
    torch.cuda.synchronize()
    assert torch.allclose(lp.cpu(), ref, atol=2e-5)
    assert torch.equal(ac.cpu(), want)
    assert torch.allclose(alp.cpu(), ref.gather(1, want[:, None])[:, 0], atol=2e-5)
    # ties: the FIRST maximum, as torch.argmax
    tie = torch.zeros(2, 300)
    tie[0, 17] = tie[0, 200] = 5.0
    tie[1, 299] = tie[1, 3] = 2.0
    _, ac2, _ = ops.log_softmax_cols(tie.to(DEV), 0, 300, 1.0, want_action=True)
    assert ac2.cpu().tolist() == [17, 3]


@pytest.mark.parametrize("cout,cin,dtype", [(128, 128, torch.bfloat16), (160, 64, torch.bfloat16), (256, 256, torch.float16)])
def test_device_pack_conv3x3_bit_exact(cout, cin, dtype):
    """vpt_pack_conv3x3 (the C-ABI re-pack a torch-free host would use) == packing.pack_conv3x3, bit for bit."""
    g = torch.Generator().manual_seed(12)
    W = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(DEV)
    gain = (1 + 0.2 * torch.randn(cin, generator=g)).to(DEV)
    bias = (0.1 * torch.randn(cin, generator=g)).to(DEV)
    wref, sa_ref, sg_ref = packing.pack_conv3x3(W, gain, bias, dtype=dtype)
    wpk, sa, sg = ops.pack_conv3x3(W, gain, bias, dtype=dtype)
    torch.cuda.synchronize()
    assert torch.equal(wpk.view(torch.int16), wref.view(torch.int16))
    assert torch.allclose(sa, sa_ref, rtol=0, atol=1e-6 * float(sa_ref.abs().max()))
    assert torch.allclose(sg, sg_ref, rtol=0, atol=1e-6 * float(sg_ref.abs().max()))
    # dgrad operand: transposed + flipped weight, unit gain, no tables
    wd = (W * gain.view(1, -1, 1, 1)).to(dtype).float().permute(1, 0, 2, 3).flip(2, 3).contiguous()
    ones = torch.ones(cout, device=DEV)
    wpk_d, none_a, none_g = ops.pack_conv3x3(wd, ones, None, tables=False, dtype=dtype)
    assert none_a is None and none_g is None
    ref_d = packing.pack_conv3x3(wd, ones, None, tables=False, dtype=dtype)[0]
    bad = (wpk_d.view(torch.int16) != ref_d.view(torch.int16))
    assert int(bad.sum()) == 0, (int(bad.sum()), wpk_d[bad][:4], ref_d[bad][:4])


@pytest.mark.parametrize("n,k", [(8763, 2048), (256, 65536), (2048, 256)])
def test_device_pack_linear_bit_exact(n, k):
    g = torch.Generator().manual_seed(13)
    W = (torch.randn(n, k, generator=g) / k ** 0.5).to(DEV)
    assert torch.equal(ops.pack_linear(W).view(torch.int16), packing.pack_linear(W).view(torch.int16))
    # transposed: the input-gradient operand W^T with the reduction dimension (n) padded to a multiple of 64
    if k <= 8192:
        np_ = (n + 63) // 64 * 64
        wt = torch.zeros(k, np_, device=DEV)
        wt[:, :n] = W.t()
        assert torch.equal(ops.pack_linear(W, transposed=True, k_pad=np_).view(torch.int16), packing.pack_linear(wt).view(torch.int16))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_device_pack_first_convs_and_permutes_bit_exact(dtype):
    """vpt_pack_conv_first / vpt_pack_conv3d_t5 / vpt_chw_to_blocked (what finishes "a torch-free host can prepare a model") ==
    the packing.py restatements, bit for bit; and vpt_workspace_bytes agrees with the sizes ops.py allocates."""
    from vpt_amd import _native
    g = torch.Generator().manual_seed(9)
    for cout in (128, 64, 192):
        W = (torch.randn(cout, 3, 3, 3, generator=g) * 0.3).to(DEV)
        b = (torch.randn(cout, generator=g) * 0.1).to(DEV)
        # (reference on the CPU: IEEE fp32 division W / 255; torch's GPU division kernel differs from it by one fp32 ulp on rare
        # elements, which an fp16 rounding tie then exposes -- the device pack uses the correctly rounded quotient)
        got, want = ops.pack_conv_first(W, b, dtype=dtype).cpu(), packing.pack_conv_first(W.cpu(), b.cpu(), dtype=dtype)
        bad = (got.view(torch.int16) != want.view(torch.int16)).nonzero()
        assert bad.numel() == 0, (cout, bad[:6].tolist(), got.view(-1)[:0].dtype, [(float(got[tuple(i)]), float(want[tuple(i)])) for i in bad[:6].tolist()])
    for o in (128, 96):
        W = (torch.randn(o, 3, 5, 1, 1, generator=g) * 0.3).to(DEV)
        b = (torch.randn(o, generator=g) * 0.1).to(DEV)
        f1, b1 = ops.pack_conv3d_t5(W, b, dtype=dtype)
        f2, b2 = packing.pack_conv3d_t5(W, b, dtype=dtype)
        assert torch.equal(f1.view(torch.int16), f2.view(torch.int16)) and torch.equal(b1, b2)
    c, h, w = 64, 16, 16
    m = torch.randn(7, c * h * w, generator=g).to(DEV)
    assert torch.equal(ops.chw_to_blocked(m, c, h, w), packing.chw_to_blocked_columns(m, c, h, w))
    v = torch.randn(c * h * w, generator=g).to(DEV)
    assert torch.equal(ops.chw_to_blocked(v, c, h, w), packing.chw_to_blocked_vector(v, c, h, w))
    lib = _native.load("bf16")
    assert lib.vpt_workspace_bytes(1, 8, 0, 0, 64, 128) == 4 * lib.vpt_conv3x3_wgrad_scratch_floats(8, 64, 128)
    assert lib.vpt_workspace_bytes(2, 8, 0, 0, 0, 128) == 4 * (8 * (9 * 128 + 4) + 1 * 2 * 9 * 128)      # per-frame sums + one 32-frame block of d_sa / d_sg partials
    assert lib.vpt_workspace_bytes(3, 16, 1024, 256, 0, 0) == 4 * 16 * 1024 * 256 and lib.vpt_workspace_bytes(99, 1, 1, 1, 1, 1) == -1


@pytest.mark.parametrize("frames,c,h,w", [(3, 128, 64, 64), (2, 256, 32, 32), (5, 64, 16, 16), (2, 96, 32, 48)])
@pytest.mark.parametrize("fmt", ["bf16", "fp16"])
def test_group_norm_n_folded_into_block0(frames, c, h, w, fmt):
    """CnnDownStack.forward after the pool (lib/impala_cnn.py:118-121): x = n(P); x = x + conv1(conv0(x)) -- with the GroupNorm `n`
    FOLDED (vpt_channel_stats + vpt_nfold_coef + vpt_conv3x3_forward_folded on Q = n.weight * P: no affine pass, x never written) against
    (1) the fp32 composition of the reference's layers and (2) the unfolded HIP path (vpt_frame_affine_forward -> conv -> conv + res)."""
    dt = torch.bfloat16 if fmt == "bf16" else torch.float16
    g = torch.Generator().manual_seed(23)
    P_ = torch.relu(torch.randn(frames, c, h, w, generator=g) * 1.3 + 0.4).to(dt)            # a pooled post-ReLU tensor
    gn, bn = 1 + 0.2 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    gn[3] = -0.7                                                                             # a negative gain must work too
    conv = []
    for _ in range(2):
        conv.append({"layer.weight": torch.randn(c, c, 3, 3, generator=g) * (1.6 / (c * 9) ** 0.5),
                     "norm.weight": 1 + 0.2 * torch.randn(c, generator=g), "norm.bias": 0.1 * torch.randn(c, generator=g)})
    x_ref = O.group_norm_1(P_.float(), gn, bn)
    ref = x_ref + O._norm_conv_relu(conv[1], "", O._norm_conv_relu(conv[0], "", x_ref))
    dev = lambda t_: t_.to(DEV)
    pk = [ops.pack_conv3x3(dev(cv["layer.weight"]), dev(cv["norm.weight"]), dev(cv["norm.bias"]), dtype=dt) for cv in conv]
    Pb = packing.nchw_to_blocked(P_.float(), dtype=dt).to(DEV)
    tot = _stats_of(P_.float()).to(DEV)
    # ---- unfolded HIP path
    s_x = torch.zeros(frames, 2, dtype=torch.float64, device=DEV)
    xb = ops.frame_affine(Pb, dev(gn), dev(bn), tot, stats_out=s_x)
    s_y = torch.zeros(frames, 2, dtype=torch.float64, device=DEV)
    yb = ops.conv3x3(xb, *pk[0], s_x, c, stats_out=s_y)
    s_o = torch.zeros(frames, 2, dtype=torch.float64, device=DEV)
    unfolded = ops.conv3x3(yb, *pk[1], s_y, c, res=xb, stats_out=s_o)
    # ---- folded: Q = gain * P as the producer would store it
    Qb = packing.nchw_to_blocked((P_.float() * gn.view(1, -1, 1, 1)).to(dt).float(), dtype=dt).to(DEV)
    chs = ops.channel_stats(Qb)
    q = packing.blocked_to_nchw(Qb.cpu(), c, h, w).double()
    want_chs = torch.stack([q.sum((2, 3)), (q * q).sum((2, 3))], -1)
    assert torch.allclose(chs.cpu(), want_chs, rtol=1e-6, atol=1e-4)
    wp = (conv[0]["layer.weight"] * conv[0]["norm.weight"].view(1, -1, 1, 1)).to(dt).double()
    m = packing.edge_tap_matrix("cpu", torch.float64)
    pad = (c + 127) // 128 * 128
    tabs = []
    for v in (bn, gn):
        tap = (wp * v.double().view(1, -1, 1, 1)).sum(1).view(c, 9)
        t_ = torch.zeros(9, pad)
        t_[:, :c] = (m @ tap.t()).float()
        tabs.append(t_.contiguous().to(DEV))
    kk, rs, rsc, rb = ops.nfold_coef(tot, chs, dev(gn), dev(bn), pk[0][1], pk[0][2], tabs[0], tabs[1], h * w, c)
    s_y2 = torch.zeros(frames, 2, dtype=torch.float64, device=DEV)
    y2 = ops.conv3x3_folded(Qb, pk[0][0], pk[0][1], pk[0][2], None, c, kk_frame=kk, rs_frame=rs, stats_out=s_y2)
    s_o2 = torch.zeros(frames, 2, dtype=torch.float64, device=DEV)
    folded = ops.conv3x3_folded(y2, *pk[1], s_y2, c, res=Qb, res_scale=rsc, res_bias=rb, stats_out=s_o2)
    torch.cuda.synchronize()
    # the scalars against their definitions
    n_tot = c * h * w
    mu_p = tot[:, 0].cpu() / n_tot
    r_p = torch.rsqrt(tot[:, 1].cpu() / n_tot - mu_p * mu_p + 1e-5)
    assert torch.allclose(rsc.cpu().double(), r_p, rtol=1e-5)
    xq = r_p.view(-1, 1, 1, 1) * q + (bn.double().view(1, -1, 1, 1) - (r_p * mu_p).view(-1, 1, 1, 1) * gn.double().view(1, -1, 1, 1))
    assert torch.allclose(rb.cpu().double(), bn.double().view(1, -1) - (r_p * mu_p).view(-1, 1) * gn.double().view(1, -1), rtol=1e-5, atol=1e-6)
    r_x = torch.rsqrt(xq.reshape(frames, -1).var(1, unbiased=False) + 1e-5)
    assert torch.allclose(rs.cpu().double(), r_x * r_p, rtol=1e-4), (rs.cpu(), r_x * r_p)
    out_f = packing.blocked_to_nchw(folded.cpu(), c, h, w)
    out_u = packing.blocked_to_nchw(unfolded.cpu(), c, h, w)
    e_ref, e_unf, e_u_ref = _relerr(out_f, ref), _relerr(out_f, out_u), _relerr(out_u, ref)
    print(f"NFOLD[{fmt}] {frames}x{c}x{h}x{w}: folded vs fp32 reference {e_ref:.2e} (unfolded path {e_u_ref:.2e}); folded vs unfolded {e_unf:.2e}")
    assert e_ref < 2e-2 and e_ref < 1.5 * e_u_ref + 2e-3, (e_ref, e_u_ref)
    assert torch.allclose(s_o2.cpu(), _stats_of(out_f.float()), rtol=5e-3, atol=1.0)
    assert torch.allclose(s_y2.cpu(), s_y.cpu(), rtol=2e-2)

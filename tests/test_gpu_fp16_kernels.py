"""The precision="fp16" build of the kernels (libvpt_hip_f16.so: same sources, IEEE-half MFMA operands) against fp32
references, per kernel.  Needs an MI355X.  Inputs are rounded to fp16 first so what remains is the rounding of the
weights and of the stored outputs: 11-bit significands -> bounds 8x tighter than the bf16 tests' (2.5e-3 of max)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import vpt_amd  # noqa: E402,F401
from vpt_amd import _native, ops, packing  # noqa: E402
from oracle import vpt_oracle as O  # noqa: E402

DEV = "cuda"
H = torch.float16


def _relerr(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp(min=1e-12)).item()


def _stats_of(x):
    flat = x.reshape(x.shape[0], -1).double()
    return torch.stack([flat.sum(1), (flat * flat).sum(1)], dim=1).contiguous()


def test_library_reports_its_format():
    assert _native.load("fp16").vpt_operand_format() == b"fp16"
    assert _native.load("bf16").vpt_operand_format() == b"bf16"


def test_mixed_formats_rejected():
    x = torch.zeros(1, 1, 16, 16, 32, dtype=H, device=DEV)
    w, sa, sg = packing.pack_conv3x3(torch.zeros(32, 32, 3, 3, device=DEV), torch.ones(32, device=DEV), torch.zeros(32, device=DEV))  # bf16 pack
    with pytest.raises(TypeError):
        ops.conv3x3(x, w, sa, sg, torch.zeros(1, 2, dtype=torch.float64, device=DEV), 32)


@pytest.mark.parametrize("frames,h,w,cin,cout,use_res", [(2, 16, 16, 128, 128, True), (1, 32, 32, 64, 160, False), (2, 32, 32, 256, 256, True)])
@pytest.mark.parametrize("tiling", ["throughput", "latency"])
def test_conv3x3_fp16(frames, h, w, cin, cout, use_res, tiling):
    g = torch.Generator().manual_seed(1)
    W = torch.randn(cout, cin, 3, 3, generator=g) * (1.6 / (cin * 9) ** 0.5)
    gain = 1 + 0.2 * torch.randn(cin, generator=g)
    bias = 0.1 * torch.randn(cin, generator=g)
    xh = (torch.relu(torch.randn(frames, cin, h, w, generator=g)) + 0.2 * torch.randn(frames, cin, h, w, generator=g)).to(H)
    res = torch.randn(frames, cout, h, w, generator=g).to(H) if use_res else None
    ref = O._norm_conv_relu({"norm.weight": gain, "norm.bias": bias, "layer.weight": W}, "", xh.float())
    if use_res:
        ref = ref + res.float()
    wpk, sa, sg = packing.pack_conv3x3(W.to(DEV), gain.to(DEV), bias.to(DEV), dtype=H)
    st_out = torch.zeros(frames, 2, dtype=torch.float64, device=DEV)
    y = ops.conv3x3(packing.nchw_to_blocked(xh.float(), dtype=H).to(DEV), wpk, sa, sg, _stats_of(xh.float()).to(DEV), cout,
                    res=packing.nchw_to_blocked(res.float(), dtype=H).to(DEV) if use_res else None, stats_out=st_out, tiling=tiling)
    torch.cuda.synchronize()
    assert y.dtype == H
    err = _relerr(packing.blocked_to_nchw(y.cpu(), cout, h, w), ref)
    assert err < 2.5e-3, f"conv3x3 fp16 rel err {err}"
    assert torch.allclose(st_out.cpu(), _stats_of(ref), rtol=1e-3, atol=0.5)


def test_conv_first_pool_affine_fp16():
    g = torch.Generator().manual_seed(2)
    cout = 128
    W = torch.randn(cout, 3, 3, 3, generator=g) * 0.3
    b = 0.1 * torch.randn(cout, generator=g)
    img = torch.randint(0, 256, (2, 128, 128, 3), generator=g, dtype=torch.uint8)
    ref = F.max_pool2d(torch.relu(F.conv2d(img.permute(0, 3, 1, 2).float() / 255.0, W, b, padding=1)), 3, 2, 1)
    st = torch.zeros(2, 2, dtype=torch.float64, device=DEV)
    y = ops.conv_first(img.to(DEV), packing.pack_conv_first(W.to(DEV), b.to(DEV), dtype=H), cout, stats_out=st)
    torch.cuda.synchronize()
    assert y.dtype == H
    out = packing.blocked_to_nchw(y.cpu(), cout, 64, 64)
    assert _relerr(out, ref) < 2e-3, _relerr(out, ref)
    assert torch.allclose(st.cpu(), _stats_of(out), rtol=1e-4, atol=1e-2)
    # max-pool on fp16 bit patterns is exact; affine rounds once
    st2 = torch.zeros(2, 2, dtype=torch.float64, device=DEV)
    p = ops.maxpool(y, stats_out=st2)
    torch.cuda.synchronize()
    assert torch.equal(packing.blocked_to_nchw(p.cpu(), cout, 32, 32), F.max_pool2d(out, 3, 2, 1))
    gain, bias = 1 + 0.2 * torch.randn(cout, generator=g), 0.1 * torch.randn(cout, generator=g)
    z = ops.frame_affine(p, gain.to(DEV), bias.to(DEV), st2)
    torch.cuda.synchronize()
    refz = O.group_norm_1(F.max_pool2d(out, 3, 2, 1), gain, bias)
    assert _relerr(packing.blocked_to_nchw(z.cpu(), cout, 32, 32), refz) < 1e-3


@pytest.mark.parametrize("m,n,k,bias,relu,res,splitk", [(300, 8763, 2048, True, False, False, 1), (513, 2048, 8192, True, False, True, 1),
                                                        (40, 256, 16384, False, False, False, 16), (128, 4096, 4096, True, True, True, 1),
                                                        (1, 8763, 2048, True, False, False, 1)])
def test_linear_fp16(m, n, k, bias, relu, res, splitk):
    g = torch.Generator().manual_seed(4)
    A = torch.randn(m, k, generator=g).to(H)
    W = torch.randn(n, k, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g) if bias else None
    r = torch.randn(m, n, generator=g) if res else None
    ref = A.float() @ W.to(H).float().t()
    if bias:
        ref = ref + b
    if relu:
        ref = torch.relu(ref)
    if res:
        ref = ref + r
    o32, o16 = ops.linear(A.to(DEV), packing.pack_linear(W.to(DEV), dtype=H), n, bias=b.to(DEV) if bias else None,
                          res=r.to(DEV) if res else None, relu=relu, out_f32=True, out_bf16=(splitk == 1), splitk=splitk)
    torch.cuda.synchronize()
    assert _relerr(o32.cpu(), ref) < 5e-4, _relerr(o32.cpu(), ref)
    if o16 is not None:
        assert o16.dtype == H and _relerr(o16.cpu().float(), ref) < 1.5e-3


def test_layernorm_and_attention_emit_fp16():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(130, 2048, generator=g)
    gain, bias = 1 + 0.2 * torch.randn(2048, generator=g), 0.1 * torch.randn(2048, generator=g)
    _, o16 = ops.layernorm(x.to(DEV), gain.to(DEV), bias.to(DEV), dtype=H)
    torch.cuda.synchronize()
    assert o16.dtype == H and _relerr(o16.cpu().float(), O.layer_norm(x, gain, bias)) < 1e-3

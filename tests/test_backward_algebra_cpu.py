"""CPU (fp64) checks of the backward ALGEBRA the round-5 kernels implement, against torch autograd on the reference's order of operations
(GroupNorm(1, C) -> zero-pad -> conv3x3 -> ReLU [+ residual], /root/reference/lib/util.py:75-82, lib/impala_cnn.py:50-52,114-119):

  * the folded layer's input gradient dx = conv^T(W', dacc) + c0 + c1 x and its parameter gradients through the edge-class tables
    (vpt_conv_bwd_prep_kernel + vpt_conv_bwd_finish_kernel + training.conv_param_grads) -- the formulas every `prepare` variant shares;
  * the round-5 block path: conv1's dgrad epilogue writes conv0's operand dacc0 = rstd0 dy [y > 0] and u = sum rstd0 dy y, and the reduce-only
    pass recovers S = sum dacc0 / rstd0, T1 = u / rstd0 - <SA, S>, T2 = <SG, S> (vpt_conv3x3_kernel mode 6, vpt_conv_bwd_prep_kernel<PRE>);
  * the arg-max mask format (oracle/pool_mask.py) against torch's max_pool2d indices INCLUDING ties, and the routing rule of
    vpt_conv_bwd_prep_pooled_kernel (gate [P > 0], value at the arg-max = P) against autograd through conv -> ReLU -> max_pool2d;
  * the GroupNorm-`n` backward applied per element from the reduce pass's (A, B) (the pooled kernel's NFOLD path).
The edge tables come from the product's packing code, so the class / tap conventions the kernels rely on are what is tested."""
import numpy as np
import torch
import torch.nn.functional as F

import vpt_amd  # noqa: F401
from vpt_amd import packing
from vpt_amd.training import conv_param_grads
from oracle import pool_mask

D = torch.float64
EPS = 1e-5


def _classes(h, w):
    ey = torch.tensor([0 if y == 0 else (2 if y == h - 1 else 1) for y in range(h)])
    ex = torch.tensor([0 if x == 0 else (2 if x == w - 1 else 1) for x in range(w)])
    return (ey.view(-1, 1) * 3 + ex.view(1, -1))                       # [h, w] edge class of every pixel


def _tables(W, gain, bias):
    """SA / SG [9, Cout] of the fold: sums over the taps inside the image of W * bias / W * gain (vpt_pack_conv3x3, exact in fp64)."""
    m = packing.edge_tap_matrix("cpu", D)                               # [9 classes, 9 taps]
    cout = W.shape[0]
    sa = m @ (W * bias.view(1, -1, 1, 1)).sum(1).reshape(cout, 9).t()
    sg = m @ (W * gain.view(1, -1, 1, 1)).sum(1).reshape(cout, 9).t()
    return sa, sg


def _layer(x, W, gain, bias, res=None):
    mu = x.mean(dim=(1, 2, 3), keepdim=True)
    rstd = torch.rsqrt(x.var(dim=(1, 2, 3), unbiased=False, keepdim=True) + EPS)
    v = F.conv2d((x - mu) * rstd * gain.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1), W, padding=1)
    return torch.relu(v) + (res if res is not None else 0), v


def _finish(S_f, tv, sa, sg, mu, rstd, n):
    """vpt_conv_bwd_finish_kernel: per-frame edge-class sums S_f [F, 9, Cout] and data term tv [F] -> (c0, c1, dSA, dSG)."""
    T1 = tv - (sa.unsqueeze(0) * S_f).sum((1, 2))
    T2 = (sg.unsqueeze(0) * S_f).sum((1, 2))
    c1 = -(rstd * rstd) * T1 / n
    c0 = -(rstd / n) * T2 - c1 * mu
    return c0, c1, S_f.sum(0), (-(rstd * mu).view(-1, 1, 1) * S_f).sum(0)


def _class_sums(dz, e):
    f, cout = dz.shape[:2]
    S = torch.zeros(f, 9, cout, dtype=D)
    for c in range(9):
        S[:, c] = (dz * (e == c).view(1, 1, *e.shape)).sum((2, 3))
    return S


def _backward_through_fold(x, W, gain, bias, dz_times_rstd, tv, S_f):
    """dx and the parameter gradients from the operand dacc = rstd dz, the data term and the class sums -- what dgrad / wgrad / finish / host compute."""
    f, cin, h, w = x.shape
    n = cin * h * w
    mu = x.mean(dim=(1, 2, 3))
    rstd = torch.rsqrt(x.var(dim=(1, 2, 3), unbiased=False) + EPS)
    sa, sg = _tables(W, gain, bias)
    c0, c1, d_sa, d_sg = _finish(S_f, tv, sa, sg, mu, rstd, n)
    Wp = W * gain.view(1, -1, 1, 1)
    dx = F.conv_transpose2d(dz_times_rstd, Wp, padding=1) + c0.view(-1, 1, 1, 1) + c1.view(-1, 1, 1, 1) * x
    # wgrad kernel: dw_raw[o][tap][c] = sum_{f, p} dacc[f][o][p] x[f][c][p + tap]
    xp = F.pad(x, (1, 1, 1, 1))
    dw_raw = torch.stack([torch.einsum("foyx,fcyx->oc", dz_times_rstd, xp[:, :, kh:kh + h, kw:kw + w]) for kh in range(3) for kw in range(3)], 1)
    pad = lambda t: torch.cat([t, torch.zeros(9, 0, dtype=D)], 1)
    dW, dgain, dbias = conv_param_grads(dw_raw.contiguous(), pad(d_sa), pad(d_sg), W, gain, bias)
    return dx, dW, dgain, dbias


def test_folded_layer_backward_equals_autograd():
    g = torch.Generator().manual_seed(1)
    f, cin, cout, h, w = 3, 6, 5, 8, 10
    x = torch.randn(f, cin, h, w, generator=g, dtype=D).requires_grad_(True)
    W = (torch.randn(cout, cin, 3, 3, generator=g, dtype=D) * 0.3).requires_grad_(True)
    gain = (1 + 0.3 * torch.randn(cin, generator=g, dtype=D)).requires_grad_(True)
    bias = (0.2 * torch.randn(cin, generator=g, dtype=D)).requires_grad_(True)
    res = torch.randn(f, cout, h, w, generator=g, dtype=D)
    dY = torch.randn(f, cout, h, w, generator=g, dtype=D)
    out, v = _layer(x, W, gain, bias, res)
    gx, gW, gg, gb = torch.autograd.grad((out * dY).sum(), [x, W, gain, bias])
    with torch.no_grad():
        xd = x.detach()
        rstd = torch.rsqrt(xd.var(dim=(1, 2, 3), unbiased=False) + EPS)
        vv = out.detach() - res                                         # what prepare sees: stored output minus stored residual
        dz = dY * (vv > 0)
        dx, dW, dgain, dbias = _backward_through_fold(xd, W.detach(), gain.detach(), bias.detach(), dz * rstd.view(-1, 1, 1, 1), (dz * vv).sum((1, 2, 3)),
                                                      _class_sums(dz, _classes(h, w)))
    for a, b in ((dx, gx), (dW, gW), (dgain, gg), (dbias, gb)):
        assert torch.allclose(a, b, rtol=1e-9, atol=1e-10), float((a - b).abs().max())


def test_block_backward_with_the_gate_in_conv1s_dgrad_epilogue():
    """x -> y = conv0(x) (no residual) -> out = conv1(y) + x.  Mode 6: the dgrad of conv1 produces dacc0 = rstd0 dy [y > 0] and
    u = sum rstd0 dy y; the reduce-only pass divides the sums of dacc0 by rstd0.  Everything downstream must equal autograd."""
    g = torch.Generator().manual_seed(2)
    f, c, h, w = 2, 5, 8, 8
    x = torch.randn(f, c, h, w, generator=g, dtype=D).requires_grad_(True)
    mk = lambda: ((torch.randn(c, c, 3, 3, generator=g, dtype=D) * 0.3).requires_grad_(True), (1 + 0.3 * torch.randn(c, generator=g, dtype=D)).requires_grad_(True),
                  (0.2 * torch.randn(c, generator=g, dtype=D)).requires_grad_(True))
    (W0, g0, b0), (W1, g1, b1) = mk(), mk()
    y, _ = _layer(x, W0, g0, b0)
    out, _ = _layer(y, W1, g1, b1, res=x)
    dOut = torch.randn(f, c, h, w, generator=g, dtype=D)
    gx, gW0, gg0, gb0 = torch.autograd.grad((out * dOut).sum(), [x, W0, g0, b0])
    with torch.no_grad():
        xd, yd = x.detach(), y.detach()
        e = _classes(h, w)
        # conv1 (residual layer): the ordinary prepare
        rstd_y = torch.rsqrt(yd.var(dim=(1, 2, 3), unbiased=False) + EPS)
        v1 = out.detach() - xd
        dz1 = dOut * (v1 > 0)
        sa1, sg1 = _tables(W1.detach(), g1.detach(), b1.detach())
        c0, c1, _, _ = _finish(_class_sums(dz1, e), (dz1 * v1).sum((1, 2, 3)), sa1, sg1, yd.mean(dim=(1, 2, 3)), rstd_y, c * h * w)
        dy = F.conv_transpose2d(dz1 * rstd_y.view(-1, 1, 1, 1), W1.detach() * g1.detach().view(1, -1, 1, 1), padding=1) + c0.view(-1, 1, 1, 1) + c1.view(-1, 1, 1, 1) * yd
        # mode 6 epilogue: conv0's operand and data term straight from dy and xin = y
        rstd0 = torch.rsqrt(xd.var(dim=(1, 2, 3), unbiased=False) + EPS).view(-1, 1, 1, 1)
        dacc0 = rstd0 * dy * (yd > 0)
        u = (rstd0 * dy * yd).sum((1, 2, 3))                           # closed gates contribute 0 because y = 0 there
        # reduce-only pass: sums of dacc0, scaled back
        S_f = _class_sums(dacc0, e) / rstd0.view(-1, 1, 1)
        dx0, dW0, dg0, db0 = _backward_through_fold(xd, W0.detach(), g0.detach(), b0.detach(), dacc0, u / rstd0.view(-1), S_f)
        dx = dx0 + dOut                                                 # + the skip connection (mode 3's `skip`)
    for a, b in ((dx, gx), (dW0, gW0), (dg0, gg0), (db0, gb0)):
        assert torch.allclose(a, b, rtol=1e-9, atol=1e-10), float((a - b).abs().max())


def test_argmax_masks_follow_torchs_first_maximum_rule():
    """Integer-valued inputs with many ties and exact zeros: decoded masks == torch's max_pool2d indices wherever the maximum is positive."""
    rng = np.random.default_rng(3)
    x = np.maximum(rng.integers(-3, 4, size=(2, 3, 12, 16)).astype(np.float64), 0.0)
    pooled, masks = pool_mask.pool_argmax_masks(x)
    ref, idx = F.max_pool2d(torch.from_numpy(x), 3, 2, 1, return_indices=True)
    assert np.array_equal(pooled, ref.numpy())
    k = pool_mask.decode_first_position(masks)
    py, px = np.meshgrid(np.arange(6), np.arange(8), indexing="ij")
    yy, xx = 2 * py - 1 + k // 3, 2 * px - 1 + k % 3
    live = pooled > 0
    assert live.mean() > 0.5 and (~live).any()
    assert np.array_equal((yy * 16 + xx)[live], idx.numpy()[live])
    assert (masks >> 9 == 0).all()
    # corner windows: the positions outside the image carry a 1
    assert ((masks[..., 0, 0] & 0b111100100) == 0b111100100).all()      # k = 0, 1, 2 (row -1) and 0, 3, 6 (column -1): bits 8, 7, 6, 5, 2


def test_pooled_layer_backward_from_masks_equals_autograd():
    """GN -> conv -> ReLU -> max_pool2d: the pooled kernel's routing (gate [P > 0], first zero bit, value at the arg-max = P) and the shared finish
    algebra against autograd; then the same with the GroupNorm-`n` backward applied per element from the reduce pass's sums (NFOLD)."""
    g = torch.Generator().manual_seed(4)
    f, cin, cout, h, w = 2, 4, 6, 8, 12
    x = torch.randn(f, cin, h, w, generator=g, dtype=D).requires_grad_(True)
    W = (torch.randn(cout, cin, 3, 3, generator=g, dtype=D) * 0.4).requires_grad_(True)
    gain = (1 + 0.3 * torch.randn(cin, generator=g, dtype=D)).requires_grad_(True)
    bias = (0.2 * torch.randn(cin, generator=g, dtype=D) - 0.3).requires_grad_(True)
    ng = (1 + 0.3 * torch.randn(cout, generator=g, dtype=D)).requires_grad_(True)
    nb = (0.2 * torch.randn(cout, generator=g, dtype=D)).requires_grad_(True)
    pre, _ = _layer(x, W, gain, bias)
    P = F.max_pool2d(pre, 3, 2, 1)
    mu_p = P.mean(dim=(1, 2, 3), keepdim=True)
    r_p = torch.rsqrt(P.var(dim=(1, 2, 3), unbiased=False, keepdim=True) + EPS)
    xn = (P - mu_p) * r_p * ng.view(1, -1, 1, 1) + nb.view(1, -1, 1, 1)
    G = torch.randn(f, cout, h // 2, w // 2, generator=g, dtype=D)       # gradient w.r.t. n(P)
    gx, gW, gg, gb, gng, gnb = torch.autograd.grad((xn * G).sum(), [x, W, gain, bias, ng, nb])
    with torch.no_grad():
        Pd, xd = P.detach(), x.detach()
        # reduce pass (vpt_affine_bwd_reduce_kernel): A, B per frame; d gain / d bias per channel
        xh = (Pd - mu_p.detach()) * r_p.detach()
        cnt = cout * (h // 2) * (w // 2)
        A = (G * ng.detach().view(1, -1, 1, 1)).sum((1, 2, 3), keepdim=True) / cnt
        B = (G * ng.detach().view(1, -1, 1, 1) * xh).sum((1, 2, 3), keepdim=True) / cnt
        assert torch.allclose((G * xh).sum((0, 2, 3)), gng, rtol=1e-9, atol=1e-10) and torch.allclose(G.sum((0, 2, 3)), gnb, rtol=1e-9, atol=1e-10)
        dP = r_p.detach() * (G * ng.detach().view(1, -1, 1, 1) - A - xh * B)   # NFOLD: formed per element inside the pooled kernel
        # pooled kernel: masks -> arg-max position, gate [P > 0], scatter to the pre-pool resolution
        _, masks = pool_mask.pool_argmax_masks(pre.detach().numpy())
        k = torch.from_numpy(pool_mask.decode_first_position(masks))
        gd = dP * (Pd > 0)
        dz = torch.zeros(f, cout, h, w, dtype=D)
        py, px = torch.meshgrid(torch.arange(h // 2), torch.arange(w // 2), indexing="ij")
        yy, xx = (2 * py - 1).view(1, 1, h // 2, w // 2) + k // 3, (2 * px - 1).view(1, 1, h // 2, w // 2) + k % 3
        live = Pd > 0
        fi, ci = torch.meshgrid(torch.arange(f), torch.arange(cout), indexing="ij")
        fi, ci = fi.view(f, cout, 1, 1).expand_as(k), ci.view(f, cout, 1, 1).expand_as(k)
        dz.index_put_((fi[live], ci[live], yy[live], xx[live]), gd[live], accumulate=True)
        rstd = torch.rsqrt(xd.var(dim=(1, 2, 3), unbiased=False) + EPS)
        tv = (gd * Pd).sum((1, 2, 3))                                   # the layer's value at the arg-max is the pooled value itself
        dx, dW, dgain, dbias = _backward_through_fold(xd, W.detach(), gain.detach(), bias.detach(), dz * rstd.view(-1, 1, 1, 1), tv, _class_sums(dz, _classes(h, w)))
    for a, b in ((dx, gx), (dW, gW), (dgain, gg), (dbias, gb)):
        assert torch.allclose(a, b, rtol=1e-9, atol=1e-10), float((a - b).abs().max())

"""CPU (fp64) checks of the round-4 fold algebra -- the formulas vpt_nfold_coef_kernel / vpt_conv3x3_kernel modes 0 + 5 and
vpt_dense_fold_epilogue_kernel implement -- against the reference's order of operations
(lib/impala_cnn.py:99-100,118-121: x = n(pool); block0(x) with conv = GroupNorm -> zero-pad -> conv -> ReLU, lib/util.py:75-82;
lib/impala_cnn.py:177-184 + lib/util.py:61-62: dense = LayerNorm -> Linear -> ReLU).  The edge-class tables come from the product's own
packing code (packing.edge_tap_matrix), so the class / tap conventions the kernels rely on are what is tested."""
import torch
import torch.nn.functional as F

import vpt_amd  # noqa: F401
from vpt_amd import packing

D = torch.float64
EPS = 1e-5


def _gn1(t, g, b):
    mu = t.mean(dim=(1, 2, 3), keepdim=True)
    var = t.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
    return (t - mu) * torch.rsqrt(var + EPS) * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)


def _edge_table(w_oc9):
    """[Cout, 9 taps] per-tap channel sums -> [9 edge classes, Cout] sums over the taps that read inside the image."""
    m = packing.edge_tap_matrix("cpu", D)               # [9 classes, 9 taps]
    return m @ w_oc9.t()


def _per_pixel(tab, h, w):
    """[9, Cout] edge-class table -> [Cout, h, w]: the value of each pixel's class (e = 3 * row class + column class)."""
    ey = torch.tensor([0 if y == 0 else (2 if y == h - 1 else 1) for y in range(h)])
    ex = torch.tensor([0 if x == 0 else (2 if x == w - 1 else 1) for x in range(w)])
    e = ey.view(-1, 1) * 3 + ex.view(1, -1)             # [h, w]
    return tab[e].permute(2, 0, 1)


def test_group_norm_n_folded_into_block0_algebra():
    g = torch.Generator().manual_seed(3)
    f, c, co, h, w = 3, 32, 48, 10, 12
    P = torch.relu(torch.randn(f, c, h, w, generator=g, dtype=D)) * 1.7          # the pooled tensor (post-ReLU maxima)
    gam, bet = 1 + 0.3 * torch.randn(c, generator=g, dtype=D), 0.2 * torch.randn(c, generator=g, dtype=D)       # n
    g0, b0 = 1 + 0.3 * torch.randn(c, generator=g, dtype=D), 0.2 * torch.randn(c, generator=g, dtype=D)         # block 0 conv0's norm
    W0 = torch.randn(co, c, 3, 3, generator=g, dtype=D) / (9 * c) ** 0.5
    g1, b1 = 1 + 0.3 * torch.randn(co, generator=g, dtype=D), 0.2 * torch.randn(co, generator=g, dtype=D)       # conv1's norm (Cin = co)
    W1 = torch.randn(c, co, 3, 3, generator=g, dtype=D) / (9 * co) ** 0.5        # back to c channels: the residual adds x

    # ---- reference order ----
    x = _gn1(P, gam, bet)
    y_ref = torch.relu(F.conv2d(_gn1(x, g0, b0), W0, padding=1))
    out_ref = torch.relu(F.conv2d(_gn1(y_ref, g1, b1), W1, padding=1)) + x

    # ---- folded: the producer stores Q = gamma * P and the per-channel sums of Q; x is never formed ----
    hw = h * w
    Q = P * gam.view(1, -1, 1, 1)
    mu_p = P.mean(dim=(1, 2, 3))
    r_p = torch.rsqrt(P.var(dim=(1, 2, 3), unbiased=False) + EPS)
    kappa = r_p * mu_p
    bch = bet.view(1, -1) - kappa.view(-1, 1) * gam.view(1, -1)                   # b[f][c]:  x = r_P Q + b
    S1, S2 = Q.sum(dim=(2, 3)), (Q * Q).sum(dim=(2, 3))                           # chs_out of the producers
    sum_x = (r_p.view(-1, 1) * S1 + hw * bch).sum(1)
    sum_x2 = (r_p.view(-1, 1) ** 2 * S2 + 2 * r_p.view(-1, 1) * bch * S1 + hw * bch ** 2).sum(1)
    mu_x = sum_x / (c * hw)
    r_x = torch.rsqrt(sum_x2 / (c * hw) - mu_x ** 2 + EPS)
    assert torch.allclose(mu_x, x.mean(dim=(1, 2, 3)), rtol=1e-12, atol=1e-12)
    assert torch.allclose(r_x, torch.rsqrt(x.var(dim=(1, 2, 3), unbiased=False) + EPS), rtol=1e-10)

    Wp = W0 * g0.view(1, -1, 1, 1)                                                # W' (the packed weights, unrounded here)
    SA = _edge_table((W0 * b0.view(1, -1, 1, 1)).sum(1).reshape(co, 9))           # the layer's own tables (packing.pack_conv3x3's edge_sa / edge_sg)
    SG = _edge_table(Wp.sum(1).reshape(co, 9))
    TB = _edge_table((Wp * bet.view(1, -1, 1, 1)).sum(1).reshape(co, 9))          # engine._nfold_tables
    TG = _edge_table((Wp * gam.view(1, -1, 1, 1)).sum(1).reshape(co, 9))
    v = lambda t: t.view(-1, 1, 1, 1)
    kk = (_per_pixel(SA, h, w).unsqueeze(0) + v(r_x) * _per_pixel(TB, h, w).unsqueeze(0) - v(r_x * kappa) * _per_pixel(TG, h, w).unsqueeze(0)
          - v(r_x * mu_x) * _per_pixel(SG, h, w).unsqueeze(0))                    # kk_frame, per pixel
    y_fold = torch.relu(v(r_x * r_p) * F.conv2d(Q, Wp, padding=1) + kk)           # rs_frame * acc + kk_frame
    assert torch.allclose(y_fold, y_ref, rtol=1e-9, atol=1e-9), float((y_fold - y_ref).abs().max())

    # conv1: the ordinary fold on y (statistics from conv0's epilogue) + the residual as res_scale[f] * Q + res_bias[f][c]
    mu_y = y_fold.mean(dim=(1, 2, 3))
    r_y = torch.rsqrt(y_fold.var(dim=(1, 2, 3), unbiased=False) + EPS)
    W1p = W1 * g1.view(1, -1, 1, 1)
    SA1 = _edge_table((W1 * b1.view(1, -1, 1, 1)).sum(1).reshape(c, 9))
    SG1 = _edge_table(W1p.sum(1).reshape(c, 9))
    conv1 = torch.relu(v(r_y) * F.conv2d(y_fold, W1p, padding=1) - v(r_y * mu_y) * _per_pixel(SG1, h, w).unsqueeze(0) + _per_pixel(SA1, h, w).unsqueeze(0))
    out_fold = conv1 + v(r_p) * Q + bch.view(f, c, 1, 1)
    assert torch.allclose(out_fold, out_ref, rtol=1e-9, atol=1e-9), float((out_fold - out_ref).abs().max())


def test_dense_layer_norm_folded_into_the_gemm_algebra():
    g = torch.Generator().manual_seed(4)
    f, k, n = 5, 384, 24
    x = torch.relu(torch.randn(f, k, generator=g, dtype=D)) * 2.0                # the last block's output, flattened per frame
    gain, bias = 1 + 0.2 * torch.randn(k, generator=g, dtype=D), 0.1 * torch.randn(k, generator=g, dtype=D)     # dense.norm (per element)
    W, b = torch.randn(n, k, generator=g, dtype=D) / k ** 0.5, 0.1 * torch.randn(n, generator=g, dtype=D)
    ref = torch.relu(F.linear(F.layer_norm(x, (k,), gain, bias, EPS), W, b))
    # folded: GEMM on the raw tensor with W * gain; the split-K reduction applies r (acc - mu SG[n]) + SB[n] and the ReLU
    mu = x.mean(1)
    r = torch.rsqrt(x.var(1, unbiased=False) + EPS)
    Wg = W * gain.view(1, -1)
    SG, SB = Wg.sum(1), W @ bias + b
    parts = torch.stack([x[:, i::4] @ Wg[:, i::4].t() for i in range(4)])         # any split of K: the partial sums add up
    out = torch.relu(r.view(-1, 1) * (parts.sum(0) - mu.view(-1, 1) * SG.view(1, -1)) + SB.view(1, -1))
    assert torch.allclose(out, ref, rtol=1e-10, atol=1e-10), float((out - ref).abs().max())


def test_packing_edge_tables_are_the_fold_tables():
    """packing.pack_conv3x3's (edge_sa, edge_sg) -- what the kernels' epilogues read -- equal the tables of the algebra above built from the
    ROUNDED packed weights (the kernels multiply rounded weights, so the tables must be sums of the same numbers)."""
    g = torch.Generator().manual_seed(5)
    c, co = 64, 96
    W = torch.randn(co, c, 3, 3, generator=g) * 0.05
    gain, bias = 1 + 0.2 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    for dt in (torch.bfloat16, torch.float16):
        _, sa, sg = packing.pack_conv3x3(W, gain, bias, dtype=dt)
        Wp = (W * gain.view(1, -1, 1, 1)).to(dt).double()
        SG = _edge_table(Wp.sum(1).reshape(co, 9))
        SA = _edge_table((W.double() * bias.double().view(1, -1, 1, 1)).sum(1).reshape(co, 9))
        assert torch.allclose(sg[:, :co].double(), SG, rtol=1e-5, atol=1e-6)
        assert torch.allclose(sa[:, :co].double(), SA, rtol=1e-5, atol=1e-6)

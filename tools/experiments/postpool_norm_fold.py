"""Worked check (CPU, fp64) of the design in DESIGN.md section 10 (1): the post-pool GroupNorm `n` of a CnnDownStack
(lib/impala_cnn.py:99-100,118-119) without a pass of its own.

    x    = n(P)                = gamma_c (P - mu_P) r_P + beta_c                     (P: the pooled tensor; mu_P, r_P: its frame statistics)
    conv0 input norm0(x)      = g0_c (x - mu_x) r_x + b0_c                           (mu_x, r_x: frame statistics of x)
                              = (g0_c gamma_c) (r_x r_P) P + [ g0_c beta_c r_x - g0_c gamma_c (r_x r_P mu_P) - g0_c (r_x mu_x) + b0_c ]

so  conv(W, norm0(x)) = (r_x r_P) conv(W'', P) + r_x T1[e] - (r_x r_P mu_P) T2[e] - (r_x mu_x) T3[e] + T4[e]   with  W'' = W g0 gamma  and four
edge tables (sums over the taps that are inside the image for edge class e):  T1 = sum W g0 beta, T2 = sum W'', T3 = sum W g0, T4 = sum W b0,
and mu_x, r_x follow from PER-CHANNEL sums of P:   S1_c = sum P_c,  S2_c = sum P_c^2  (N = H W pixels, C channels)
    mu_x   = 1 / (C N) sum_c [ gamma_c r_P (S1_c - N mu_P) + beta_c N ]
    E[x^2] = 1 / (C N) sum_c [ gamma_c^2 r_P^2 (S2_c - 2 mu_P S1_c + N mu_P^2) + 2 gamma_c r_P beta_c (S1_c - N mu_P) + beta_c^2 N ]
The residual x of block 0 is a_c P + b_c with a_c = gamma_c r_P, b_c = beta_c - gamma_c r_P mu_P (conv1's epilogue).

    python tools/experiments/postpool_norm_fold.py        # prints the max error of the folded form against the direct one
"""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
D = torch.float64
C, Co, H, W_ = 32, 48, 12, 10
EPS = 1e-5
P = torch.relu(torch.randn(2, C, H, W_, dtype=D)) * 1.7
gamma, beta = 1 + 0.3 * torch.randn(C, dtype=D), 0.2 * torch.randn(C, dtype=D)
g0, b0 = 1 + 0.3 * torch.randn(C, dtype=D), 0.2 * torch.randn(C, dtype=D)
Wt = torch.randn(Co, C, 3, 3, dtype=D) / (9 * C) ** 0.5


def gn1(t, g, b):
    mu = t.mean(dim=(1, 2, 3), keepdim=True)
    var = t.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
    return (t - mu) * torch.rsqrt(var + EPS) * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)


# ---- direct: the reference's order (norm, zero-pad, conv) ----
x = gn1(P, gamma, beta)
direct = F.conv2d(gn1(x, g0, b0), Wt, padding=1)

# ---- folded ----
N = H * W_
mu_P = P.mean(dim=(1, 2, 3))
r_P = torch.rsqrt(P.var(dim=(1, 2, 3), unbiased=False) + EPS)
S1, S2 = P.sum(dim=(2, 3)), (P * P).sum(dim=(2, 3))                      # [F, C]: the per-channel sums the pool kernel would emit
d1 = S1 - N * mu_P[:, None]
mu_x = ((gamma * r_P[:, None]) * d1 + beta * N).sum(1) / (C * N)
ex2 = ((gamma * r_P[:, None]) ** 2 * (S2 - 2 * mu_P[:, None] * S1 + N * mu_P[:, None] ** 2) + 2 * gamma * r_P[:, None] * beta * d1 + beta ** 2 * N).sum(1) / (C * N)
r_x = torch.rsqrt(ex2 - mu_x ** 2 + EPS)
assert torch.allclose(mu_x, x.mean(dim=(1, 2, 3))) and torch.allclose(r_x, torch.rsqrt(x.var(dim=(1, 2, 3), unbiased=False) + EPS))

W2 = Wt * (g0 * gamma).view(1, C, 1, 1)
ones = torch.ones(1, 1, H, W_, dtype=D)


def table(wc):                       # per-pixel sum over the taps inside the image of sum_c wc[o, c, tap]  -> [Co, H, W]: nine distinct values (edge classes)
    return F.conv2d(ones, wc.sum(1, keepdim=True), padding=1)[0]


T1, T2, T3, T4 = table(Wt * (g0 * beta).view(1, C, 1, 1)), table(W2), table(Wt * g0.view(1, C, 1, 1)), table(Wt * b0.view(1, C, 1, 1))
f_ = lambda v: v.view(-1, 1, 1, 1)
folded = f_(r_x * r_P) * F.conv2d(P, W2, padding=1) + f_(r_x) * T1 - f_(r_x * r_P * mu_P) * T2 - f_(r_x * mu_x) * T3 + T4
print("conv0 pre-activation: max |folded - direct| =", float((folded - direct).abs().max()), " (max |direct| =", float(direct.abs().max()), ")")
a_c, b_c = gamma * r_P[:, None], beta - gamma * r_P[:, None] * mu_P[:, None]
print("residual x = a_c P + b_c:  max error =", float((a_c[:, :, None, None] * P + b_c[:, :, None, None] - x).abs().max()))
edge = torch.stack([T2[0, y, x_] for y in (0, 1, H - 1) for x_ in (0, 1, W_ - 1)])
print("nine edge classes of T2[0]:", [round(float(v), 4) for v in edge])

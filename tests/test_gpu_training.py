"""Backward kernels and the BC step (every layer, CNN included) against torch autograd through the CPU oracle, in BOTH operand
formats (fp16 = the parity mode, with loss scaling; bf16).  Needs an MI355X.  Gradients pass through 16-bit MFMA GEMMs (fp32
accumulate): per-kernel bounds are relative L2 on identical inputs (3e-2; typically 0.5-1.5e-2), fp32 kernels (LN / attention
backward) 1e-3; end-to-end bounds per format are tests/parity.py:GRAD_BOUNDS, calibrated by the CPU emulator of the rounding points."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import vpt_amd  # noqa: E402,F401
from vpt_amd import ops, packing  # noqa: E402
from vpt_amd.training import BCTrainer, linear_backward  # noqa: E402
from vpt_amd.lib.policy import MinecraftAgentPolicy  # noqa: E402
from vpt_amd.lib.types import minecraft_action_space  # noqa: E402
from oracle import vpt_oracle as O  # noqa: E402
from tests import parity as P  # noqa: E402

DEV = "cuda"


def _l2(a, ref):
    return float((a - ref).norm() / ref.norm().clamp(min=1e-30))


def test_nll_backward():
    g = torch.Generator().manual_seed(1)
    m, nb, nc, temp = 6, 8641, 121, 2.0
    zb = (torch.randn(m, nb, generator=g) * 2).requires_grad_(True)
    zc = (torch.randn(m, nc, generator=g) * 2).requires_grad_(True)
    ab = torch.randint(0, nb, (m,), generator=g)
    ac = torch.randint(0, nc, (m,), generator=g)
    lb, lc = torch.log_softmax(zb / temp, -1), torch.log_softmax(zc / temp, -1)
    loss = -(lb.gather(1, ab[:, None]) + lc.gather(1, ac[:, None])).sum() / 24.0
    gb, gc = torch.autograd.grad(loss, [zb, zc])
    dz = ops.nll_backward(lb.detach().to(DEV), lc.detach().to(DEV), ab.to(DEV), ac.to(DEV), 8768, 1.0 / (24.0 * temp))
    torch.cuda.synchronize()
    dz = dz.cpu().float()
    assert _l2(dz[:, :nb], gb) < 6e-3 and _l2(dz[:, nb:nb + nc], gc) < 6e-3
    assert float(dz[:, nb + nc:].abs().max()) == 0.0


@pytest.mark.parametrize("m,d,relu_in", [(70, 2048, False), (33, 256, True), (5, 1024, True), (9, 3072, False), (40, 4096, True), (7, 1000, False)])
def test_layernorm_backward(m, d, relu_in):
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(m, d, generator=g) * 1.5 + 0.3).requires_grad_(True)
    gain = (1 + 0.2 * torch.randn(d, generator=g)).requires_grad_(True)
    bias = (0.1 * torch.randn(d, generator=g)).requires_grad_(True)
    dy = torch.randn(m, d, generator=g)
    add = torch.randn(m, d, generator=g)
    y = O.layer_norm(torch.relu(x) if relu_in else x, gain, bias)
    gx, gg, gb = torch.autograd.grad((y * dy).sum(), [x, gain, bias])
    dg, db = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    dx = ops.layernorm_backward(x.detach().to(DEV), gain.detach().to(DEV), dy.to(DEV), dg, db, relu_in=relu_in, dx_add=add.to(DEV))
    torch.cuda.synchronize()
    assert _l2(dx.cpu() - add, gx) < 1e-4
    assert _l2(dg.cpu(), gg) < 1e-4 and _l2(db.cpu(), gb) < 1e-4


def test_linear_backward_and_colsum():
    g = torch.Generator().manual_seed(3)
    m, n, k = 200, 300, 256
    x = torch.randn(m, k, generator=g)
    W = torch.randn(n, k, generator=g) / k ** 0.5
    dy = torch.randn(m, n, generator=g)
    np_ = 320
    dy16 = ops.gate_cast(dy.to(DEV), np_)
    x16 = x.to(torch.bfloat16).to(DEV)
    dx, _, dw = linear_backward(dy16, n, x16, W.to(DEV))
    bsum = torch.zeros(n, device=DEV)
    ops.column_sum_(bsum, dy16, n)
    torch.cuda.synchronize()
    dyb, xb, Wb = dy.to(torch.bfloat16).float(), x.to(torch.bfloat16).float(), W.to(torch.bfloat16).float()
    assert _l2(dx.cpu(), dyb @ Wb) < 3e-3
    assert _l2(dw.cpu(), dyb.t() @ xb) < 3e-3
    assert _l2(bsum.cpu(), dyb.sum(0)) < 1e-3


@pytest.mark.parametrize("bsz,t,first_flags,heads", [(2, 70, [False, True], 2), (1, 128, [False], 2), (2, 5, [False, False], 2),
                                                     (16, 128, [False] * 15 + [True], 16)])   # 1024 workgroups, several per CU: barrier races show here
def test_attention_backward(bsz, t, first_flags, heads):
    g = torch.Generator().manual_seed(4)
    maxlen = 128
    hid = heads * 128
    ld = 3 * hid + 10 * heads
    qkvr = torch.randn(bsz * t, ld, generator=g)
    qkvr[:, :hid] *= 2.0
    qkvr.requires_grad_(True)
    kmem = torch.randn(bsz, maxlen, hid, generator=g)
    vmem = torch.randn(bsz, maxlen, hid, generator=g)
    state_mask = torch.rand(bsz, 1, maxlen, generator=g) > 0.3
    first_b = torch.tensor(first_flags)
    b_nd = (0.5 * torch.randn(10, maxlen, generator=g)).requires_grad_(True)
    dout = torch.randn(bsz * t, hid, generator=g)
    q = qkvr[:, :hid].reshape(bsz, t, heads, 128).permute(0, 2, 1, 3)
    k_full = torch.cat([kmem, qkvr[:, hid:2 * hid].reshape(bsz, t, hid)], 1)
    v_full = torch.cat([vmem, qkvr[:, 2 * hid:3 * hid].reshape(bsz, t, hid)], 1)
    kh = k_full.reshape(bsz, -1, heads, 128).permute(0, 2, 1, 3)
    vh = v_full.reshape(bsz, -1, heads, 128).permute(0, 2, 1, 3)
    logits = q @ kh.transpose(-1, -2) / 128.0
    vis, _ = O.band_visibility(t, maxlen, first_b, state_mask)
    logits = logits + (~vis).float().unsqueeze(1) * O.NEG_MASK
    logits = logits + O.rel_pos_bias(qkvr[:, 3 * hid:].reshape(bsz, t, heads, 10), b_nd, t, maxlen)
    out = (torch.softmax(logits, -1) @ vh).permute(0, 2, 1, 3).reshape(bsz * t, hid)
    gq, gb = torch.autograd.grad((out * dout).sum(), [qkvr, b_nd])
    memvalid = (state_mask & ~first_b.view(bsz, 1, 1)).reshape(bsz, maxlen).to(torch.uint8)
    db = torch.zeros(10, maxlen, device=DEV)
    dq = ops.masked_attention_backward(qkvr.detach().to(DEV), kmem.to(DEV), vmem.to(DEV), memvalid.to(DEV), b_nd.detach().to(DEV),
                                       dout.to(DEV), db, bsz, t, heads, hid)
    torch.cuda.synchronize()
    dq = dq.cpu()
    for name, sl in [("dQ", slice(0, hid)), ("dK", slice(hid, 2 * hid)), ("dV", slice(2 * hid, 3 * hid)), ("dR", slice(3 * hid, ld))]:
        err = _l2(dq[:, sl], gq[:, sl])
        assert err < 1e-3, f"{name} rel L2 {err}"
    assert _l2(db.cpu(), gb) < 1e-3


@pytest.fixture(scope="module", params=["bf16", "fp16"])
def trainer_1x(request):
    pk = O.policy_kwargs_for("1x")
    cfg = O.config_from_policy_kwargs(pk, dict(temperature=2.0))
    sd = O.synthetic_state_dict(cfg, seed=0)
    pol = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0), precision=request.param)
    pol.load_state_dict(sd, strict=False)
    return pol.to(DEV), cfg, sd


_ORACLE_CACHE = {}


@pytest.mark.parametrize("train_cnn", [False, True])
def test_bc_gradients_vs_oracle(trainer_1x, train_cnn):
    """Every trainable tensor's gradient against the fp32 oracle pinned to the reference (cosine >= 0.93, norm within
    25 %) and against autograd through the bf16-emulating oracle.  A bf16 forward flips ~1-2 % of the ReLU gates,
    which alone moves gradients 15-30 % in relative L2 even with exact autograd (reproduced on the CPU emulation,
    oracle/vpt_oracle_bf16.py); the per-kernel tests above check the backward math itself at 1e-3 on identical inputs."""
    from oracle import vpt_oracle_bf16 as OB
    pol, cfg, sd = trainer_1x
    mode = pol.precision
    tr = BCTrainer(pol, train_cnn=train_cnn)
    b, t = 2, 6
    g = torch.Generator().manual_seed(5)
    img = torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8)
    first = torch.zeros(b, t, dtype=torch.bool)
    ab = torch.randint(0, 8641, (b, t), generator=g)
    ac = torch.randint(0, 121, (b, t), generator=g)
    torch.set_num_threads(max(1, min(32, len(__import__("os").sched_getaffinity(0)))))
    if "ref" not in _ORACLE_CACHE:
        _ORACLE_CACHE["ref"] = O.bc_loss_and_grads(sd, cfg, img, first, O.initial_state(cfg, b), ab, ac)[:2]
    if mode not in _ORACLE_CACHE:      # autograd through the emulator of THIS format's rounding points (same gates as the GPU, up to summation order)
        _ORACLE_CACHE[mode] = OB.bc_loss_and_grads(sd, cfg, img, first, O.initial_state(cfg, b), ab, ac, rnd=OB.Rounding(mode, mode, mode))[:2]
    (loss_ref, grads_ref), (loss_em, grads_em) = _ORACLE_CACHE["ref"], _ORACLE_CACHE[mode]
    loss, grads, _ = tr.loss_and_grads(img.to(DEV), first.to(DEV), pol.initial_state(b), ab.to(DEV), ac.to(DEV))
    torch.cuda.synchronize()
    assert abs(float(loss) - loss_ref) < 2e-2 and abs(float(loss) - loss_em) < 1e-2, (float(loss), loss_ref, loss_em)
    l2_em, cos_ref = {}, {}
    for name in tr.trainable:
        ref, em = grads_ref[name], grads_em[name]
        if float(ref.norm()) == 0.0:
            continue
        mine = grads[name].cpu().reshape(ref.shape)
        l2_em[name] = _l2(mine, em)
        cos_ref[name] = float((mine * ref).sum() / (mine.norm() * ref.norm()))
    print(f"PARITY[{mode}] BC grads vs the {mode}-emulating oracle: worst rel-L2", sorted(l2_em.items(), key=lambda kv: -kv[1])[:4])
    print(f"PARITY[{mode}] BC grads vs fp32 oracle: worst cosine", sorted(cos_ref.items(), key=lambda kv: kv[1])[:4])
    assert len(l2_em) >= (125 if train_cnn else 60)
    # the CPU emulation and the GPU round at the same points but sum in different orders, so ~1 % of the ReLU
    # gates still differ (tools/bc_grad_diag.py): the bound that holds is on direction and norm, not on L2.
    # Yardstick: the emulation's own distance to the fp32 oracle (0.2-0.3 in the trunk, 0.3-0.64 in the CNN, whose
    # gradients pass through up to 15 more ReLU / max-pool layers) -- the GPU must not be further away than that.
    worst = {}
    for name in l2_em:
        ref, em = grads_ref[name], grads_em[name]
        mine = grads[name].cpu().reshape(ref.shape)
        d_gpu, d_em = _l2(mine, ref), _l2(em, ref)
        worst[name] = (d_gpu, d_em)
        # per tensor the two noise realisations differ (which gates flip depends on the summation order), so the
        # per-tensor bound is loose and the tight statement is the average over all tensors below
        assert d_gpu < 1.5 * d_em + 0.1, (name, d_gpu, d_em)
        cos_em = float((em * ref).sum() / (em.norm() * ref.norm()))
        assert cos_ref[name] > min(0.93, cos_em - 0.15), (name, cos_ref[name], cos_em)
        ratio = float(mine.norm() / ref.norm())
        assert 0.75 < ratio < 1.3, (name, ratio)
    print("PARITY BC grads: largest (GPU-vs-fp32, emulation-vs-fp32) rel-L2", sorted(worst.items(), key=lambda kv: -kv[1][0])[:3])
    mean_gpu = sum(v[0] for v in worst.values()) / len(worst)
    mean_em = sum(v[1] for v in worst.values()) / len(worst)
    mean_cos = sum(cos_ref.values()) / len(cos_ref)
    print(f"PARITY[{mode}] BC grads: mean rel-L2 to the fp32 oracle over {len(worst)} tensors: GPU {mean_gpu:.3f}, {mode} emulation {mean_em:.3f}; "
          f"cosine to the fp32 oracle: mean {mean_cos:.4f}, worst {min(cos_ref.values()):.3f}")
    assert mean_gpu < 1.15 * mean_em + 0.02, (mean_gpu, mean_em)
    GB = P.GRAD_BOUNDS[mode]                       # absolute, per format (not relative to our own emulator)
    assert mean_gpu < GB["l2_mean"] and mean_cos > GB["cos_mean"] and min(cos_ref.values()) > GB["cos_min_small"], \
        (mean_gpu, mean_cos, min(cos_ref.values()))
    bad = {k: v for k, v in l2_em.items() if v > (0.75 if "cnn" in k else 0.4)}
    assert not bad, bad


@pytest.mark.parametrize("train_cnn", [False, True])
def test_bc_step_reduces_loss(trainer_1x, train_cnn):
    """A few Adam steps on a fixed batch from the seeded weights (restored first: the fixture is shared) lower the loss."""
    pol, cfg, sd = trainer_1x
    pol.load_state_dict(sd, strict=False)
    tr = BCTrainer(pol, lr=1e-4, weight_decay=0.0, train_cnn=train_cnn)
    b, t = 2, 4
    g = torch.Generator().manual_seed(6)
    img = torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8).to(DEV)
    first = torch.zeros(b, t, dtype=torch.bool, device=DEV)
    ab = torch.randint(0, 8641, (b, t), generator=g).to(DEV)
    ac = torch.randint(0, 121, (b, t), generator=g).to(DEV)
    losses = []
    try:
        for _ in range(6):
            loss, _ = tr.step(img, first, pol.initial_state(b), ab, ac)
            losses.append(loss)
        torch.cuda.synchronize()
    finally:
        pol.load_state_dict(sd, strict=False)
    print("BC losses on a fixed batch:", [round(l, 3) for l in losses])
    assert losses[-1] < losses[0] - 0.2 and all(b_ < a_ + 0.05 for a_, b_ in zip(losses, losses[1:]))


# ---------------------------------------------------------------------------------------------------------
# CNN backward pieces
# ---------------------------------------------------------------------------------------------------------
def _stats_of(x_nchw):
    f = x_nchw.shape[0]
    flat = x_nchw.reshape(f, -1).double()
    return torch.stack([flat.sum(1), (flat * flat).sum(1)], dim=1).contiguous()


VALID = {0: [1, 2], 1: [0, 1, 2], 2: [0, 1]}


def _tap_sum(tab, cout):
    """tab [9 edge classes, >=cout] -> [cout, 3, 3]: sum over the edge classes in which tap (kh, kw) is inside the image."""
    out = torch.zeros(cout, 3, 3, dtype=tab.dtype)
    for ey in range(3):
        for ex in range(3):
            for kh in VALID[ey]:
                for kw in VALID[ex]:
                    out[:, kh, kw] += tab[ey * 3 + ex, :cout]
    return out


@pytest.mark.parametrize("per_element", [False, True])
def test_frame_affine_backward(per_element):
    g = torch.Generator().manual_seed(11)
    f, c, h, w = 3, 64, 16, 16
    x = torch.randn(f, c, h, w, generator=g).to(torch.bfloat16).float().requires_grad_(True)
    n_g = c * h * w if per_element else c
    gain = (1 + 0.2 * torch.randn(n_g, generator=g)).requires_grad_(True)
    bias = (0.1 * torch.randn(n_g, generator=g)).requires_grad_(True)
    dy = torch.randn(f, c, h, w, generator=g).to(torch.bfloat16).float()
    if per_element:
        y = O.layer_norm(x.reshape(f, -1), gain, bias).reshape(f, c, h, w)
        gk = packing.chw_to_blocked_vector(gain.detach(), c, h, w)
    else:
        y = O.group_norm_1(x, gain, bias)
        gk = gain.detach()
    gx, gg, gb = torch.autograd.grad((y * dy).sum(), [x, gain, bias])
    dg, db = torch.zeros(n_g, device=DEV), torch.zeros(n_g, device=DEV)
    dx = ops.frame_affine_backward(packing.nchw_to_blocked(x.detach()).to(DEV), packing.nchw_to_blocked(dy).to(DEV), gk.to(DEV),
                                   _stats_of(x.detach()).to(DEV), dg, db, per_element=per_element)
    torch.cuda.synchronize()
    assert _l2(packing.blocked_to_nchw(dx.cpu(), c, h, w), gx) < 8e-3
    if per_element:
        dg_c, db_c = dg.cpu(), db.cpu()
        assert _l2(dg_c, packing.chw_to_blocked_vector(gg, c, h, w)) < 1e-3 and _l2(db_c, packing.chw_to_blocked_vector(gb, c, h, w)) < 1e-3
    else:
        assert _l2(dg.cpu(), gg) < 1e-3 and _l2(db.cpu(), gb) < 1e-3


def test_maxpool_backward():
    g = torch.Generator().manual_seed(12)
    f, c, h, w = 2, 32, 16, 16
    pre = torch.relu(torch.randn(f, c, h, w, generator=g)).to(torch.bfloat16).float()
    pre[0, :, 4:8, 4:8] = 0.5  # plateaus: exercise the first-maximum tie rule
    pre.requires_grad_(True)
    pooled = torch.nn.functional.max_pool2d(pre, 3, 2, 1)
    dp = torch.randn(f, c, h // 2, w // 2, generator=g).to(torch.bfloat16).float()
    (gpre,) = torch.autograd.grad((pooled * dp).sum(), [pre])
    out = ops.maxpool_backward(packing.nchw_to_blocked(pre.detach()).to(DEV), packing.nchw_to_blocked(pooled.detach()).to(DEV),
                               packing.nchw_to_blocked(dp).to(DEV))
    torch.cuda.synchronize()
    assert _l2(packing.blocked_to_nchw(out.cpu(), c, h, w), gpre) < 8e-3


@pytest.mark.parametrize("frames,h,cin,cout,use_res", [(2, 16, 64, 128, True), (2, 32, 128, 96, False)])
def test_conv_layer_backward_dx_and_bias(frames, h, cin, cout, use_res):
    """GN -> conv3x3 -> ReLU (+res): input gradient (dgrad kernel + statistics terms) and GroupNorm-bias gradient
    (edge-table path) against autograd on the same bf16 inputs."""
    g = torch.Generator().manual_seed(13)
    w_ = h
    W = (torch.randn(cout, cin, 3, 3, generator=g) * (1.6 / (cin * 9) ** 0.5)).requires_grad_(True)
    gain = (1 + 0.2 * torch.randn(cin, generator=g)).requires_grad_(True)
    bias = (0.1 * torch.randn(cin, generator=g)).requires_grad_(True)
    x = (torch.relu(torch.randn(frames, cin, h, w_, generator=g)) + 0.2 * torch.randn(frames, cin, h, w_, generator=g)).to(torch.bfloat16).float().requires_grad_(True)
    res = torch.randn(frames, cout, h, w_, generator=g).to(torch.bfloat16).float() if use_res else None
    dY = torch.randn(frames, cout, h, w_, generator=g).to(torch.bfloat16).float()
    sd = {"norm.weight": gain, "norm.bias": bias, "layer.weight": W}
    y = O._norm_conv_relu(sd, "", x) + (res if use_res else 0)
    gx, gW, gg, gb = torch.autograd.grad((y * dY).sum(), [x, W, gain, bias])
    # GPU: forward (for the saved output), prepare, dgrad
    wpk, sa, sg = packing.pack_conv3x3(W.detach().to(DEV), gain.detach().to(DEV), bias.detach().to(DEV))
    xb = packing.nchw_to_blocked(x.detach()).to(DEV)
    st_in = _stats_of(x.detach()).to(DEV)
    resb = packing.nchw_to_blocked(res).to(DEV) if use_res else None
    yb = ops.conv3x3(xb, wpk, sa, sg, st_in, cout, res=resb)
    dacc, coef, d_sa, d_sg, t12 = ops.conv_backward_prepare(packing.nchw_to_blocked(dY).to(DEV), yb, resb, st_in, sa, sg, cin, want_t12=True)
    n = cin * h * w_
    from vpt_amd.training import conv_dgrad_coef
    coef_host = conv_dgrad_coef(st_in, t12, n)     # host restatement of the finish kernel's (c0, c1)
    assert _l2(coef.cpu(), coef_host.cpu()) < 1e-5
    dx = ops.conv3x3_dgrad(dacc, packing.pack_conv3x3_dgrad(W.detach().to(DEV), gain.detach().to(DEV)), cin, xin=xb, coef=coef)
    torch.cuda.synchronize()
    err = _l2(packing.blocked_to_nchw(dx.cpu(), cin, h, w_), gx)
    assert err < 8e-2, f"dx rel L2 {err}"  # a few ReLU gates differ between the GPU and CPU forwards
    dbeta = (_tap_sum(d_sa.cpu(), cout).unsqueeze(1) * W.detach()).sum(dim=(0, 2, 3))
    assert _l2(dbeta, gb) < 1e-1


@pytest.mark.parametrize("fmt", ["bf16", "fp16"])
@pytest.mark.parametrize("frames,h,c", [(3, 16, 64), (2, 32, 128), (2, 64, 32), (1, 16, 160)])
def test_block_backward_gated_dgrad_equals_the_two_step_path(frames, h, c, fmt):
    """Round 5: conv1's dgrad writes conv0's backward operand directly (vpt_conv3x3_dgrad_gated = vpt_conv3x3_kernel mode 6) and
    vpt_conv_backward_reduce replaces conv0's per-element prepare pass.  On IDENTICAL inputs -- a CnnBasicBlock x + conv1(conv0(x)),
    lib/impala_cnn.py:50-52 -- the pair must reproduce what dgrad -> prepare gave: the operand dacc0 (the same values up to ONE 16-bit
    rounding: the old path rounds dy, then rstd * dy; the new one rounds once), the coefficients (c0, c1), T1 / T2 and the edge-table sums,
    and the block's input gradient."""
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[fmt]
    g = torch.Generator().manual_seed(131 + h + c)
    mk = lambda: ((torch.randn(c, c, 3, 3, generator=g) * (1.6 / (c * 9) ** 0.5)).to(DEV), (1 + 0.2 * torch.randn(c, generator=g)).to(DEV), (0.1 * torch.randn(c, generator=g)).to(DEV))
    (W0, g0, b0), (W1, g1, b1) = mk(), mk()
    x = (torch.relu(torch.randn(frames, c, h, h, generator=g)) + 0.2 * torch.randn(frames, c, h, h, generator=g)).to(dt).float()
    dout = (torch.randn(frames, c, h, h, generator=g) * (1e-2 if fmt == "fp16" else 1.0)).to(dt).float()
    wpk0, sa0, sg0 = packing.pack_conv3x3(W0, g0, b0, dtype=dt)
    wpk1, sa1, sg1 = packing.pack_conv3x3(W1, g1, b1, dtype=dt)
    xb = packing.nchw_to_blocked(x, dtype=dt).to(DEV)
    st_x = _stats_of(x).to(DEV)
    st_y = torch.zeros(frames, 2, dtype=torch.float64, device=DEV)
    yb = ops.conv3x3(xb, wpk0, sa0, sg0, st_x, c, stats_out=st_y)
    ob = ops.conv3x3(yb, wpk1, sa1, sg1, st_y, c, res=xb)
    doutb = packing.nchw_to_blocked(dout, dtype=dt).to(DEV)
    wt0, wt1 = packing.pack_conv3x3_dgrad(W0, g0, dtype=dt), packing.pack_conv3x3_dgrad(W1, g1, dtype=dt)
    # conv1 (residual layer): common to both paths
    dacc1, coef1, _, _ = ops.conv_backward_prepare(doutb, ob, xb, st_y, sa1, sg1, c)
    # two-step: plain dgrad, then conv0's prepare
    dy = ops.conv3x3_dgrad(dacc1, wt1, c, xin=yb, coef=coef1)
    dacc0_ref, coef0_ref, dsa_ref, dsg_ref, t12_ref = ops.conv_backward_prepare(dy, yb, None, st_x, sa0, sg0, c, want_t12=True)
    dx_ref = ops.conv3x3_dgrad(dacc0_ref, wt0, c, skip=doutb, xin=xb, coef=coef0_ref)
    # fused: gated dgrad, then the reduction
    dacc0, gate_u = ops.conv3x3_dgrad_gated(dacc1, wt1, c, yb, coef1, st_x, c)
    coef0, dsa, dsg, t12 = ops.conv_backward_reduce(dacc0, gate_u, st_x, sa0, sg0, c, want_t12=True)
    dx = ops.conv3x3_dgrad(dacc0, wt0, c, skip=doutb, xin=xb, coef=coef0)
    torch.cuda.synchronize()
    eps = 2.0 ** -8 if fmt == "bf16" else 2.0 ** -11
    a, r = dacc0.float(), dacc0_ref.float()
    assert torch.equal(a == 0, r == 0) or float(((a == 0) != (r == 0)).float().mean()) < 1e-4     # same gates (a value may round to 0 on one side only)
    atol = 2e-7 if fmt == "fp16" else 1e-30          # (IEEE half: values below 6e-5 are subnormal, their step is 6e-8)
    assert float(((a - r).abs() <= 3.5 * eps * r.abs() + atol).float().mean()) > 0.999, float((a - r).abs().max())
    e = dict(dacc=_l2(a, r), t12=_l2(t12, t12_ref), coef=_l2(coef0, coef0_ref), dsa=_l2(dsa, dsa_ref), dsg=_l2(dsg, dsg_ref), dx=_l2(dx.float(), dx_ref.float()))
    print(f"PARITY gated dgrad vs dgrad + prepare [{fmt}] {frames}x{c}x{h}x{h}: " + " ".join(f"{k} {v:.2e}" for k, v in e.items()))
    tol = 4 * eps
    assert e["dacc"] < tol and e["dx"] < 2 * tol, e
    assert e["t12"] < 4 * tol and e["coef"] < 4 * tol and e["dsa"] < 4 * tol and e["dsg"] < 4 * tol, e


@pytest.mark.parametrize("frames,h,cin,cout", [(3, 16, 64, 96), (2, 32, 32, 128), (1, 64, 64, 32)])
def test_conv_wgrad_kernel(frames, h, cin, cout):
    g = torch.Generator().manual_seed(14)
    dacc = torch.randn(frames, cout, h, h, generator=g).to(torch.bfloat16).float()
    x = torch.randn(frames, cin, h, h, generator=g).to(torch.bfloat16).float()
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    ref = torch.zeros(cout, 9, cin)
    for kh in range(3):
        for kw in range(3):
            ref[:, kh * 3 + kw, :] = torch.einsum("foyx,fcyx->oc", dacc, xp[:, :, kh:kh + h, kw:kw + h])
    out = ops.conv3x3_wgrad(packing.nchw_to_blocked(dacc).to(DEV), packing.nchw_to_blocked(x).to(DEV))
    torch.cuda.synchronize()
    err = _l2(out.cpu(), ref)
    assert err < 2e-3, f"wgrad rel L2 {err}"


def test_conv_layer_param_grads():
    """dW, dgain, dbias of GN -> conv3x3 -> ReLU + res through prepare + wgrad + host mapping vs autograd."""
    from vpt_amd.training import conv_param_grads
    g = torch.Generator().manual_seed(15)
    frames, h, cin, cout = 3, 16, 64, 128
    W = (torch.randn(cout, cin, 3, 3, generator=g) * (1.6 / (cin * 9) ** 0.5)).requires_grad_(True)
    gain = (1 + 0.2 * torch.randn(cin, generator=g)).requires_grad_(True)
    bias = (0.1 * torch.randn(cin, generator=g)).requires_grad_(True)
    x = (torch.relu(torch.randn(frames, cin, h, h, generator=g)) + 0.2 * torch.randn(frames, cin, h, h, generator=g)).to(torch.bfloat16).float()
    res = torch.randn(frames, cout, h, h, generator=g).to(torch.bfloat16).float()
    dY = torch.randn(frames, cout, h, h, generator=g).to(torch.bfloat16).float()
    y = O._norm_conv_relu({"norm.weight": gain, "norm.bias": bias, "layer.weight": W}, "", x) + res
    gW, gg, gb = torch.autograd.grad((y * dY).sum(), [W, gain, bias])
    Wd, gd, bd = W.detach().to(DEV), gain.detach().to(DEV), bias.detach().to(DEV)
    wpk, sa, sg = packing.pack_conv3x3(Wd, gd, bd)
    xb, resb = packing.nchw_to_blocked(x).to(DEV), packing.nchw_to_blocked(res).to(DEV)
    st_in = _stats_of(x).to(DEV)
    yb = ops.conv3x3(xb, wpk, sa, sg, st_in, cout, res=resb)
    dacc, coef, d_sa, d_sg = ops.conv_backward_prepare(packing.nchw_to_blocked(dY).to(DEV), yb, resb, st_in, sa, sg, cin)
    dw_raw = ops.conv3x3_wgrad(dacc, xb)
    dW, dgain, dbias = conv_param_grads(dw_raw, d_sa, d_sg, Wd, gd, bd)
    torch.cuda.synchronize()
    eW, eg, eb = _l2(dW.cpu(), gW), _l2(dgain.cpu(), gg), _l2(dbias.cpu(), gb)
    print(f"PARITY conv layer param grads: dW {eW:.3e} dgain {eg:.3e} dbias {eb:.3e}")
    assert eW < 6e-2 and eg < 1e-1 and eb < 1e-1


@pytest.mark.parametrize("fmt", ["bf16", "fp16"])
@pytest.mark.parametrize("frames,cout,h,w", [(2, 128, 128, 128), (1, 64, 128, 128), (12, 128, 128, 128), (7, 64, 32, 80),
                                             (3, 192, 64, 64)])   # 12 frames = 768 tiles: every persistent workgroup sweeps several; 192 channels (the 3x model): two channel tiles, the second half empty
def test_conv_first_backward(frames, cout, h, w, fmt):
    """vpt_conv_first_backward (recompute -> arg-max search -> nine per-offset GEMMs with one-hot gradient fragments, no scatter) against autograd
    of conv -> ReLU -> max_pool2d on the same 16-bit-rounded weights and pooled values, both operand formats."""
    dt = torch.bfloat16 if fmt == "bf16" else torch.float16
    g = torch.Generator().manual_seed(16)
    W = (torch.randn(cout, 3, 3, 3, generator=g) * 0.3).requires_grad_(True)
    b = (0.1 * torch.randn(cout, generator=g)).requires_grad_(True)
    img = torch.randint(0, 256, (frames, h, w, 3), generator=g, dtype=torch.uint8)
    dP = torch.randn(frames, cout, h // 2, w // 2, generator=g).to(dt).float()
    Wb = ((W.detach() / 255.0).to(dt).float() * 255.0).requires_grad_(True)  # the kernel rounds W / 255 to 16 bits; compare like for like
    y = torch.relu(torch.nn.functional.conv2d(img.permute(0, 3, 1, 2).float() / 255.0, Wb, b, padding=1))
    y = y + (y.detach().to(dt).float() - y.detach())  # the kernel pools 16-bit-rounded values (straight-through here)
    pooled = torch.nn.functional.max_pool2d(y, 3, 2, 1)
    gW, gb = torch.autograd.grad((pooled * dP).sum(), [Wb, b])
    dW, db = ops.conv_first_backward(img.to(DEV), packing.pack_conv_first(W.detach().to(DEV), b.detach().to(DEV), dtype=dt),
                                     packing.nchw_to_blocked(dP, dtype=dt).to(DEV), cout)
    dW = ops.conv_first_grad_to_reference(dW)
    torch.cuda.synchronize()
    eW, eb = _l2(dW.cpu(), gW), _l2(db.cpu(), gb)
    print(f"PARITY[{fmt}] conv_first backward {frames}x{cout}x{h}x{w}: dW {eW:.3e} db {eb:.3e}")
    assert eW < 5e-3 and eb < 5e-3      # measured 4e-7 ... 4e-4 (every gradient enters the fp32 accumulation un-merged; round 3: 2e-3 through a bf16 scatter)


@pytest.mark.parametrize("fmt", ["bf16", "fp16"])
@pytest.mark.parametrize("frames,cout", [(3, 128), (2, 64), (9, 128)])
def test_conv_first_backward_with_the_n_backward_folded_in(frames, cout, fmt):
    """Round 6: stack 0's GroupNorm `n` backward inside the first conv's backward kernel (vpt_conv_first_backward_nfold) against the two-pass chain it
    replaces -- vpt_frame_affine_backward (reduce + apply: d(pooled) written in 16 bits) -> vpt_conv_first_backward -- on the SAME pooled tensor, statistics
    and incoming gradient.  Same arithmetic, same rounding point; the pooled value comes from the kernel's own arg-max search instead of a load.  A fused
    multiply-add contracted differently may move a d(pooled) value by one 16-bit ulp, hence a bound instead of torch.equal."""
    dt = torch.bfloat16 if fmt == "bf16" else torch.float16
    g = torch.Generator().manual_seed(61)
    h = w = 128
    W = torch.randn(cout, 3, 3, 3, generator=g) * 0.3
    b = 0.1 * torch.randn(cout, generator=g)
    img = torch.randint(0, 256, (frames, h, w, 3), generator=g, dtype=torch.uint8).to(DEV)
    wfrag = packing.pack_conv_first(W.to(DEV), b.to(DEV), dtype=dt)
    s_pool = torch.zeros(frames, 2, dtype=torch.float64, device=DEV)
    pooled = ops.conv_first(img, wfrag, cout, stats_out=s_pool)
    ng = (1 + 0.3 * torch.randn(cout, generator=g)).to(DEV)
    G = packing.nchw_to_blocked(torch.randn(frames, cout, h // 2, w // 2, generator=g) * (1e-2 if fmt == "fp16" else 1.0), dtype=dt).to(DEV)
    dg1, db1, dg2, db2 = (torch.zeros(cout, device=DEV) for _ in range(4))
    dp_ref = ops.frame_affine_backward(pooled, G, ng, s_pool, dg1, db1)
    dW1, dB1 = ops.conv_first_backward(img, wfrag, dp_ref, cout)
    ab = ops.frame_affine_backward_reduce(pooled, G, ng, s_pool, dg2, db2)
    dW2, dB2 = ops.conv_first_backward(img, wfrag, G, cout, nfold=(ng, s_pool, ab))
    torch.cuda.synchronize()
    assert torch.equal(dg1, dg2) and torch.equal(db1, db2)            # the same pass 1, fixed summation order
    eW, eb = _l2(dW2.cpu(), dW1.cpu()), _l2(dB2.cpu(), dB1.cpu())
    print(f"PARITY[{fmt}] first-conv backward with the n backward folded in, {frames}x{cout}: dW {eW:.2e} db {eb:.2e} vs the two-pass chain")
    assert eW < 2e-3 and eb < 2e-3
    # run-to-run: the folded kernel is as reproducible as the plain one
    dW3, dB3 = ops.conv_first_backward(img, wfrag, G, cout, nfold=(ng, s_pool, ab))
    assert torch.equal(dW2, dW3) and torch.equal(dB2, dB3)


def test_conv_prepare_fused_pool_backward():
    """prepare(dy=None, dpooled, argmax) vs prepare(dy = maxpool_backward(...)): same routing (all-zero windows differ only
    where the ReLU gate is closed); the fused route skips one bf16 rounding where a pixel wins several windows."""
    g = torch.Generator().manual_seed(17)
    f, cin, cout, h = 3, 64, 64, 32
    W = torch.randn(cout, cin, 3, 3, generator=g) * (1.6 / (cin * 9) ** 0.5)
    gain, bias = 1 + 0.2 * torch.randn(cin, generator=g), 0.1 * torch.randn(cin, generator=g)
    x = torch.relu(torch.randn(f, cin, h, h, generator=g)).to(torch.bfloat16).float()
    wpk, sa, sg = packing.pack_conv3x3(W.to(DEV), gain.to(DEV), bias.to(DEV))
    xb, st_in = packing.nchw_to_blocked(x).to(DEV), _stats_of(x).to(DEV)
    pre = ops.conv3x3(xb, wpk, sa, sg, st_in, cout)
    pooled, am = ops.maxpool(pre, want_argmax=True)
    dp = packing.nchw_to_blocked(torch.randn(f, cout, h // 2, h // 2, generator=g)).to(DEV)
    dpre = ops.maxpool_backward(pre, pooled, dp)
    ref = ops.conv_backward_prepare(dpre, pre, None, st_in, sa, sg, cin, want_t12=True)
    got = ops.conv_backward_prepare(None, pre, None, st_in, sa, sg, cin, dpooled=dp, argmax=am, want_t12=True)
    torch.cuda.synchronize()
    assert ((ref[0] != 0) == (got[0] != 0)).all()
    assert _l2(got[0].float().cpu(), ref[0].float().cpu()) < 5e-3      # two bf16 roundings vs one
    for r, o in zip(ref[1:], got[1:]):
        assert _l2(o.double().cpu(), r.double().cpu()) < 2e-3
    # exact reference: torch's max-pool backward on the same pre-pool values, gate, times rstd -- one bf16 rounding away
    pre_n = packing.blocked_to_nchw(pre.cpu(), cout, h, h).requires_grad_(True)
    (gpre,) = torch.autograd.grad((torch.nn.functional.max_pool2d(pre_n, 3, 2, 1) * packing.blocked_to_nchw(dp.cpu(), cout, h // 2, h // 2)).sum(), [pre_n])
    n = cin * h * h
    mu = x.reshape(f, -1).double().mean(1)
    rstd = torch.rsqrt(x.reshape(f, -1).double().var(1, unbiased=False) + 1e-5).float()
    exact = gpre * (pre_n.detach() > 0) * rstd.view(f, 1, 1, 1)
    assert _l2(packing.blocked_to_nchw(got[0].cpu(), cout, h, h), exact) < 3e-3


@pytest.mark.parametrize("fmt", ["bf16", "fp16"])
@pytest.mark.parametrize("f,cin,cout,h,w", [(3, 64, 64, 32, 32), (2, 32, 128, 64, 64), (2, 64, 96, 16, 16), (1, 32, 32, 48, 32)])
def test_conv_prepare_pooled_equals_prepare_with_argmax(f, cin, cout, h, w, fmt):
    """Round 5: the backward of a pool-fused firstconv from (dpooled, pooled, arg-max masks) alone (vpt_conv_backward_prepare_pooled) against
    round 4's path on the same layer -- vpt_conv_backward_prepare with the pre-pool tensor and vpt_maxpool_forward's arg-max bytes: the same
    operand dacc (sums of at most four 16-bit gradients in fp32, one rounding), the same T1 / T2, coefficients and edge-table sums."""
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[fmt]
    g = torch.Generator().manual_seed(177)
    W = torch.randn(cout, cin, 3, 3, generator=g) * (1.6 / (cin * 9) ** 0.5)
    gain, bias = 1 + 0.2 * torch.randn(cin, generator=g), 0.1 * torch.randn(cin, generator=g) - 0.4
    x = torch.relu(torch.randn(f, cin, h, w, generator=g)).to(dt).float()
    wpk, sa, sg = packing.pack_conv3x3(W.to(DEV), gain.to(DEV), bias.to(DEV), dtype=dt)
    xb, st_in = packing.nchw_to_blocked(x, dtype=dt).to(DEV), _stats_of(x).to(DEV)
    pre = ops.conv3x3(xb, wpk, sa, sg, st_in, cout)
    pooled, am = ops.maxpool(pre, want_argmax=True)
    pooled2, mask = ops.conv3x3_pool_argmax(xb, wpk, sa, sg, st_in, cout)
    assert torch.equal(pooled.view(torch.int16), pooled2.view(torch.int16))
    dp = packing.nchw_to_blocked(torch.randn(f, cout, h // 2, w // 2, generator=g) * (1e-2 if fmt == "fp16" else 1.0), dtype=dt).to(DEV)
    ref = ops.conv_backward_prepare(None, pre, None, st_in, sa, sg, cin, dpooled=dp, argmax=am, want_t12=True)
    got = ops.conv_backward_prepare_pooled(dp, pooled2, mask, st_in, sa, sg, cin, want_t12=True)
    torch.cuda.synchronize()
    a, r = got[0].float(), ref[0].float()
    assert a.shape == r.shape == pre.shape
    same = (got[0].view(torch.int16) == ref[0].view(torch.int16)) | ((a == 0) & (r == 0))
    print(f"PARITY pooled prepare vs prepare + arg-max [{fmt}] {f}x{cin}->{cout} {h}x{w}: dacc identical at {float(same.float().mean()):.6f}, rel-L2 {_l2(a.cpu(), r.cpu()):.2e}; "
          + " ".join(f"{n} {_l2(o.double().cpu(), q.double().cpu()):.2e}" for n, o, q in zip(("coef", "dsa", "dsg", "t12"), got[1:], ref[1:])))
    assert float(same.float().mean()) > 0.9999 and _l2(a.cpu(), r.cpu()) < 1e-4
    for o, q in zip(got[1:], ref[1:]):
        assert _l2(o.double().cpu(), q.double().cpu()) < 1e-4
    # ... and with the stack's GroupNorm `n` backward applied on the fly (nfold): the incoming gradient is G = d loss / d n(pooled); the two-pass
    # vpt_frame_affine_backward -> prepare_pooled chain is the reference (same arithmetic, same 16-bit rounding point of d(pooled))
    ng = (1 + 0.3 * torch.randn(cout, generator=g)).to(DEV)
    s_pool = _stats_of(packing.blocked_to_nchw(pooled2.cpu(), cout, h // 2, w // 2)).to(DEV)
    G = packing.nchw_to_blocked(torch.randn(f, cout, h // 2, w // 2, generator=g) * (1e-2 if fmt == "fp16" else 1.0), dtype=dt).to(DEV)
    dg1, db1, dg2, db2 = (torch.zeros(cout, device=DEV) for _ in range(4))
    dp_ref = ops.frame_affine_backward(pooled2, G, ng, s_pool, dg1, db1)
    ref2 = ops.conv_backward_prepare_pooled(dp_ref, pooled2, mask, st_in, sa, sg, cin, want_t12=True)
    ab = ops.frame_affine_backward_reduce(pooled2, G, ng, s_pool, dg2, db2)
    got2 = ops.conv_backward_prepare_pooled(G, pooled2, mask, st_in, sa, sg, cin, want_t12=True, nfold=(ng, s_pool, ab))
    torch.cuda.synchronize()
    assert torch.allclose(dg1, dg2, rtol=1e-5, atol=1e-6) and torch.allclose(db1, db2, rtol=1e-5, atol=1e-6)      # (the same pass 1; fp32 atomics in arrival order)
    a2, r2 = got2[0].float(), ref2[0].float()
    same2 = (got2[0].view(torch.int16) == ref2[0].view(torch.int16)) | ((a2 == 0) & (r2 == 0))
    print(f"PARITY pooled prepare with the n backward folded in [{fmt}]: dacc identical at {float(same2.float().mean()):.6f}, rel-L2 {_l2(a2.cpu(), r2.cpu()):.2e}; "
          + " ".join(f"{n} {_l2(o.double().cpu(), q.double().cpu()):.2e}" for n, o, q in zip(("coef", "dsa", "dsg", "t12"), got2[1:], ref2[1:])))
    eps = 2.0 ** -8 if fmt == "bf16" else 2.0 ** -11
    assert float(same2.float().mean()) > 0.995 and _l2(a2.cpu(), r2.cpu()) < eps      # (a fused multiply-add contracted differently may move a value by one 16-bit ulp)
    for o, q in zip(got2[1:], ref2[1:]):
        assert _l2(o.double().cpu(), q.double().cpu()) < eps


@pytest.mark.parametrize("fmt", ["bf16", "fp16"])
@pytest.mark.parametrize("f,cin,cout,h,w", [(3, 32, 64, 8, 8), (2, 64, 32, 16, 16), (2, 32, 32, 32, 32)])
def test_conv_prepare_kernels_bitwise_reproducible(f, cin, cout, h, w, fmt):
    """Every output of the three `prepare` kernels is a function of the inputs alone: 12 launches, torch.equal.  8 x 8 / 16 x 16 images are the shapes
    where TWO lanes of one wave share an edge-class entry of the ordered LDS reduction (add_edge_sums_ordered, J > 1 -- W = 8 in the per-pixel
    kernels, pooled width 8 in the pooled one): a lost addend or an order that depends on wave arrival shows here."""
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[fmt]
    g = torch.Generator().manual_seed(71)
    W = torch.randn(cout, cin, 3, 3, generator=g) * (1.6 / (cin * 9) ** 0.5)
    gain, bias = 1 + 0.2 * torch.randn(cin, generator=g), 0.1 * torch.randn(cin, generator=g)
    x = torch.relu(torch.randn(f, cin, h, w, generator=g)).to(dt).float()
    wpk, sa, sg = packing.pack_conv3x3(W.to(DEV), gain.to(DEV), bias.to(DEV), dtype=dt)
    xb, st_in = packing.nchw_to_blocked(x, dtype=dt).to(DEV), _stats_of(x).to(DEV)
    # (the layer output only gates and centres the sums here: any non-negative tensor serves, and the conv kernel needs multiples of 16 pixels)
    y = packing.nchw_to_blocked(torch.relu(torch.randn(f, cout, h, w, generator=g)), dtype=dt).to(DEV)
    sc = 1e-2 if fmt == "fp16" else 1.0
    dy = packing.nchw_to_blocked(torch.randn(f, cout, h, w, generator=g) * sc, dtype=dt).to(DEV)
    res = packing.nchw_to_blocked(torch.randn(f, cout, h, w, generator=g), dtype=dt).to(DEV)

    def runs(fn):
        outs = []
        for _ in range(12):
            d_sa, d_sg = torch.zeros_like(sa), torch.zeros_like(sg)
            o = fn(d_sa, d_sg)
            outs.append([t.clone() for t in o if t is not None] + [d_sa, d_sg])
        torch.cuda.synchronize()
        for o in outs[1:]:
            for a, b in zip(o, outs[0]):
                assert torch.equal(a.view(torch.int16) if a.dtype == dt else a, b.view(torch.int16) if b.dtype == dt else b)

    runs(lambda d_sa, d_sg: ops.conv_backward_prepare(dy, y, None, st_in, sa, sg, cin, d_sa=d_sa, d_sg=d_sg, want_t12=True))
    runs(lambda d_sa, d_sg: ops.conv_backward_prepare(dy, y, res, st_in, sa, sg, cin, d_sa=d_sa, d_sg=d_sg, want_t12=True))
    if w >= 16:     # the pool-fused pair (pre-pool width >= 16): pooled width 8 at w = 16
        pooled, mask = ops.conv3x3_pool_argmax(xb, wpk, sa, sg, st_in, cout)
        dp = packing.nchw_to_blocked(torch.randn(f, cout, h // 2, w // 2, generator=g) * sc, dtype=dt).to(DEV)
        runs(lambda d_sa, d_sg: ops.conv_backward_prepare_pooled(dp, pooled, mask, st_in, sa, sg, cin, d_sa=d_sa, d_sg=d_sg, want_t12=True))


def test_bc_gradients_bitwise_reproducible(trainer_1x):
    """behavioural_cloning.py:117-122: loss.backward() on one device returns the same bits for the same batch, and so does this backward --
    every cross-workgroup sum goes through a partial slab added in a fixed order (vpt_reduce.hip), every in-workgroup sum through an ordered
    LDS reduction, the frame scalars through fp64 sums of fp32 partials (exact).  The same batch 20 times in one process: torch.equal on every
    gradient tensor and on the loss.  B = 2, T = 70: three 32-query tiles per head (the dK / dV slab slots and the db_nd rows are exercised);
    140 frames in CNN chunks of 48 alternate over three streams with per-stream accumulators merged in stream order."""
    pol, cfg, sd = trainer_1x
    tr = BCTrainer(pol, train_cnn=True)
    b, t = 2, 70
    g = torch.Generator().manual_seed(77)
    img = torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8).to(DEV)
    first = torch.zeros(b, t, dtype=torch.bool, device=DEV)
    first[1, 0] = True
    ab, ac = torch.randint(0, 8641, (b, t), generator=g).to(DEV), torch.randint(0, 121, (b, t), generator=g).to(DEV)
    eng = pol._engine
    saved = eng.cnn_chunk, tr.cnn_streams
    try:
        eng.cnn_chunk, tr.cnn_streams = 48, 3
        runs = []
        for _ in range(20):
            loss, grads, _ = tr.loss_and_grads(img, first, pol.initial_state(b), ab, ac)
            runs.append((loss.clone(), {k: v.clone() for k, v in grads.items()}))
        torch.cuda.synchronize()
    finally:
        eng.cnn_chunk, tr.cnn_streams = saved
    l0, g0 = runs[0]
    assert len(g0) >= 120 and all(bool(torch.isfinite(v).all()) for v in g0.values())
    differing = set()
    for loss, grads in runs[1:]:
        assert torch.equal(loss, l0)
        differing |= {k for k in g0 if not torch.equal(grads[k], g0[k])}
    assert not differing, sorted(differing)[:8]


def test_trainer_checkpoint_resume(trainer_1x, tmp_path):
    """Policy weights (.weights format) + BCTrainer.state_dict() restore a run: the resumed step equals the uninterrupted
    one up to the order of the fp32 atomics inside the backward (same inputs, same Adam moments and step count)."""
    import copy
    pol, cfg, sd = trainer_1x
    g = torch.Generator().manual_seed(41)
    b, t = 2, 3
    img = torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8).to(DEV)
    first = torch.zeros(b, t, dtype=torch.bool, device=DEV)
    ab, ac = torch.randint(0, 8641, (b, t), generator=g).to(DEV), torch.randint(0, 121, (b, t), generator=g).to(DEV)
    start = copy.deepcopy(pol.state_dict())
    try:
        tr = BCTrainer(pol, train_cnn=True)
        st = pol.initial_state(b)
        for _ in range(2):
            _, st = tr.step(img, first, st, ab, ac)
        torch.save(pol.state_dict(), tmp_path / "w.weights")
        torch.save(tr.state_dict(), tmp_path / "opt.pt")
        st_saved = [(m.clone(), (k.clone(), v.clone())) for m, (k, v) in st]
        loss_a, _ = tr.step(img, first, st, ab, ac)
        after_a = {k: v.detach().clone() for k, v in pol.named_parameters()}
        # resume in a fresh trainer
        pol.load_state_dict(torch.load(tmp_path / "w.weights"), strict=False)
        tr2 = BCTrainer(pol, train_cnn=True, lr=1.0)          # hyper-parameters come from the checkpoint
        tr2.load_state_dict(torch.load(tmp_path / "opt.pt"))
        assert tr2.step_count == 2 and tr2.lr == tr.lr
        loss_b, _ = tr2.step(img, first, st_saved, ab, ac)
        torch.cuda.synchronize()
        assert abs(loss_a - loss_b) < 1e-5
        for k, v in pol.named_parameters():
            assert torch.allclose(v.detach(), after_a[k], rtol=0, atol=2e-6), (k, float((v.detach() - after_a[k]).abs().max()))
        moved = max(float((after_a[k] - start[k].to(DEV)).abs().max()) for k in after_a)
        assert moved > 1e-4                                     # (the three steps did change the weights: lr 1.81e-4 each)
    finally:
        pol.load_state_dict(start, strict=False)


def test_bc_gradients_independent_of_cnn_chunking(trainer_1x):
    """The CNN forward / backward runs in frame chunks (1024 by default) whose weight-gradient pieces are accumulated
    in place: ragged chunks must give the same gradients as one chunk (frames are independent, the statistics are
    fp64 across tiles, the split-K dense layer sums its slices in a fixed order).  Chunks of 10 frames keep the dense
    layer on the GEMM kernel in both runs (M <= 8 rows would take the GEMV kernel, whose fp32 summation order over
    K = 65536 differs in the last bits)."""
    pol, cfg, sd = trainer_1x
    tr = BCTrainer(pol, train_cnn=True)
    b, t = 3, 10                                    # 30 frames: one chunk vs chunks of 10, 10, 10
    g = torch.Generator().manual_seed(51)
    img = torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8).to(DEV)
    first = torch.zeros(b, t, dtype=torch.bool, device=DEV)
    ab, ac = torch.randint(0, 8641, (b, t), generator=g).to(DEV), torch.randint(0, 121, (b, t), generator=g).to(DEV)
    eng = pol._engine
    saved = eng.cnn_chunk
    try:
        eng.cnn_chunk = 1024
        l1, g1, _ = tr.loss_and_grads(img, first, pol.initial_state(b), ab, ac)
        g1 = {k: v.clone() for k, v in g1.items()}
        eng.cnn_chunk = 10                      # 10, 10, 10
        l2, g2, _ = tr.loss_and_grads(img, first, pol.initial_state(b), ab, ac)
        (pd1, _, _), _ = pol({"img": img}, first, pol.initial_state(b))          # inference path, ragged chunks too
        eng.cnn_chunk = 1024
        (pd2, _, _), _ = pol({"img": img}, first, pol.initial_state(b))
        torch.cuda.synchronize()
    finally:
        eng.cnn_chunk = saved
    assert abs(float(l1) - float(l2)) < 1e-5
    assert torch.equal(pd1["buttons"], pd2["buttons"])
    worst = 0.0
    for k, v in g1.items():
        if float(v.norm()) == 0:
            continue
        e = _l2(g2[k].float().cpu(), v.float().cpu())
        worst = max(worst, e)
        # Chunked == unchunked up to the association of the per-chunk partial sums (each chunk stream accumulates into its own buffers, merged in stream
        # order): measured 3.0e-7 in both formats on the final tree.  Until round 5 the bounds here had grown to 2e-2: fp32 atomics in arrival order made the
        # cancelling stack-0 sums scatter (up to 5e-4 in fp16), LDS float atomics in the `prepare` kernels added discrete alternatives, and the LayerNorm
        # backward's lost addend beside another process was still unexplained.  All three are gone (ordered reductions everywhere: DESIGN.md sections 5, 8b), so
        # the bound is back at the order of fp32 additions: what this test exists to catch -- a chunk's contribution dropped, doubled, or normalised with a
        # chunk-dependent statistic -- is O(0.1 ... 1).
        assert e < 1e-5, (k, e)
    print(f"PARITY BC gradients, 3 CNN chunks vs 1: worst rel-L2 {worst:.2e}")


def test_heads_logprob_backward_matches_autograd():
    """vpt_heads_logprob_backward: d/dz of log_softmax(z / T) for ARBITRARY incoming gradients (the autograd boundary)."""
    g = torch.Generator().manual_seed(11)
    m, nb, nc, temp = 5, 8641, 121, 2.0
    zb = (torch.randn(m, nb, generator=g) * 2).requires_grad_(True)
    zc = (torch.randn(m, nc, generator=g) * 2).requires_grad_(True)
    lb, lc = torch.log_softmax(zb / temp, -1), torch.log_softmax(zc / temp, -1)
    gb_in, gc_in, gv_in = torch.randn(m, nb, generator=g), torch.randn(m, nc, generator=g), torch.randn(m, generator=g)
    gb, gc = torch.autograd.grad((lb * gb_in).sum() + (lc * gc_in).sum(), [zb, zc])
    dz = ops.heads_logprob_backward(lb.detach().to(DEV), lc.detach().to(DEV), gb_in.to(DEV), gc_in.to(DEV), gv_in.to(DEV), 8768, temp)
    torch.cuda.synchronize()
    dz = dz.cpu().float()
    assert _l2(dz[:, :nb], gb) < 6e-3 and _l2(dz[:, nb:nb + nc], gc) < 6e-3
    assert _l2(dz[:, nb + nc], gv_in) < 6e-3 and float(dz[:, nb + nc + 1:].abs().max()) == 0.0
    dz2 = ops.heads_logprob_backward(lb.detach().to(DEV), lc.detach().to(DEV), None, gc_in.to(DEV), None, 8768, temp).cpu().float()
    assert float(dz2[:, :nb].abs().max()) == 0.0 and _l2(dz2[:, nb:nb + nc], gc) < 6e-3


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_reference_bc_loop_runs_unchanged(mode):
    """The statements of behavioural_cloning.py:57-67,99-122, literally, on the HIP policy: get_output_for_observation ->
    get_logprob_of_action -> detach the state -> (-log_prob / BATCH_SIZE).backward() x 8 -> (no-op) clip -> th.optim.Adam.step().
    The accumulated param.grad must equal BCTrainer's hand-driven gradients of the same 8 frames (same kernels) and point
    where the fp32 oracle's autograd points; Adam must move the weights."""
    th = torch
    from vpt_amd.lib.tree_util import tree_map
    pk = O.policy_kwargs_for("1x")
    cfg = O.config_from_policy_kwargs(pk, dict(temperature=2.0))
    sd = O.synthetic_state_dict(cfg, seed=0)
    policy = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0), precision=mode)
    policy.load_state_dict(sd, strict=False)
    policy = policy.to(DEV)
    BATCH_SIZE, LEARNING_RATE, WEIGHT_DECAY, MAX_GRAD_NORM = 8, 0.000181, 0.039428, 5.0
    g = torch.Generator().manual_seed(77)
    images = torch.randint(0, 256, (BATCH_SIZE, 128, 128, 3), generator=g, dtype=torch.uint8)
    a_b, a_c = torch.randint(0, 8641, (BATCH_SIZE,), generator=g), torch.randint(0, 121, (BATCH_SIZE,), generator=g)

    # ---- behavioural_cloning.py:57-67 ----
    trainable_parameters = policy.parameters()
    optimizer = th.optim.Adam(trainable_parameters, lr=LEARNING_RATE, weight_decay=WEIGHT_DECAY)
    dummy_first = th.from_numpy(np.array((False,))).to(DEV)
    episode_hidden_states = {}
    before = {n: p.detach().clone() for n, p in policy.named_parameters()}
    # ---- behavioural_cloning.py:86-122 (one batch; every sample from episode 0) ----
    batch_loss = 0
    for i in range(BATCH_SIZE):
        episode_id = 0
        agent_action = {"buttons": a_b[i].reshape(1, 1).to(DEV), "camera": a_c[i].reshape(1, 1).to(DEV)}   # agent._env_action_to_agent
        agent_obs = {"img": images[i:i + 1].to(DEV)}                                                       # agent._env_obs_to_agent
        if episode_id not in episode_hidden_states:
            episode_hidden_states[episode_id] = policy.initial_state(1)
        agent_state = episode_hidden_states[episode_id]
        pi_distribution, v_prediction, new_agent_state = policy.get_output_for_observation(agent_obs, agent_state, dummy_first)
        log_prob = policy.get_logprob_of_action(pi_distribution, agent_action)
        new_agent_state = tree_map(lambda x: x.detach(), new_agent_state)
        episode_hidden_states[episode_id] = new_agent_state
        loss = -log_prob / BATCH_SIZE
        batch_loss += loss.item()
        loss.backward()
    th.nn.utils.clip_grad_norm_(trainable_parameters, MAX_GRAD_NORM)       # exhausted generator: a no-op, as in the reference
    grads_loop = {n: p.grad.detach().clone() for n, p in policy.named_parameters() if p.grad is not None}
    optimizer.step()
    optimizer.zero_grad()
    torch.cuda.synchronize()

    assert "value_head.linear.weight" not in grads_loop          # no gradient under the BC loss, as in the reference
    moved = sum(int(not torch.equal(before[n], p.detach())) for n, p in policy.named_parameters() if n in grads_loop)
    assert moved == len(grads_loop) and len(grads_loop) >= 129

    # (1) same kernels driven by hand: BCTrainer on the same 8 frames, T = 1 each, state carried
    pol2 = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0), precision=mode)
    pol2.load_state_dict(sd, strict=False)
    pol2 = pol2.to(DEV)
    tr = BCTrainer(pol2, train_cnn=True, optimizer_state=False)
    st, acc, loss_sum = pol2.initial_state(1), {}, 0.0
    so, acc_ref, loss_ref = O.initial_state(cfg, 1), {}, 0.0
    first = torch.zeros(1, 1, dtype=torch.bool)
    for i in range(BATCH_SIZE):
        l, gr, st = tr.loss_and_grads(images[i:i + 1, None].to(DEV), first.to(DEV), st, a_b[i].reshape(1, 1).to(DEV), a_c[i].reshape(1, 1).to(DEV))
        loss_sum += float(l) / BATCH_SIZE
        for n, v in gr.items():
            acc[n] = acc.get(n, 0) + v.reshape(before[n].shape) / BATCH_SIZE
        # (2) the fp32 oracle's autograd (the reference's gradients, tests/golden/make_golden_bc.py pins it)
        lr_, gr_, so = O.bc_loss_and_grads(sd, cfg, images[i:i + 1, None], first, so, a_b[i].reshape(1, 1), a_c[i].reshape(1, 1))
        loss_ref += lr_ / BATCH_SIZE
        for n, v in gr_.items():
            acc_ref[n] = acc_ref.get(n, 0) + v / BATCH_SIZE
    torch.cuda.synchronize()
    assert abs(batch_loss - loss_sum) < 1e-4 and abs(batch_loss - loss_ref) < 2e-2
    worst, worst_cos, cos_all = 0.0, 1.0, []
    for n, gl in grads_loop.items():
        if float(acc[n].norm()) == 0.0:
            continue
        e = _l2(gl, acc[n])
        worst = max(worst, e)
        # autograd boundary == hand-driven trainer: the same kernels on the same inputs in the same order, and since round 6 every reduction of the backward
        # has a fixed order -- measured 0.0 (bit-identical) in both formats.  The bound admits a different association of the eight per-frame gradients
        # only (param.grad accumulates them one by one, the trainer sums them in this test): 1e-5.  (Round 5 held this at 2e-2 to admit the defect that
        # round 6 root-caused: DESIGN.md section 8b.)
        assert e < 1e-5, (n, e)
        ref = acc_ref[n]
        if float(ref.norm()) > 0:
            cos = float((gl.cpu() * ref).sum() / (gl.cpu().norm() * ref.norm()))
            worst_cos = min(worst_cos, cos)
            cos_all.append(cos)
            assert cos > P.GRAD_BOUNDS[mode]["cos_min"], (n, cos)    # per-format bound (tests/parity.py: ReLU-gate flips of a 16-bit forward)
    assert sum(cos_all) / len(cos_all) > P.GRAD_BOUNDS[mode]["cos_mean"], sum(cos_all) / len(cos_all)
    print(f"PARITY[{mode}] reference BC loop: mean cosine {sum(cos_all) / len(cos_all):.4f}; param.grad vs BCTrainer worst rel-L2 {worst:.2e}; worst cosine vs fp32 oracle autograd {worst_cos:.3f}; "
          f"loss {batch_loss:.4f} (trainer {loss_sum:.4f}, oracle {loss_ref:.4f})")


def test_autograd_boundary_with_logit_mask():
    """obs["mask"] under autograd: a masked logit was overwritten by the constant LOG0 in the forward
    (lib/action_head.py:170-171), so no gradient reaches it -- the bias gradient of an always-masked action is exactly 0, the
    others match torch autograd through the oracle."""
    pk = O.policy_kwargs_for("1x")
    cfg = O.config_from_policy_kwargs(pk, dict(temperature=2.0))
    sd = O.synthetic_state_dict(cfg, seed=0)
    pol = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0))
    pol.load_state_dict(sd, strict=False)
    pol = pol.to(DEV)
    g = torch.Generator().manual_seed(91)
    b, t = 1, 3
    img = torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8)
    mb = torch.rand(b, t, 1, 8641, generator=g) > 0.3
    mb[..., :50] = False            # actions 0..49 never available
    mb[..., 100] = True
    first = torch.zeros(b, t, dtype=torch.bool)
    tgt = torch.full((b, t, 1, 1), 100, dtype=torch.int64)
    (pd, _, _), _ = pol({"img": img.to(DEV), "mask": {"buttons": mb.to(DEV)}}, first.to(DEV), pol.initial_state(b))
    assert pd["buttons"].requires_grad
    loss = -(pd["buttons"].gather(-1, tgt.to(DEV)).mean() + pd["camera"][..., 7].mean())
    loss.backward()
    torch.cuda.synchronize()
    gb = pol.pi_head.buttons.linear_layer.bias.grad.cpu()
    assert float(gb[:50].abs().max()) == 0.0
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point}
    ref = O.policy_forward(leaves, cfg, img, first, O.initial_state(cfg, b), grad=True, mask={"buttons": mb})
    loss_ref = -(ref["buttons"].gather(-1, tgt).mean() + ref["camera"][..., 7].mean())
    gref, = torch.autograd.grad(loss_ref, [leaves["pi_head.buttons.linear_layer.bias"]])
    assert abs(float(loss) - float(loss_ref)) < 2e-2
    assert _l2(gb, gref) < 5e-2, _l2(gb, gref)


def test_fp16_loss_scale_overflow_skips_the_step_on_the_device():
    """precision="fp16": a loss scale far beyond IEEE half's range makes the 16-bit gradient buffers overflow; vpt_grads_nonfinite_multi
    must flag it, the Adam launch must leave EVERY tensor (weights and moments) untouched, the step count must not advance and the scale
    must halve -- then, at a sane scale, the same trainer trains.  (What torch.cuda.amp.GradScaler does for an fp16 run of
    behavioural_cloning.py:117-122; the skip decision never leaves the device.)"""
    pk = O.policy_kwargs_for("1x")
    cfg = O.config_from_policy_kwargs(pk, dict(temperature=2.0))
    sd = O.synthetic_state_dict(cfg, seed=0)
    pol = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0), precision="fp16")
    pol.load_state_dict(sd, strict=False)
    pol = pol.to(DEV)
    g = torch.Generator().manual_seed(61)
    b, t = 2, 3
    img = torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8).to(DEV)
    first = torch.zeros(b, t, dtype=torch.bool, device=DEV)
    ab, ac = torch.randint(0, 8641, (b, t), generator=g).to(DEV), torch.randint(0, 121, (b, t), generator=g).to(DEV)
    tr = BCTrainer(pol, lr=1e-4, weight_decay=0.0, loss_scale=2.0 ** 30)
    before = {k: v.detach().clone() for k, v in pol.named_parameters()}
    loss, _ = tr.step(img, first, pol.initial_state(b), ab, ac)
    torch.cuda.synchronize()
    assert tr.skipped_steps == 1 and tr.step_count == 0 and tr.loss_scale == 2.0 ** 29
    assert abs(loss - 13.8) < 0.5                                                   # the forward (and the reported loss) are unaffected
    assert all(torch.equal(v.detach(), before[k]) for k, v in pol.named_parameters())
    assert all(float(m.abs().max()) == 0.0 for m in tr.m.values()) and all(float(v.abs().max()) == 0.0 for v in tr.v.values())
    tr.loss_scale = 256.0
    losses = [tr.step(img, first, pol.initial_state(b), ab, ac)[0] for _ in range(4)]
    torch.cuda.synchronize()
    assert tr.skipped_steps == 1 and tr.step_count == 4 and losses[-1] < losses[0] - 0.1
    moved = sum(int(not torch.equal(v.detach(), before[k])) for k, v in pol.named_parameters() if k in tr.m)
    assert moved == len(tr.m)


def test_fp16_autograd_boundary_overflow_returns_no_gradients():
    """The reference's own loop (loss.backward() + th.optim.Adam, behavioural_cloning.py:117-122) over precision="fp16": a half overflow
    inside the 16-bit gradient buffers must never reach param.grad.  With the lift forced far beyond IEEE half's range the backward
    warns, leaves every .grad None (an optimizer then skips the parameters) and halves the lift; at the normal lift the same call
    yields finite gradients."""
    pk = O.policy_kwargs_for("1x")
    cfg = O.config_from_policy_kwargs(pk, dict(temperature=2.0))
    sd = O.synthetic_state_dict(cfg, seed=0)
    pol = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0), precision="fp16")
    pol.load_state_dict(sd, strict=False)
    pol = pol.to(DEV)
    g = torch.Generator().manual_seed(62)
    img = torch.randint(0, 256, (2, 128, 128, 3), generator=g, dtype=torch.uint8).to(DEV)
    first = torch.zeros(2, dtype=torch.bool, device=DEV)
    tgt = {"buttons": torch.randint(0, 8641, (2, 1), generator=g).to(DEV), "camera": torch.randint(0, 121, (2, 1), generator=g).to(DEV)}

    def backward_once():
        pol.zero_grad(set_to_none=True)
        pd, _, _ = pol.get_output_for_observation({"img": img}, pol.initial_state(2), first)
        loss = -pol.get_logprob_of_action(pd, tgt).mean()
        loss.backward()
        torch.cuda.synchronize()
        return float(loss)

    backward_once()                                          # creates the gradient engine
    eng = next(iter(pol._grad_engines.values()))
    assert eng.autograd_overflows == 0 and all(torch.isfinite(p.grad).all() for p in pol.parameters() if p.grad is not None)
    eng.autograd_lift = 2.0 ** 40
    with pytest.warns(RuntimeWarning, match="overflowed"):
        backward_once()
    assert eng.autograd_overflows == 1 and eng.autograd_lift == 2.0 ** 39
    assert all(p.grad is None for p in pol.parameters())
    eng.autograd_lift = 256.0
    backward_once()
    grads = [p.grad for p in pol.parameters() if p.grad is not None]
    assert len(grads) >= 129 and all(torch.isfinite(g_).all() for g_ in grads)

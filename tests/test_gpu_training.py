"""Backward kernels and the BC step (heads + trunk + transformer; CNN frozen) against torch autograd
through the CPU oracle.  Needs an MI355X.  Gradients pass through bf16 MFMA GEMMs (fp32 accumulate): bounds are
relative L2 per tensor (3e-2; typically 0.5-1.5e-2), fp32 kernels (LN / attention backward) 1e-3."""
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


@pytest.mark.parametrize("m,d,relu_in", [(70, 2048, False), (33, 256, True), (5, 1024, True)])
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


@pytest.mark.parametrize("bsz,t,first_flags", [(2, 70, [False, True]), (1, 128, [False]), (2, 5, [False, False])])
def test_attention_backward(bsz, t, first_flags):
    g = torch.Generator().manual_seed(4)
    heads, maxlen = 2, 128
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


@pytest.fixture(scope="module")
def trainer_1x():
    pk = O.policy_kwargs_for("1x")
    cfg = O.config_from_policy_kwargs(pk, dict(temperature=2.0))
    sd = O.synthetic_state_dict(cfg, seed=0)
    pol = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0))
    pol.load_state_dict(sd, strict=False)
    return pol.to(DEV), cfg, sd


def test_bc_gradients_vs_oracle(trainer_1x):
    """Every trainable tensor's gradient against the fp32 oracle pinned to the reference (cosine >= 0.93, norm within
    25 %) and against autograd through the bf16-emulating oracle.  A bf16 forward flips ~1-2 % of the ReLU gates,
    which alone moves gradients 15-30 % in relative L2 even with exact autograd (reproduced on the CPU emulation,
    oracle/vpt_oracle_bf16.py); the per-kernel tests above check the backward math itself at 1e-3 on identical inputs."""
    from oracle import vpt_oracle_bf16 as OB
    pol, cfg, sd = trainer_1x
    tr = BCTrainer(pol)
    b, t = 2, 6
    g = torch.Generator().manual_seed(5)
    img = torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8)
    first = torch.zeros(b, t, dtype=torch.bool)
    ab = torch.randint(0, 8641, (b, t), generator=g)
    ac = torch.randint(0, 121, (b, t), generator=g)
    torch.set_num_threads(max(1, min(32, len(__import__("os").sched_getaffinity(0)))))
    loss_ref, grads_ref, _ = O.bc_loss_and_grads(sd, cfg, img, first, O.initial_state(cfg, b), ab, ac)
    loss_em, grads_em = OB.bc_loss_and_grads(sd, cfg, img, first, O.initial_state(cfg, b), ab, ac)
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
    print("PARITY BC grads vs bf16-emulating oracle: worst rel-L2", sorted(l2_em.items(), key=lambda kv: -kv[1])[:4])
    print("PARITY BC grads vs fp32 oracle: worst cosine", sorted(cos_ref.items(), key=lambda kv: kv[1])[:4])
    assert len(l2_em) >= 60
    # the CPU emulation and the GPU round at the same points but sum in different orders, so ~1 % of the ReLU
    # gates still differ (tools/bc_grad_diag.py): the bound that holds is on direction and norm, not on L2
    bad = {k: v for k, v in l2_em.items() if v > 0.4}
    assert not bad, bad
    assert min(cos_ref.values()) > 0.93, min(cos_ref.values())
    for name, c in cos_ref.items():
        ref = grads_ref[name]
        ratio = float(grads[name].cpu().norm() / ref.norm())
        assert 0.8 < ratio < 1.25, (name, ratio)


def test_bc_step_reduces_loss(trainer_1x):
    pol, cfg, sd = trainer_1x
    tr = BCTrainer(pol, lr=3e-4, weight_decay=0.0)
    b, t = 2, 4
    g = torch.Generator().manual_seed(6)
    img = torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8).to(DEV)
    first = torch.zeros(b, t, dtype=torch.bool, device=DEV)
    ab = torch.randint(0, 8641, (b, t), generator=g).to(DEV)
    ac = torch.randint(0, 121, (b, t), generator=g).to(DEV)
    losses = []
    for _ in range(4):
        loss, _ = tr.step(img, first, pol.initial_state(b), ab, ac)
        losses.append(loss)
    torch.cuda.synchronize()
    print("BC losses on a fixed batch:", losses)
    assert losses[-1] < losses[0] - 0.05

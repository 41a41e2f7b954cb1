"""The other BASELINE.json configurations at reduced B x T (same architecture, widths and code paths as the full
sizes): 2x / 3x policy forward, the 4x inverse-dynamics model, and a 3x BC step -- against the fp32 oracle.
Tolerances as in tests/test_gpu_policy.py (bf16 MFMA operands: log-probs 3e-3 rel-L2, 1e-2 max/max)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import vpt_amd  # noqa: E402,F401
from vpt_amd.lib.policy import InverseActionPolicy, MinecraftAgentPolicy  # noqa: E402
from vpt_amd.lib.types import idm_action_space, minecraft_action_space  # noqa: E402
from vpt_amd.training import BCTrainer  # noqa: E402
from oracle import vpt_oracle as O  # noqa: E402

DEV = "cuda"


def _l2(a, ref):
    a, ref = a.double(), ref.double()
    return float((a - ref).norm() / ref.norm())


def _threads():
    import os
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))


@pytest.mark.parametrize("model,t", [("2x", 6), ("3x", 5)])
def test_policy_forward_wide_models(model, t):
    _threads()
    pk = O.policy_kwargs_for(model)
    cfg = O.config_from_policy_kwargs(pk, dict(temperature=2.0))
    sd = O.synthetic_state_dict(cfg, seed=0)
    pol = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0))
    missing, unexpected = pol.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    pol = pol.to(DEV)
    g = torch.Generator().manual_seed(21)
    b = 2
    img = torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8)
    first = torch.zeros(b, t, dtype=torch.bool)
    first[1, 0] = True
    ref = O.policy_forward(sd, cfg, img, first, O.initial_state(cfg, b))
    (pd, vpred, _), state = pol({"img": img.to(DEV)}, first.to(DEV), pol.initial_state(b))
    torch.cuda.synchronize()
    for head in ("buttons", "camera"):
        got, want = pd[head].cpu(), ref[head]
        e, m = _l2(got, want), float((got - want).abs().max() / want.abs().max())
        print(f"PARITY {model} forward {head}: rel-L2 {e:.3e} max/max {m:.3e}")
        assert e < 3e-3 and m < 1e-2
    k_ref = ref["state_out"][-1][1][0]
    assert _l2(state[-1][1][0].cpu(), k_ref) < 6e-2
    assert torch.equal(state[0][0].cpu(), ref["state_out"][0][0])


def test_idm_4x_forward():
    """BASELINE.json config 3's architecture (hid 4096, 32 heads, channels 256/512/512, 2 unmasked layers) on a 16-frame window."""
    _threads()
    kw = O.idm_kwargs_for("4x")
    cfg = O.idm_config_from_kwargs(kw, dict(temperature=2.0))
    sd = O.idm_synthetic_state_dict(cfg, seed=0)
    pol = InverseActionPolicy(idm_action_space(), pi_head_kwargs=dict(temperature=2.0), idm_net_kwargs=kw)
    missing, unexpected = pol.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    pol = pol.to(DEV)
    g = torch.Generator().manual_seed(22)
    t = 16
    img = torch.randint(0, 256, (1, t, 128, 128, 3), generator=g, dtype=torch.uint8)
    ref = O.idm_forward(sd, cfg, img)
    (pd, _, _), _ = pol({"img": img.to(DEV)}, first=None, state_in=pol.initial_state(1))
    torch.cuda.synchronize()
    for head in ("buttons", "camera"):
        got, want = pd[head].cpu(), ref[head]
        e, m = _l2(got, want), float((got - want).abs().max())
        print(f"PARITY 4x IDM {head}: rel-L2 {e:.3e} max|d| {m:.3e}")
        assert e < 1.5e-2 and m < 3e-2   # log-probs of 2- / 11-way softmaxes are O(1): absolute bound, as in test_gpu_idm.py


def test_bc_step_3x():
    """Config 5's model: one 3x BC step (all parameters) -- loss and the gradient direction of a few tensors vs the fp32 oracle."""
    _threads()
    pk = O.policy_kwargs_for("3x")
    cfg = O.config_from_policy_kwargs(pk, dict(temperature=2.0))
    sd = O.synthetic_state_dict(cfg, seed=0)
    pol = MinecraftAgentPolicy(minecraft_action_space(), pk, dict(temperature=2.0))
    pol.load_state_dict(sd, strict=False)
    pol = pol.to(DEV)
    b, t = 2, 4
    g = torch.Generator().manual_seed(23)
    img = torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8)
    first = torch.zeros(b, t, dtype=torch.bool)
    ab, ac = torch.randint(0, 8641, (b, t), generator=g), torch.randint(0, 121, (b, t), generator=g)
    loss_ref, grads_ref, _ = O.bc_loss_and_grads(sd, cfg, img, first, O.initial_state(cfg, b), ab, ac)
    tr = BCTrainer(pol, train_cnn=True)
    loss, grads, _ = tr.loss_and_grads(img.to(DEV), first.to(DEV), pol.initial_state(b), ab.to(DEV), ac.to(DEV))
    torch.cuda.synchronize()
    assert abs(float(loss) - loss_ref) < 2e-2
    worst = 1.0
    for name in tr.trainable:
        ref = grads_ref[name]
        if float(ref.norm()) == 0.0:
            continue
        mine = grads[name].cpu().reshape(ref.shape)
        assert torch.isfinite(mine).all(), name
        cos = float((mine * ref).sum() / (mine.norm() * ref.norm()))
        worst = min(worst, cos)
        assert cos > 0.7, (name, cos)     # bf16 gate flips: see test_bc_gradients_vs_oracle for the calibrated bound
    print(f"PARITY 3x BC gradients: worst cosine vs fp32 oracle {worst:.3f}")

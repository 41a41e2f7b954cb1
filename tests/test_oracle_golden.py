"""The oracle (oracle/vpt_oracle.py) against the golden vectors produced by the live reference
(tests/golden/make_golden.py).  CPU only.  Tolerance: fp32 CPU vs fp32 CPU, different op order ->
1e-4 absolute on log-probs (values O(10)), exact on masks and deterministic action indices."""
import os

import numpy as np
import torch

from oracle import vpt_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs(seed, b, t):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8)


def _setup():
    cfg = O.config_from_policy_kwargs(O.policy_kwargs_for("1x"), dict(temperature=2.0))
    sd = O.synthetic_state_dict(cfg, seed=0)
    return cfg, sd


def test_policy_chunks_match_reference(golden_1x):
    torch.set_num_threads(8)
    cfg, sd = _setup()
    G = golden_1x
    b = 2
    state = O.initial_state(cfg, b)
    for tag, t, first0 in [("A", 4, [False, True]), ("B", 3, [False, False]), ("C", 1, [False, False])]:
        img = _inputs(100 + ord(tag), b, t)
        first = torch.zeros(b, t, dtype=torch.bool)
        first[:, 0] = torch.tensor(first0)
        out = O.policy_forward(sd, cfg, img, first, state)
        state = out["state_out"]
        np.testing.assert_allclose(out["buttons"].numpy(), G[f"{tag}_buttons"], atol=1e-4, rtol=0)
        np.testing.assert_allclose(out["camera"].numpy(), G[f"{tag}_camera"], atol=1e-4, rtol=0)
        np.testing.assert_allclose(out["vpred"].numpy(), G[f"{tag}_vpred"], atol=1e-4, rtol=0)
        if tag == "A":
            np.testing.assert_allclose(out["latent"].numpy(), G["A_latent"], atol=2e-4, rtol=0)
        for l, (m, (k, v)) in enumerate(state):
            assert np.array_equal(m.numpy(), G[f"{tag}_mask{l}"])
            np.testing.assert_allclose(k[:, -4:].numpy(), G[f"{tag}_Ktail{l}"], atol=1e-4, rtol=1e-4)
            np.testing.assert_allclose(v[:, -4:].numpy(), G[f"{tag}_Vtail{l}"], atol=1e-4, rtol=1e-4)
            np.testing.assert_allclose(k.double().sum(dim=(1, 2)).numpy(), G[f"{tag}_Ksum{l}"], rtol=1e-4, atol=1e-2)
            np.testing.assert_allclose(v.double().sum(dim=(1, 2)).numpy(), G[f"{tag}_Vsum{l}"], rtol=1e-4, atol=1e-2)
        # logits are log-probabilities
        assert torch.allclose(out["buttons"].exp().sum(-1), torch.ones(b, t, 1), atol=1e-4)

    # act(): deterministic argmax indices must be bit-exact, log_prob / value close
    img = _inputs(999, b, 1)
    out = O.policy_forward(sd, cfg, img, torch.zeros(b, 1, dtype=torch.bool), state)
    ab = out["buttons"][:, 0].argmax(-1)
    ac = out["camera"][:, 0].argmax(-1)
    assert np.array_equal(ab.numpy(), G["act_buttons"])
    assert np.array_equal(ac.numpy(), G["act_camera"])
    lp = out["buttons"][:, 0].gather(-1, ab.unsqueeze(-1)).sum(-1)[:, 0] + out["camera"][:, 0].gather(-1, ac.unsqueeze(-1)).sum(-1)[:, 0]
    np.testing.assert_allclose(lp.numpy(), G["act_log_prob"], atol=1e-4)
    v = O.denormalize_value(sd, "value_head.", out["vpred"])[:, 0]
    np.testing.assert_allclose(v.numpy(), G["act_vpred"], atol=1e-4)


def test_cnn_taps_match_reference(golden_1x):
    cfg, sd = _setup()
    G = golden_1x
    img = _inputs(7, 1, 2)
    x = img.reshape(2, 128, 128, 3).float() / 255.0
    x = x.permute(0, 3, 1, 2)
    for s in range(3):
        x = O.cnn_stack(sd, f"net.img_process.cnn.stacks.{s}.", x)
        np.testing.assert_allclose(x.mean(dim=(1, 2, 3)).numpy(), G[f"cnn_stack{s}_mean"], atol=1e-5)
        np.testing.assert_allclose(x.std(dim=(1, 2, 3)).numpy(), G[f"cnn_stack{s}_std"], atol=1e-5)
        np.testing.assert_allclose(x[:, :8, :4, :4].numpy(), G[f"cnn_stack{s}_head"], atol=1e-4)


def test_chunked_equals_one_pass():
    """evaluating T=6 as 3+3 with KV carry equals one pass (banded mask + ring update)."""
    cfg, sd = _setup()
    b = 1
    img = _inputs(5, b, 6)
    first = torch.zeros(b, 6, dtype=torch.bool)
    one = O.policy_forward(sd, cfg, img, first, O.initial_state(cfg, b))
    st = O.initial_state(cfg, b)
    outs = []
    for c in range(2):
        o = O.policy_forward(sd, cfg, img[:, 3 * c:3 * c + 3], first[:, :3], st)
        st = o["state_out"]
        outs.append(o["buttons"])
    np.testing.assert_allclose(torch.cat(outs, 1).numpy(), one["buttons"].numpy(), atol=1e-4)


def test_idm_matches_reference():
    """Inverse dynamics model (lib/policy.py:342-467) on the `tiny` structure-preserving config."""
    import os
    G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "idm_tiny_seed0.npz")))
    kw = O.idm_kwargs_for("tiny")
    cfg = O.idm_config_from_kwargs(kw, dict(temperature=2.0))
    sd = O.idm_synthetic_state_dict(cfg, seed=0)
    g = torch.Generator().manual_seed(42)
    img = torch.randint(0, 256, (1, 12, 128, 128, 3), generator=g, dtype=torch.uint8)
    out = O.idm_forward(sd, cfg, img)
    np.testing.assert_allclose(out["buttons"].numpy(), G["buttons"], atol=1e-4)
    np.testing.assert_allclose(out["camera"].numpy(), G["camera"], atol=1e-4)
    assert np.array_equal(out["buttons"].argmax(-1).numpy(), G["ac_buttons"])
    assert np.array_equal(out["camera"].argmax(-1).numpy(), G["ac_camera"])
    lp = out["buttons"].max(-1).values.sum(-1) + out["camera"].max(-1).values.sum(-1)
    np.testing.assert_allclose(lp.numpy(), G["log_prob"], atol=1e-3)


def test_bc_gradients_match_reference():
    """Oracle autograd of the BC loss vs the live reference's loss.backward() (tests/golden/make_golden_bc.py)."""
    import os
    G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "bc_1x_seed0.npz")))
    cfg, sd = _setup()
    b, t = 2, 3
    state = O.initial_state(cfg, b)
    warm = O.policy_forward(sd, cfg, _inputs(300, b, 4), torch.zeros(b, 4, dtype=torch.bool), state)
    img = _inputs(301, b, t)
    first = torch.zeros(b, t, dtype=torch.bool)
    loss, grads, _ = O.bc_loss_and_grads(sd, cfg, img, first, warm["state_out"],
                                         torch.from_numpy(G["act_buttons"]), torch.from_numpy(G["act_camera"]))
    assert abs(loss - float(G["loss"])) < 1e-4
    checked = 0
    for name, gr in grads.items():
        ref_norm = float(G["norm/" + name])
        mine = float(gr.double().norm())
        assert abs(mine - ref_norm) <= 2e-3 * max(ref_norm, 1e-6) + 1e-7, (name, mine, ref_norm)
        np.testing.assert_allclose(gr.reshape(-1)[:16].numpy(), G["head/" + name], rtol=1e-2, atol=1e-7 + 2e-2 * ref_norm / max(gr.numel() ** 0.5, 1))
        checked += 1
    assert checked == 134
    # value head receives no gradient from the BC loss
    assert float(G["norm/value_head.linear.weight"]) == 0.0


def test_oracle_logit_mask_matches_live_reference_golden():
    """obs["mask"] (lib/policy.py:257-266 -> lib/action_head.py:170-171): masked logits become LOG0 = -100 before the softmax.
    Golden: tests/golden/make_golden_mask.py (unmodified reference)."""
    G = dict(np.load(os.path.join(ROOT, "tests", "golden", "mask_1x_seed0.npz")))
    g = torch.Generator().manual_seed(321)
    b, t = 2, 3
    img = torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8)
    gm = torch.Generator().manual_seed(5)
    mb = torch.rand(b, t, 1, 8641, generator=gm) > 0.3
    mc = torch.rand(b, t, 1, 121, generator=gm) > 0.5
    mb[..., 0] = True
    mc[..., 60] = True
    cfg = O.config_from_policy_kwargs(O.policy_kwargs_for("1x"), dict(temperature=2.0))
    sd = O.synthetic_state_dict(cfg, seed=0)
    out = O.policy_forward(sd, cfg, img, torch.zeros(b, t, dtype=torch.bool), O.initial_state(cfg, b), mask={"buttons": mb, "camera": mc})
    for k in ("buttons", "camera"):
        assert np.abs(out[k].numpy() - G[k]).max() < 2e-4
        assert np.array_equal(out[k].argmax(-1).numpy(), G["argmax_" + k])
        assert float(out[k][~(mb if k == "buttons" else mc)].max()) < -90.0      # masked entries sit at LOG0 - logsumexp

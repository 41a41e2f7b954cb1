"""CPU side of the drop-in proof (no GPU, no kernel launches): the UNMODIFIED reference wrappers construct over the HIP policy
classes, the `.model` / `.weights` formats round-trip between the two implementations with strict key / shape equality, and the
test families used by the GPU action tests have the properties their docstrings claim.  Needs the packaged reference
(oracle/_ref/vpt_reference.zip, made from /root/reference by __graft_entry__.build()); skipped without it."""
import pickle

import pytest
import torch

import vpt_amd  # noqa: F401
from vpt_amd.lib import policy as hip_policy
from oracle import vpt_oracle as O
from tests import parity as P
from tests import ref_env


@pytest.fixture(scope="module")
def R():
    ref = ref_env.reference()
    if ref is None:
        pytest.skip("oracle/_ref/vpt_reference.zip is absent")
    return ref


def test_reference_wrappers_construct_over_the_hip_policy_and_weights_round_trip(R, tmp_path):
    pk = O.policy_kwargs_for("1x")
    ph = dict(temperature=2.0)
    cfg = O.config_from_policy_kwargs(pk, ph)
    sd = O.synthetic_state_dict(cfg, seed=0)
    # `.model`: what run_agent.py:11-14 reads
    mp = tmp_path / "m.model"
    with open(mp, "wb") as f:
        pickle.dump(ref_env.model_file_dict(pk, dict(temperature="2.0")), f)
    params = pickle.load(open(mp, "rb"))
    policy_kwargs = params["model"]["args"]["net"]["args"]
    pi_head_kwargs = params["model"]["args"]["pi_head_opts"]
    pi_head_kwargs["temperature"] = float(pi_head_kwargs["temperature"])
    ag = R.agent
    env = ref_env.FakeEnv(ag)
    ref_cls = ag.MinecraftAgentPolicy
    ag.MinecraftAgentPolicy = hip_policy.MinecraftAgentPolicy
    try:
        hip_agent = ag.MineRLAgent(env, device="cpu", policy_kwargs=policy_kwargs, pi_head_kwargs=pi_head_kwargs)   # (cpu: construction only)
    finally:
        ag.MinecraftAgentPolicy = ref_cls
    ref_agent = ag.MineRLAgent(env, device="cpu", policy_kwargs=policy_kwargs, pi_head_kwargs=pi_head_kwargs)
    assert isinstance(hip_agent.policy, hip_policy.MinecraftAgentPolicy) and isinstance(ref_agent.policy, R.policy.MinecraftAgentPolicy)
    # the two module trees expose the same tensors under the same names
    hs, rs = hip_agent.policy.state_dict(), ref_agent.policy.state_dict()
    assert list(hs.keys()) == list(rs.keys()) or set(hs.keys()) == set(rs.keys())
    assert all(tuple(hs[k].shape) == tuple(rs[k].shape) for k in rs)
    # `.weights` written from either implementation loads STRICTLY into the other, through the wrappers' own load_weights
    wp = tmp_path / "ref.weights"
    torch.save(sd, wp)
    hip_agent.load_weights(str(wp))                       # agent.py:132-135: load_state_dict(strict=False) + reset()
    ref_agent.load_weights(str(wp))
    out = tmp_path / "hip.weights"
    torch.save(hip_agent.policy.state_dict(), out)
    res = ref_agent.policy.load_state_dict(torch.load(out, map_location="cpu"), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    res = hip_agent.policy.load_state_dict(ref_agent.policy.state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in sd.items():
        assert torch.equal(hip_agent.policy.state_dict()[k], v), k
    assert len(hip_agent.hidden_state) == cfg["n_layers"] and hip_agent.hidden_state[0][1][0].shape == (1, cfg["maxlen"], cfg["hidsize"])
    # a forward on the CPU must fail loudly: there is no CPU path
    with pytest.raises((RuntimeError, Exception)):
        hip_agent.get_action({"pov": torch.zeros(128, 128, 3, dtype=torch.uint8).numpy()})


def test_idm_wrapper_constructs_over_the_hip_policy(R):
    kw = O.idm_kwargs_for("tiny")
    im = R.idm
    ref_cls = im.InverseActionPolicy
    im.InverseActionPolicy = hip_policy.InverseActionPolicy
    try:
        agent = im.IDMAgent(idm_net_kwargs=kw, pi_head_kwargs=dict(temperature=2.0), device="cpu")
    finally:
        im.InverseActionPolicy = ref_cls
    ref_agent = im.IDMAgent(idm_net_kwargs=kw, pi_head_kwargs=dict(temperature=2.0), device="cpu")
    hs, rs = agent.policy.state_dict(), ref_agent.policy.state_dict()
    assert set(hs.keys()) == set(rs.keys()) and all(tuple(hs[k].shape) == tuple(rs[k].shape) for k in rs)
    res = ref_agent.policy.load_state_dict(hs, strict=True)
    assert not res.missing_keys and not res.unexpected_keys


def test_peaked_heads_and_structured_frames_have_the_claimed_properties():
    pk = O.policy_kwargs_for("1x")
    cfg = O.config_from_policy_kwargs(pk, dict(temperature=2.0))
    sd = O.synthetic_state_dict(cfg, seed=0)
    sp = O.synthetic_state_dict(cfg, seed=0, heads="peaked")
    assert all(torch.equal(sd[k], sp[k]) for k in sd if not k.startswith("pi_head."))
    for h in ("buttons", "camera"):
        b = sp[f"pi_head.{h}.linear_layer.bias"]
        top = b.topk(2).values
        assert abs(float(top[0] - top[1]) - 8.0) < 1e-5                       # the prior's top-2 gap: 8 logit units = 4 nat at T = 2
        assert torch.allclose(sp[f"pi_head.{h}.linear_layer.weight"], sd[f"pi_head.{h}.linear_layer.weight"] / 0.3)
    g = torch.Generator().manual_seed(3)
    fr = P.structured_frames(2, 5, g)
    assert fr.shape == (2, 5, 128, 128, 3) and fr.dtype == torch.uint8
    f = fr.float()
    # low-frequency content: neighbouring pixels are close, frames differ from each other
    assert float((f[:, :, 1:] - f[:, :, :-1]).abs().mean()) < 12.0 and float((f[0, 0] - f[0, 1]).abs().mean()) > 20.0
    # oracle margins on the peaked family (B x T = 2 x 4): median top-2 margin >= 1 nat on both heads
    img = P.structured_frames(2, 4, torch.Generator().manual_seed(5))
    ref = O.policy_forward(sp, cfg, img, torch.zeros(2, 4, dtype=torch.bool), O.initial_state(cfg, 2))
    for h in ("buttons", "camera"):
        t2 = ref[h].topk(2, -1).values
        assert float((t2[..., 0] - t2[..., 1]).median()) >= 1.0, h

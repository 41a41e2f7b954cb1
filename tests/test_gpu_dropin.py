"""Drop-in proof: the reference's OWN wrappers and file formats over the HIP policy.  Needs an MI355X.

north_star: "keeps the MinecraftAgentPolicy / MineRLAgent API surface and the .model / .weights loading format so it drops into
run_agent.py, run_inverse_dynamics_model.py and behavioural_cloning.py unchanged".  These tests import the UNMODIFIED reference
modules (oracle/_ref/vpt_reference.zip = /root/reference's agent.py, inverse_dynamics_model.py and lib/, byte for byte; the four
import stubs of oracle/ref_stubs stand in for gym / gym3 / minerl / cv2, which this image does not have), re-point the ONE name
each wrapper module imported from lib.policy to vpt_amd.lib.policy's class (INTEGRATION.md "one-line swap") and then run

  run_agent.py:8-24          pickle `.model` -> policy kwargs; MineRLAgent(env, ...); agent.load_weights(`.weights`)
  agent.py:106-206           MineRLAgent.__init__ / validate_env / load_weights / reset / get_action on a 640x360 observation
                             (resize_image, _env_obs_to_agent, policy.act, _agent_action_to_env -> MineRL action dict)
  inverse_dynamics_model.py  IDMAgent.__init__ / load_weights / predict_actions(128 frames)

side by side with the same wrappers over the reference's own policy on the CPU: the environment actions must be EQUAL wherever the
reference's top-2 margin exceeds the noise band (tests/parity.py), on trained-policy-like ("peaked") heads where that is (nearly)
every step.  Plus: a `.weights` written from the HIP policy loads into the reference policy with strict=True, and config 2's
sequence shape (2x, B = 1, T = 128) against the reference policy ITSELF rather than the oracle.
"""
import functools
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import vpt_amd  # noqa: E402,F401
from vpt_amd.lib import policy as hip_policy  # noqa: E402
from oracle import vpt_oracle as O  # noqa: E402
from tests import parity as P  # noqa: E402
from tests import ref_env  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module")
def R():
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    ref = ref_env.reference()
    if ref is None:
        pytest.skip("oracle/_ref/vpt_reference.zip is absent (it is packaged from /root/reference by __graft_entry__.build())")
    return ref


def _pov_frames(n, seed, h=360, w=640):
    """n MineRL-sized observations with low-frequency content (see parity.structured_frames)."""
    g = torch.Generator().manual_seed(seed)
    low = torch.randint(0, 256, (n, 3, 3, 5), generator=g).float()
    up = torch.nn.functional.interpolate(low, size=(h, w), mode="bilinear", align_corners=False)
    nz = torch.randint(-12, 13, (n, 3, h, w), generator=g).float()
    return (up + nz).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous().numpy()


def _deterministic(agent_obj, log):
    """MineRLAgent.get_action hard-codes stochastic=True (agent.py:201-204) and the two agents draw from different generators
    (CPU / GPU), so for the side-by-side run the bound method is wrapped to take the arg-max and to record the distribution.
    The wrapper module and the policy class are untouched."""
    orig = agent_obj.policy.act

    @functools.wraps(orig)
    def act(obs, first, state_in, stochastic=True, taken_action=None, return_pd=False):
        ac, state, res = orig(obs, first, state_in, stochastic=False, taken_action=taken_action, return_pd=True)
        log.append({k: v.detach().float().cpu() for k, v in res["pd"].items()})
        return ac, state, res
    agent_obj.policy.act = act
    return agent_obj


def test_minerl_agent_runs_unchanged_over_hip_policy(R, tmp_path):
    ag_mod = R.agent
    pk, ph = dict(ag_mod.POLICY_KWARGS), dict(ag_mod.PI_HEAD_KWARGS)       # the 2x foundation model, agent.py:16-38
    cfg = O.config_from_policy_kwargs(pk, ph)
    sd = O.synthetic_state_dict(cfg, seed=0, heads="peaked")
    model_path, weights_path = str(tmp_path / "synthetic-2x.model"), str(tmp_path / "synthetic-2x.weights")
    with open(model_path, "wb") as f:
        pickle.dump(ref_env.model_file_dict(pk, dict(temperature="2.0")), f)   # run_agent.py:14 casts the temperature to float
    torch.save(sd, weights_path)

    # ---- run_agent.py:8-16, statement for statement --------------------------------------------------------
    agent_parameters = pickle.load(open(model_path, "rb"))
    policy_kwargs = agent_parameters["model"]["args"]["net"]["args"]
    pi_head_kwargs = agent_parameters["model"]["args"]["pi_head_opts"]
    pi_head_kwargs["temperature"] = float(pi_head_kwargs["temperature"])
    env = ref_env.FakeEnv(ag_mod)
    ref_class = ag_mod.MinecraftAgentPolicy
    ag_mod.MinecraftAgentPolicy = hip_policy.MinecraftAgentPolicy           # <- the one-line swap
    try:
        agent = ag_mod.MineRLAgent(env, device=DEV, policy_kwargs=policy_kwargs, pi_head_kwargs=pi_head_kwargs)
    finally:
        ag_mod.MinecraftAgentPolicy = ref_class
    agent.load_weights(weights_path)
    assert isinstance(agent.policy, hip_policy.MinecraftAgentPolicy) and agent.policy.precision == "fp16"
    ref_agent = ag_mod.MineRLAgent(env, device="cpu", policy_kwargs=policy_kwargs, pi_head_kwargs=pi_head_kwargs)
    ref_agent.load_weights(weights_path)
    assert isinstance(ref_agent.policy, R.policy.MinecraftAgentPolicy)

    # ---- the unmodified acting loop (stochastic, as run_agent.py:22-24): a valid MineRL action every step -------------------
    n = 16
    frames = _pov_frames(n, seed=7)
    with torch.no_grad():
        ref_keys = set(ref_agent.get_action({"pov": frames[0]}).keys())     # what ActionTransformer.policy2env emits (a subset of the env's keys)
    ref_agent.reset()
    assert ref_keys <= set(ag_mod.TARGET_ACTION_SPACE.keys()) and {"camera", "attack", "forward", "inventory"} <= ref_keys
    for i in range(4):
        a = agent.get_action({"pov": frames[i]})
        assert set(a.keys()) == ref_keys
        assert a["camera"].shape == (1, 2) and all(np.asarray(v).shape[0] == 1 for v in a.values())
        assert all(int(np.asarray(a[k]).ravel()[0]) in (0, 1) for k in a if k != "camera")
    # ---- side by side, arg-max actions, state carried inside the agents -------------------------------------------
    log_h, log_r = [], []
    _deterministic(agent, log_h).reset()
    _deterministic(ref_agent, log_r).reset()
    acts_h = [agent.get_action({"pov": f}) for f in frames]
    torch.cuda.synchronize()
    with torch.no_grad():
        acts_r = [ref_agent.get_action({"pov": f}) for f in frames]
    for h in ("buttons", "camera"):
        got = torch.stack([d[h] for d in log_h], 1)          # [1, n, 1, classes]
        want = torch.stack([d[h] for d in log_r], 1)
        m = P.head_metrics(got, want)
        print(f"DROP-IN MineRLAgent[fp16] {h} vs the reference policy on CPU: {P.fmt(m)}")
        assert m["lp_l2"] < P.BOUNDS["fp16"]["lp_l2"] and m["argmax_safe_mismatch"] == 0
        assert m["argmax_safe_frac"] >= 0.9 and m["argmax_agree"] >= 0.9, m
    same = [all(np.array_equal(np.asarray(ah[k]), np.asarray(ar[k])) for k in ah) for ah, ar in zip(acts_h, acts_r)]
    print(f"DROP-IN MineRLAgent: identical MineRL action dicts at {sum(same)}/{n} steps")
    assert sum(same) >= n - 1

    # ---- `.weights` round trip: HIP policy -> file -> the reference policy, strict ------------------------------------
    out_path = str(tmp_path / "from-hip.weights")
    torch.save(agent.policy.state_dict(), out_path)
    loaded = torch.load(out_path, map_location="cpu")
    res = ref_agent.policy.load_state_dict(loaded, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in sd.items():
        assert torch.equal(loaded[k].cpu(), v), k


def test_idm_agent_runs_unchanged_over_hip_policy(R, tmp_path):
    idm_mod = R.idm
    kw = O.idm_kwargs_for("4x")
    ph = dict(temperature=2.0)
    cfg = O.idm_config_from_kwargs(kw, ph)
    sd = O.idm_synthetic_state_dict(cfg, seed=0, heads="peaked")
    weights_path = str(tmp_path / "synthetic-idm.weights")
    torch.save(sd, weights_path)
    ref_class = idm_mod.InverseActionPolicy
    idm_mod.InverseActionPolicy = hip_policy.InverseActionPolicy            # <- the one-line swap
    try:
        agent = idm_mod.IDMAgent(idm_net_kwargs=kw, pi_head_kwargs=ph, device=DEV)   # run_inverse_dynamics_model.py:101-103
    finally:
        idm_mod.InverseActionPolicy = ref_class
    agent.load_weights(weights_path)
    ref_agent = idm_mod.IDMAgent(idm_net_kwargs=kw, pi_head_kwargs=ph, device="cpu")
    ref_agent.load_weights(weights_path)
    n = 128
    clip = _pov_frames(n, seed=11)                   # [128, 360, 640, 3]: IDMAgent._video_obs_to_agent resizes every frame itself
    got = agent.predict_actions(clip)
    torch.cuda.synchronize()
    with torch.no_grad():
        want = ref_agent.predict_actions(clip)
    assert set(got.keys()) == set(want.keys())
    agree = {k: float((np.asarray(got[k]) == np.asarray(want[k])).mean()) for k in got}
    print("DROP-IN IDMAgent[fp16]: fraction of the 128 frames with the same predicted env action per key:", {k: round(v, 3) for k, v in agree.items()})
    assert np.asarray(got["camera"]).shape == np.asarray(want["camera"]).shape == (1, n, 2)
    assert min(agree.values()) >= 0.97


def test_config2_shape_against_the_reference_policy_itself(R):
    """BASELINE.json configs[1]'s sequence shape (2x, B = 1, T = 128) against the unmodified reference MinecraftAgentPolicy.forward
    (not the oracle), both operand formats, then a second chunk on the carried KV memory."""
    pk = O.policy_kwargs_for("2x")
    ph = dict(temperature=2.0)
    cfg = O.config_from_policy_kwargs(pk, ph)
    sd = O.synthetic_state_dict(cfg, seed=0)
    from gym3.types import DictType
    space = DictType(**R.action_mapping.CameraHierarchicalMapping(n_camera_bins=11).get_action_space_update())
    R.torch_util.set_default_torch_device("cpu")
    ref_pol = R.policy.MinecraftAgentPolicy(space, pk, ph)
    ref_pol.load_state_dict(sd, strict=False)
    ref_pol.eval()
    t = 128
    g = torch.Generator().manual_seed(31)
    imgs = [P.structured_frames(1, t, g) for _ in range(2)]
    first = torch.zeros(1, t, dtype=torch.bool)
    refs, st = [], ref_pol.initial_state(1)
    with torch.no_grad():
        for img in imgs:
            (pd, vpred, _), st = ref_pol({"img": img}, first, st)
            refs.append(dict(buttons=pd["buttons"], camera=pd["camera"], vpred=vpred))
    pol = hip_policy.MinecraftAgentPolicy(space, pk, ph)
    pol.load_state_dict(sd, strict=False)
    pol = pol.to(DEV)
    for mode in ("bf16", "fp16"):
        pol.set_precision(mode)
        sg = pol.initial_state(1)
        with torch.no_grad():
            for i, (img, ref) in enumerate(zip(imgs, refs)):
                (pd, vpred, _), sg = pol({"img": img.to(DEV)}, first.to(DEV), sg)
                torch.cuda.synchronize()
                m = P.policy_metrics(dict(buttons=pd["buttons"], camera=pd["camera"], vpred=vpred), ref)
                print(f"PARITY[{mode}] vs the REFERENCE policy, 2x T=128 chunk {i}: {P.fmt(m)}")
                P.check(m, mode, f"reference 2x T=128 chunk {i}")

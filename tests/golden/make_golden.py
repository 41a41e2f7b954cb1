"""Generate golden vectors from the LIVE, UNMODIFIED reference (run in the build container only).

    python tests/golden/make_golden.py            # writes tests/golden/policy_1x_seed0.npz

The reference (/root/reference) is imported as-is; the only additions are four import stubs
(oracle/ref_stubs) for packages that are absent here and never touched by the arithmetic
(gym3.types, gym.spaces, cv2, minerl).  Weights come from oracle.vpt_oracle.synthetic_state_dict
(seeded, every 1-D parameter randomised) and are loaded with strict=True, which also pins the
state_dict key set and shapes.  The GPU box has no /root/reference: tests there read only the .npz.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_stubs"))
sys.path.insert(0, REF)

from oracle import vpt_oracle as O  # noqa: E402


def build_reference_policy(name, sd, temperature=2.0):
    import lib.torch_util as tu
    tu.set_default_torch_device("cpu")
    from gym3.types import DictType
    from lib.action_mapping import CameraHierarchicalMapping
    from lib.policy import MinecraftAgentPolicy

    mapper = CameraHierarchicalMapping(n_camera_bins=11)
    space = DictType(**mapper.get_action_space_update())
    pol = MinecraftAgentPolicy(space, O.policy_kwargs_for(name), dict(temperature=temperature))
    missing, unexpected = pol.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    pol.eval()
    return pol


def synthetic_inputs(seed, b, t):
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (b, t, 128, 128, 3), generator=g, dtype=torch.uint8)
    return img


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    name = "1x"
    cfg = O.config_from_policy_kwargs(O.policy_kwargs_for(name), dict(temperature=2.0))
    sd = O.synthetic_state_dict(cfg, seed=0)
    pol = build_reference_policy(name, sd)

    out = {}
    b = 2
    state = pol.initial_state(b)
    # chunk A: T=4 from the initial state, sequence 1 flagged `first`
    # chunk B: T=3 with the carried KV memory (different t, exercises the ring update)
    # chunk C: T=1 (the run_agent.py shape)
    chunks = [("A", 4, [False, True]), ("B", 3, [False, False]), ("C", 1, [False, False])]
    for tag, t, first0 in chunks:
        img = synthetic_inputs(100 + ord(tag), b, t)
        first = torch.zeros(b, t, dtype=torch.bool)
        first[:, 0] = torch.tensor(first0)
        with torch.no_grad():
            (pd, vpred, _), state = pol({"img": img}, first, state)
            latent, _ = pol.net({"img": img}, state_in=pol.initial_state(b), context={"first": first}) if tag == "A" else ((None, None), None)
        out[f"{tag}_buttons"] = pd["buttons"].numpy()
        out[f"{tag}_camera"] = pd["camera"].numpy()
        out[f"{tag}_vpred"] = vpred.numpy()
        if tag == "A":
            out["A_latent"] = latent[0].numpy()
        for l, (m, (k, v)) in enumerate(state):
            out[f"{tag}_mask{l}"] = m.numpy()
            out[f"{tag}_Ktail{l}"] = k[:, -4:, :].numpy()
            out[f"{tag}_Vtail{l}"] = v[:, -4:, :].numpy()
            out[f"{tag}_Ksum{l}"] = k.double().sum(dim=(1, 2)).numpy()
            out[f"{tag}_Vsum{l}"] = v.double().sum(dim=(1, 2)).numpy()

    # act() on one more frame: deterministic action indices + log_prob + denormalised value
    img = synthetic_inputs(999, b, 1)[:, 0]
    ac, state2, res = pol.act({"img": img}, torch.zeros(b, dtype=torch.bool), state, stochastic=False)
    out["act_buttons"] = ac["buttons"].numpy()
    out["act_camera"] = ac["camera"].numpy()
    out["act_log_prob"] = res["log_prob"].numpy()
    out["act_vpred"] = res["vpred"].numpy()

    # per-layer taps of the CNN on two frames (means / stds / leading values), from the reference modules
    img = synthetic_inputs(7, 1, 2)
    with torch.no_grad():
        x = pol.net.img_preprocess(img)
        x = x.reshape(2, 128, 128, 3).permute(0, 3, 1, 2)
        for s, stack in enumerate(pol.net.img_process.cnn.stacks):
            x = stack(x)
            out[f"cnn_stack{s}_mean"] = x.mean(dim=(1, 2, 3)).numpy()
            out[f"cnn_stack{s}_std"] = x.std(dim=(1, 2, 3)).numpy()
            out[f"cnn_stack{s}_head"] = x[:, :8, :4, :4].numpy()
    path = os.path.join(HERE, f"policy_{name}_seed0.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()

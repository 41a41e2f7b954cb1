"""Golden vectors for logit masking (obs["mask"], lib/policy.py:257-266 -> lib/action_head.py:170-171) from the LIVE,
UNMODIFIED reference (build container only):   python tests/golden/make_golden_mask.py   -> tests/golden/mask_1x_seed0.npz
Same model / weights / import stubs as make_golden.py; one chunk B=2, T=3 with a seeded availability mask per head."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import build_reference_policy, synthetic_inputs, O  # noqa: E402


def masks(b, t, seed=5):
    g = torch.Generator().manual_seed(seed)
    mb = torch.rand(b, t, 1, 8641, generator=g) > 0.3
    mc = torch.rand(b, t, 1, 121, generator=g) > 0.5
    mb[..., 0] = True          # at least one available action per row
    mc[..., 60] = True
    return {"buttons": mb, "camera": mc}


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cfg = O.config_from_policy_kwargs(O.policy_kwargs_for("1x"), dict(temperature=2.0))
    sd = O.synthetic_state_dict(cfg, seed=0)
    pol = build_reference_policy("1x", sd)
    b, t = 2, 3
    img = synthetic_inputs(321, b, t)
    first = torch.zeros(b, t, dtype=torch.bool)
    mk = masks(b, t)
    with torch.no_grad():
        (pd, vpred, _), _ = pol({"img": img, "mask": mk}, first, pol.initial_state(b))
    ref = O.policy_forward(sd, cfg, img, first, O.initial_state(cfg, b), mask=mk)
    for k in ("buttons", "camera"):
        d = float((pd[k] - ref[k]).abs().max())
        print(f"oracle vs live reference, masked {k}: max|d| = {d:.2e}")
        assert d < 2e-4
    np.savez_compressed(os.path.join(HERE, "mask_1x_seed0.npz"), buttons=pd["buttons"].numpy(), camera=pd["camera"].numpy(),
                        argmax_buttons=pd["buttons"].argmax(-1).numpy(), argmax_camera=pd["camera"].argmax(-1).numpy())
    print("wrote mask_1x_seed0.npz")


if __name__ == "__main__":
    main()

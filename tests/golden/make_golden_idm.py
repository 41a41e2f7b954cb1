"""Golden vectors for the inverse dynamics model from the LIVE reference (build container only).

    python tests/golden/make_golden_idm.py        # writes tests/golden/idm_tiny_seed0.npz

Same protocol as make_golden.py; the IDM structure of SURVEY.md §8a at a CPU-sized width ("tiny": hid 512,
4 heads, IMPALA width 2 -> channels 32/64/64) because the real 4x IDM needs ~0.5 B parameters."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_stubs"))
sys.path.insert(0, "/root/reference")

from oracle import vpt_oracle as O  # noqa: E402


def main():
    import lib.torch_util as tu
    tu.set_default_torch_device("cpu")
    from gym3.types import DictType
    from lib.action_mapping import IDMActionMapping
    from lib.policy import InverseActionPolicy

    torch.manual_seed(0)
    torch.set_num_threads(8)
    kw = O.idm_kwargs_for("tiny")
    cfg = O.idm_config_from_kwargs(kw, dict(temperature=2.0))
    sd = O.idm_synthetic_state_dict(cfg, seed=0)
    space = DictType(**IDMActionMapping(n_camera_bins=11).get_action_space_update())
    pol = InverseActionPolicy(space, pi_head_kwargs=dict(temperature=2.0), idm_net_kwargs=kw)
    missing, unexpected = pol.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    pol.eval()
    g = torch.Generator().manual_seed(42)
    t = 12
    img = torch.randint(0, 256, (1, t, 128, 128, 3), generator=g, dtype=torch.uint8)
    first = torch.zeros(t, 1)  # what IDMAgent.predict_actions passes (inverse_dynamics_model.py:89)
    ac, state, res = pol.predict({"img": img}, first=first, state_in=pol.initial_state(1), deterministic=True)
    out = dict(buttons=res["pd"]["buttons"].numpy(), camera=res["pd"]["camera"].numpy(),
               ac_buttons=ac["buttons"].numpy(), ac_camera=ac["camera"].numpy(), log_prob=res["log_prob"].numpy())
    for l, (m, (k, v)) in enumerate(state):
        assert m is None and k.shape[1] == 0 and v.shape[1] == 0
    path = os.path.join(HERE, "idm_tiny_seed0.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", out["buttons"].shape, out["camera"].shape)


if __name__ == "__main__":
    main()

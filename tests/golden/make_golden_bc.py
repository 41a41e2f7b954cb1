"""Golden gradients of the behavioural-cloning loss from the LIVE reference (build container only).

    python tests/golden/make_golden_bc.py        # writes tests/golden/bc_1x_seed0.npz

The reference trains with B = 1, T = 1 samples (behavioural_cloning.py:86-123); its modules accept [B, T]
chunks (SURVEY.md §7 'Semantics of sequence BC'), so the golden run calls the unmodified
MinecraftAgentPolicy.forward on a [2, 3] chunk with a carried (detached) KV memory, takes
loss = -mean(log_prob) through the reference's own pi_head.logprob, and records every parameter gradient's
L2 norm plus a leading slice -- enough to pin the oracle's autograd path (tests/test_oracle_golden.py)."""
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
from tests.golden.make_golden import build_reference_policy, synthetic_inputs  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cfg = O.config_from_policy_kwargs(O.policy_kwargs_for("1x"), dict(temperature=2.0))
    sd = O.synthetic_state_dict(cfg, seed=0)
    pol = build_reference_policy("1x", sd)
    b, t = 2, 3
    g = torch.Generator().manual_seed(5)
    # warm the KV memory with one chunk (no grad), then the training chunk
    state = pol.initial_state(b)
    with torch.no_grad():
        (_, _, _), state = pol({"img": synthetic_inputs(300, b, 4)}, torch.zeros(b, 4, dtype=torch.bool), state)
    from lib.tree_util import tree_map
    state = tree_map(lambda x: x.detach(), state)
    img = synthetic_inputs(301, b, t)
    first = torch.zeros(b, t, dtype=torch.bool)
    ab = torch.randint(0, 8641, (b, t), generator=g)
    ac = torch.randint(0, 121, (b, t), generator=g)
    (pd, vpred, _), _ = pol({"img": img}, first, state)
    log_prob = pol.pi_head.logprob({"buttons": ab.unsqueeze(-1), "camera": ac.unsqueeze(-1)}, pd)  # [b, t]
    loss = -log_prob.mean()
    pol.zero_grad()
    loss.backward()
    out = {"loss": np.float32(loss.item()), "act_buttons": ab.numpy(), "act_camera": ac.numpy()}
    for name, p in pol.named_parameters():
        gr = p.grad if p.grad is not None else torch.zeros_like(p)
        out["norm/" + name] = np.float32(gr.double().norm().item())
        out["head/" + name] = gr.reshape(-1)[:16].numpy().copy()
    path = os.path.join(HERE, "bc_1x_seed0.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; loss", loss.item())


if __name__ == "__main__":
    main()

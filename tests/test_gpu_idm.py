"""Inverse dynamics model (BASELINE.json config 3) through the reference's InverseActionPolicy API on MI355X,
against the golden vectors of the live reference (tests/golden/make_golden_idm.py) and the oracle.
Structure-preserving `tiny` width (hid 512, channels 32/64/64): the released 4x IDM has ~0.5 B parameters.
Same bf16-operand tolerances as tests/test_gpu_policy.py."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import vpt_amd  # noqa: E402,F401
from vpt_amd.lib.policy import InverseActionPolicy  # noqa: E402
from vpt_amd.lib.types import idm_action_space  # noqa: E402
from oracle import vpt_oracle as O  # noqa: E402
from tests import parity as P  # noqa: E402

DEV = "cuda"


@pytest.fixture(autouse=True)
def _inference_mode():
    """These tests exercise the inference engine (what act() / bench.py run); a grad-enabled call takes the autograd
    boundary instead (same kernels, activations kept) -- tests/test_gpu_training.py covers that path."""
    with torch.no_grad():
        yield


def _l2(a, ref):
    return float(np.linalg.norm((a - ref).ravel()) / np.linalg.norm(ref.ravel()))


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_idm_predict_vs_golden_and_oracle(mode):
    G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "idm_tiny_seed0.npz")))
    kw = O.idm_kwargs_for("tiny")
    cfg = O.idm_config_from_kwargs(kw, dict(temperature=2.0))
    sd = O.idm_synthetic_state_dict(cfg, seed=0)
    pol = InverseActionPolicy(idm_action_space(), pi_head_kwargs=dict(temperature=2.0), idm_net_kwargs=kw, precision=mode)
    missing, unexpected = pol.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    pol = pol.to(DEV)
    g = torch.Generator().manual_seed(42)
    t = 12
    img = torch.randint(0, 256, (1, t, 128, 128, 3), generator=g, dtype=torch.uint8)
    first = torch.zeros(t, 1, device=DEV)  # inverse_dynamics_model.py:89
    ac, state, res = pol.predict({"img": img.to(DEV)}, first=first, state_in=pol.initial_state(1), deterministic=True)
    torch.cuda.synchronize()
    assert res["pd"]["buttons"].shape == (1, t, 20, 2) and res["pd"]["camera"].shape == (1, t, 2, 11)
    assert ac["buttons"].shape == (1, t, 20) and ac["buttons"].dtype == torch.int64 and res["log_prob"].shape == (1, t)
    for m, (k, v) in state:
        assert m is None and k.shape == (1, 0, 512)
    eb = np.abs(res["pd"]["buttons"].cpu().numpy() - G["buttons"]).max()
    ec = np.abs(res["pd"]["camera"].cpu().numpy() - G["camera"]).max()
    lb, lc = _l2(res["pd"]["buttons"].cpu().numpy(), G["buttons"]), _l2(res["pd"]["camera"].cpu().numpy(), G["camera"])
    agree_b = float((ac["buttons"].cpu().numpy() == G["ac_buttons"]).mean())
    agree_c = float((ac["camera"].cpu().numpy() == G["ac_camera"]).mean())
    print(f"PARITY IDM vs golden: max|d| buttons {eb:.3e} camera {ec:.3e}; relL2 {lb:.3e} {lc:.3e}; "
          f"action agreement buttons {agree_b:.3f} camera {agree_c:.3f}")
    # log-probs of a binary / 11-way softmax are O(1): absolute bounds
    tol_abs, tol_l2 = (3e-2, 1.5e-2) if mode == "bf16" else (4e-3, 2e-3)
    assert eb < tol_abs and ec < tol_abs and lb < tol_l2 and lc < tol_l2
    # deterministic actions (policy.py:448-464, argmax per group): EQUAL to the live reference's outside the noise band
    for head, got_ac, want_ac in (("buttons", ac["buttons"], G["ac_buttons"]), ("camera", ac["camera"], G["ac_camera"])):
        hm = P.head_metrics(res["pd"][head], G[head])
        assert hm["argmax_safe_mismatch"] == 0, (head, hm)
        print(f"ACTIONS[{mode}] IDM {head}: {P.fmt(hm)}")
    assert agree_b > 0.9 and agree_c > (0.9 if mode == "fp16" else 0.6)
    # second sequence in the batch must not leak across the temporal conv's sequence boundary
    img2 = torch.cat([img, torch.flip(img, dims=[1])], 0)
    (pd2, _, _), _ = pol({"img": img2.to(DEV)}, first=None, state_in=pol.initial_state(2))
    torch.cuda.synchronize()
    assert torch.allclose(pd2["buttons"][0], res["pd"]["buttons"][0], atol=2e-3)

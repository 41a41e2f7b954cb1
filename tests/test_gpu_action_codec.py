"""Action codec on the device through the reference-shaped classes: bit-exact against the golden
vectors of the live reference and against the numpy oracle on fresh inputs, numpy and CUDA-tensor entry points."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import vpt_amd  # noqa: E402,F401
from vpt_amd.lib.action_mapping import CameraHierarchicalMapping, IDMActionMapping  # noqa: E402
from vpt_amd.lib.actions import ActionTransformer  # noqa: E402
from oracle import action_codec as A  # noqa: E402

G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "actions_seed0.npz")))
MU = dict(camera_binsize=2, camera_maxval=10, camera_mu=10, camera_quantization_scheme="mu_law")    # agent.py:40-45


def test_from_and_to_factored_bit_exact():
    m = CameraHierarchicalMapping(n_camera_bins=11)
    ff = m.from_factored(dict(buttons=G["buttons"], camera=G["camera"]))
    assert ff["buttons"].dtype == np.int64 and ff["buttons"].shape == (4096, 1)
    assert np.array_equal(ff["buttons"], G["ff_buttons"]) and np.array_equal(ff["camera"], G["ff_camera"])
    tf = m.to_factored(dict(buttons=G["joint_b"][:, None], camera=G["joint_c"][:, None]))
    assert np.array_equal(tf["buttons"], G["tf_buttons"]) and np.array_equal(tf["camera"], G["tf_camera"])
    # CUDA tensors stay on the device, [B, T, 1] leading shapes as produced by the policy heads
    jb = torch.as_tensor(G["joint_b"][:8640].reshape(64, 135, 1)).cuda()
    jc = torch.as_tensor(G["joint_c"][:8640].reshape(64, 135, 1)).cuda()
    tf2 = m.to_factored(dict(buttons=jb, camera=jc))
    assert tf2["buttons"].is_cuda and tf2["buttons"].shape == (64, 135, 20)
    assert np.array_equal(tf2["buttons"].cpu().numpy().reshape(-1, 20), G["tf_buttons"][:8640])
    assert m.get_zero_action() == {"buttons": 0} and m.get_action_space_update()["buttons"].eltype.n == 8641
    ident = IDMActionMapping(n_camera_bins=11)
    assert ident.to_factored(tf) is tf


def test_fresh_inputs_against_oracle_at_scale():
    rng = np.random.default_rng(7)
    n = 64 * 128 * 8                                                   # a BC batch of labels per GPU, x8
    buttons = (rng.random((n, 20)) < 0.25).astype(np.int64)
    camera = rng.integers(0, 11, (n, 2)).astype(np.int64)
    m = CameraHierarchicalMapping(n_camera_bins=11)
    ff = m.from_factored(dict(buttons=buttons, camera=camera))
    jb, jc = A.from_factored(buttons, camera)
    assert np.array_equal(ff["buttons"][:, 0], jb) and np.array_equal(ff["camera"][:, 0], jc)
    tf = m.to_factored(ff)
    ob, oc = A.to_factored(jb, jc)
    assert np.array_equal(tf["buttons"], ob) and np.array_equal(tf["camera"], oc)
    again = m.from_factored(tf)                                        # idempotence: factored -> joint -> factored -> joint
    assert np.array_equal(again["buttons"], ff["buttons"]) and np.array_equal(again["camera"], ff["camera"])


def test_camera_quantizer_bit_exact():
    t = ActionTransformer(**MU)
    assert t.camera_zero_bin() == 5
    assert np.array_equal(t.discretize_camera(G["angles"]), G["disc_mu"])
    grid = np.arange(11)[:, None].repeat(2, 1)
    und = t.undiscretize_camera(grid)
    assert und.dtype == np.float64 and np.allclose(und, G["undisc_mu"], rtol=1e-14, atol=1e-14)   # pow / log: last-ulp libm differences allowed
    lin = ActionTransformer(camera_binsize=2, camera_maxval=10, camera_quantization_scheme="linear")
    assert np.array_equal(lin.discretize_camera(G["angles"]), G["disc_lin"])
    assert np.array_equal(lin.undiscretize_camera(grid), G["undisc_lin"].astype(np.int64))
    rng = np.random.default_rng(8)
    ang = rng.normal(0, 4, (200000, 2))
    assert np.array_equal(t.discretize_camera(ang), A.discretize(ang))
    assert np.array_equal(t.discretize_camera(t.undiscretize_camera(grid)), grid)     # bin centres survive the round trip
    env = t.policy2env(dict(buttons=G["tf_buttons"][:16], camera=G["tf_camera"][:16]))
    assert set(env) == set(A.BUTTONS_ALL) | {"camera"} and env["camera"].shape == (16, 2)
    pol = t.env2policy(dict(camera=ang[:16], attack=np.ones(16)))
    assert pol["buttons"].shape == (16, 20) and pol["buttons"][:, 0].all() and not pol["buttons"][:, 1:].any()


def test_out_of_range_indices_fail_loudly():
    """The reference raises (IndexError / KeyError) on indices outside its tables; the kernels decode blindly, so the op checks."""
    m = CameraHierarchicalMapping(n_camera_bins=11)
    bad_joint = dict(buttons=np.array([[8641]], dtype=np.int64), camera=np.array([[60]], dtype=np.int64))
    with pytest.raises(IndexError):
        m.to_factored(bad_joint)
    bad_cam = dict(buttons=np.zeros((1, 20), dtype=np.int64), camera=np.array([[11, 5]], dtype=np.int64))
    with pytest.raises(IndexError):
        m.from_factored(bad_cam)


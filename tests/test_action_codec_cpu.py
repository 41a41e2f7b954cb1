"""The action-codec oracle (oracle/action_codec.py) against the golden vectors of the live reference
(tests/golden/make_golden_actions.py), and against the live reference itself when /root/reference is present."""
import os
import sys

import numpy as np
import pytest

from oracle import action_codec as A

G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "actions_seed0.npz")))


def test_from_factored_matches_reference_golden():
    jb, jc = A.from_factored(G["buttons"], G["camera"])
    assert np.array_equal(jb, G["ff_buttons"][:, 0]) and np.array_equal(jc, G["ff_camera"][:, 0])
    assert (jb == A.JOINT_INVENTORY).sum() > 100 and len(np.unique(jb)) > 500      # the vectors do exercise the space


def test_to_factored_matches_reference_golden_for_every_joint_index():
    b, c = A.to_factored(G["joint_b"], G["joint_c"])
    assert np.array_equal(b, G["tf_buttons"]) and np.array_equal(c, G["tf_camera"])
    # round trip joint -> factored -> joint: the identity, except that "camera meta action on" with the null camera bin
    # comes back as "off" (from_factored derives the flag from the camera bins, lib/action_mapping.py:187-188)
    jb, jc = A.from_factored(b, c)
    on_with_null = (G["joint_b"] != A.JOINT_INVENTORY) & (G["joint_b"] % 2 == 1) & (G["joint_c"] == 60)
    assert np.array_equal(jb[~on_with_null], G["joint_b"][~on_with_null])
    assert np.array_equal(jb[on_with_null], G["joint_b"][on_with_null] - 1) and on_with_null.sum() > 10


def test_camera_quantizer_matches_reference_golden():
    assert np.array_equal(A.discretize(G["angles"], mu_law=True), G["disc_mu"])
    assert np.array_equal(A.discretize(G["angles"], mu_law=False), G["disc_lin"])
    grid = np.arange(11)[:, None].repeat(2, 1)
    assert np.array_equal(A.undiscretize(grid, mu_law=True), G["undisc_mu"])
    assert np.array_equal(A.undiscretize(grid, mu_law=False).astype(np.float64), G["undisc_lin"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/lib"), reason="live reference not present (GPU box)")
def test_oracle_matches_live_reference_on_fresh_inputs():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle", "ref_stubs"))
    sys.path.insert(0, "/root/reference")
    from lib.action_mapping import CameraHierarchicalMapping
    from lib.actions import ActionTransformer
    rng = np.random.default_rng(123)
    buttons = (rng.random((2000, 20)) < 0.3).astype(np.int64)
    camera = rng.integers(0, 11, (2000, 2)).astype(np.int64)
    ref = CameraHierarchicalMapping(n_camera_bins=11).from_factored(dict(buttons=buttons.copy(), camera=camera.copy()))
    jb, jc = A.from_factored(buttons, camera)
    assert np.array_equal(jb, ref["buttons"][:, 0]) and np.array_equal(jc, ref["camera"][:, 0])
    t = ActionTransformer(camera_binsize=2, camera_maxval=10, camera_mu=10, camera_quantization_scheme="mu_law")
    ang = rng.normal(0, 4, (5000, 2))
    assert np.array_equal(A.discretize(ang), t.discretize_camera(ang))

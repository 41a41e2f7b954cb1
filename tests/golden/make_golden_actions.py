"""Golden vectors of the action codec from the LIVE, UNMODIFIED reference (run in the build container only).

    python tests/golden/make_golden_actions.py       # writes tests/golden/actions_seed0.npz

Inputs: random factored actions, every joint index, adversarial button patterns (several buttons of one group,
forward+back, left+right, inventory with other buttons, inventory values other than 1), camera angles incl. the clip
range and values sitting on mu-law bin edges."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_stubs"))
sys.path.insert(0, "/root/reference")


def inputs(seed=0):
    rng = np.random.default_rng(seed)
    n = 4096
    buttons = (rng.random((n, 20)) < 0.15).astype(np.int64)
    buttons[: n // 8] = (rng.random((n // 8, 20)) < 0.6).astype(np.int64)      # dense presses: tie rules
    buttons[n // 8: n // 8 + 64, 10] = rng.integers(0, 4, 64)                   # inventory values 0..3 (== 1 matters)
    buttons[-1] = 0
    camera = rng.integers(0, 11, (n, 2)).astype(np.int64)
    camera[::5] = 5                                                            # null camera
    angles = np.concatenate([rng.uniform(-15, 15, (2000, 2)), rng.normal(0, 1.0, (2000, 2)), np.zeros((4, 2)),
                             np.array([[-10.0, 10.0], [10.0000001, -10.0000001], [0.6, -0.6], [1e-9, -1e-9]])])
    joint_b = np.arange(8641, dtype=np.int64)
    joint_c = rng.integers(0, 121, 8641).astype(np.int64)
    return buttons, camera, angles, joint_b, joint_c


def main():
    from lib.action_mapping import CameraHierarchicalMapping
    from lib.actions import ActionTransformer
    buttons, camera, angles, joint_b, joint_c = inputs()
    mapper = CameraHierarchicalMapping(n_camera_bins=11)
    out = dict(buttons=buttons, camera=camera, angles=angles, joint_b=joint_b, joint_c=joint_c)
    ff = mapper.from_factored(dict(buttons=buttons.copy(), camera=camera.copy()))
    out["ff_buttons"], out["ff_camera"] = ff["buttons"], ff["camera"]
    tf = mapper.to_factored(dict(buttons=joint_b[:, None], camera=joint_c[:, None]))
    out["tf_buttons"], out["tf_camera"] = tf["buttons"], tf["camera"]
    for tag, kw in (("mu", dict(camera_binsize=2, camera_maxval=10, camera_mu=10, camera_quantization_scheme="mu_law")),   # agent.py:40-45
                    ("lin", dict(camera_binsize=2, camera_maxval=10, camera_quantization_scheme="linear"))):
        t = ActionTransformer(**kw)
        bins = t.discretize_camera(angles)
        out[f"disc_{tag}"] = bins
        out[f"undisc_{tag}"] = np.asarray(t.undiscretize_camera(np.arange(11)[:, None].repeat(2, 1)), dtype=np.float64)
    path = os.path.join(HERE, "actions_seed0.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()

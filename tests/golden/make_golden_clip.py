"""Golden vectors of the clip data path from the LIVE, UNMODIFIED reference (run in the build container only).

    python tests/golden/make_golden_clip.py     # writes tests/golden/clip_seed0.npz and tests/golden/clip_actions_seed0.json.gz

What is pinned: run_inverse_dynamics_model.json_action_to_env_action (every synthetic step below goes through the reference's
own function) and data_loader.composite_images_with_alpha (the reference's own function on random frames / cursors, cursor
positions inside, clipped by the right / bottom edge, and outside the frame).  The loader loop around them (data_loader.py:77-118:
stuck attack, hotbar tracking, null filter, cursor scaling) only exists inside a worker process that needs cv2.VideoCapture, so
the generator re-runs those statements here around the reference's json_action_to_env_action.  cv2.resize is NOT available in
this image (oracle/ref_stubs/cv2.py is an import stub): no golden vector for the resize (oracle/clip_oracle.py: parity unpinned)."""
import gzip
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_stubs"))
sys.path.insert(0, "/root/reference")

KEYS = ["key.keyboard.escape", "key.keyboard.s", "key.keyboard.q", "key.keyboard.w", "key.keyboard.e", "key.keyboard.space",
        "key.keyboard.a", "key.keyboard.d", "key.keyboard.left.shift", "key.keyboard.left.control", "key.keyboard.f",
        "key.keyboard.t", "key.keyboard.f3", "key.keyboard.tab"] + [f"key.keyboard.{i}" for i in range(0, 10)]


def synthetic_recording(rng, n, stuck_start):
    steps = []
    hotbar = 0
    for i in range(n):
        kind = rng.integers(0, 10)
        keys = [] if kind < 3 else list(rng.choice(KEYS, size=rng.integers(1, 4), replace=False))
        if kind == 9:
            keys = ["key.keyboard.t", "key.keyboard.f3"]          # only keys the agent does not use: still a null action
        moving = rng.random() < 0.5
        dx = float(rng.choice([0.0, 1.0, -3.0, 6.666, -0.4, 1500.0, -2400.0, 7.0])) if moving else 0.0
        dy = float(rng.choice([0.0, -1.0, 2.5, -6.7, 0.3, 1300.0, 13.0])) if moving else 0.0
        if rng.random() < 0.1:
            dx, dy = int(dx), int(dy)                             # some recorders write integers
        buttons = [int(b) for b in np.flatnonzero(rng.random(3) < 0.2)]
        if stuck_start and i < 6 and 0 not in buttons:
            buttons = [0] + buttons
        new_buttons = [0] if (stuck_start and i == 0) else ([int(b) for b in buttons if rng.random() < 0.3 and not (stuck_start and i < 6 and b == 0)])
        if rng.random() < 0.15:
            hotbar = int(rng.integers(0, 9))
        steps.append({"keyboard": {"keys": [str(k) for k in keys]},
                      "mouse": {"x": float(rng.uniform(0, 1279.9)), "y": float(rng.uniform(0, 719.9)), "dx": dx, "dy": dy,
                                "buttons": buttons, "newButtons": new_buttons},
                      "hotbar": hotbar, "isGuiOpen": bool(rng.random() < 0.3), "tick": i})
    return steps


def reference_loop(json_data, frame_height, json_action_to_env_action):
    """data_loader.py:77-118 (statements re-run here; the action conversion is the reference's function)."""
    import copy
    out = []
    attack_is_stuck = False
    last_hotbar = 0
    for i in range(len(json_data)):
        step_data = copy.deepcopy(json_data[i])
        if i == 0:
            if step_data["mouse"]["newButtons"] == [0]:
                attack_is_stuck = True
        elif attack_is_stuck:
            if 0 in step_data["mouse"]["newButtons"]:
                attack_is_stuck = False
        if attack_is_stuck:
            step_data["mouse"]["buttons"] = [button for button in step_data["mouse"]["buttons"] if button != 0]
        action, is_null_action = json_action_to_env_action(step_data)
        current_hotbar = step_data["hotbar"]
        if current_hotbar != last_hotbar:
            action["hotbar.{}".format(current_hotbar + 1)] = 1
        last_hotbar = current_hotbar
        if is_null_action:
            continue
        rec = {"index": i, "action": {k: (v.tolist() if isinstance(v, np.ndarray) else int(v)) for k, v in action.items()}, "gui": bool(step_data["isGuiOpen"])}
        if step_data["isGuiOpen"]:
            camera_scaling_factor = frame_height / 720
            rec["cursor"] = [int(step_data["mouse"]["x"] * camera_scaling_factor), int(step_data["mouse"]["y"] * camera_scaling_factor)]
        out.append(rec)
    return out


def main():
    from run_inverse_dynamics_model import json_action_to_env_action
    from data_loader import composite_images_with_alpha
    rng = np.random.default_rng(0)
    recs = []
    for n, stuck, height in ((200, False, 360), (120, True, 720), (40, True, 360)):
        steps = synthetic_recording(rng, n, stuck)
        per_step = []
        for s in steps:
            a, null = json_action_to_env_action(s)
            per_step.append({"action": {k: (v.tolist() if isinstance(v, np.ndarray) else int(v)) for k, v in a.items()}, "null": bool(null),
                             "camera_dtype_kind": a["camera"].dtype.kind})
        recs.append({"frame_height": height, "steps": steps, "per_step": per_step, "loop": reference_loop(steps, height, json_action_to_env_action)})
    with gzip.open(os.path.join(HERE, "clip_actions_seed0.json.gz"), "wt") as fh:
        json.dump(recs, fh)

    # composite: frames 48 x 64, a 16 x 16 BGRA cursor with every alpha value, positions inside / clipped / outside
    cursor = rng.integers(0, 256, (16, 16, 4), dtype=np.uint8)
    cursor[:, :, 3] = np.arange(256, dtype=np.uint8).reshape(16, 16)
    alpha = cursor[:, :, 3:] / 255.0                      # data_loader.py:56
    image = cursor[:, :, :3]
    positions = [(0, 0), (5, 9), (48, 32), (56, 40), (63, 47), (64, 10), (10, 48), (70, 60), (50, 0), (0, 33)]
    frames = rng.integers(0, 256, (len(positions), 48, 64, 3), dtype=np.uint8)
    out = frames.copy()
    for k, (x, y) in enumerate(positions):
        composite_images_with_alpha(out[k], image, alpha, x, y)
    np.savez_compressed(os.path.join(HERE, "clip_seed0.npz"), cursor_bgra=cursor, frames=frames, positions=np.asarray(positions), composited=out)
    print("wrote clip_seed0.npz, clip_actions_seed0.json.gz:", sum(len(r["steps"]) for r in recs), "steps,", sum(len(r["loop"]) for r in recs), "kept")


if __name__ == "__main__":
    main()

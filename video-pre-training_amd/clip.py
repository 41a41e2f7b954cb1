"""Clip data path: recorded contractor clip -> (128 x 128 x 3 RGB uint8 frames, env actions) for the BC loop.

Host-side mirror of the reference's loader body, batched over a whole clip, with the pixel work on the GPU:

  reference (per step, CPU)                                         here
  ----------------------------------------------------------------  -----------------------------------------------------
  json_action_to_env_action          run_inverse_dynamics_model.py:80-125   json_action_to_env_action (same name, same result)
  stuck attack / hotbar / null filter / cursor position             clip_steps(json_data, frame_height) -> ClipSteps
                                      data_loader.py:77-118
  composite_images_with_alpha, BGR2RGB, resize_image                ClipFrameProcessor(frames_bgr, steps) -> one HIP launch
                                      data_loader.py:34-46,113-122; agent.py:100-103   (vpt_clip_frames, csrc/vpt_clip.hip)

mp4 decoding (cv2.VideoCapture, data_loader.py:62,106) is NOT here: it is CPU / video-engine work outside the device path, and
cv2 is not part of this image; the processor takes decoded BGR frames (as VideoCapture.read() returns them) already in HBM.
The null-action filter drops a step AFTER its frame was decoded (data_loader.py:106-112): `keep` indexes the decoded frames.
"""
import copy
from dataclasses import dataclass
from typing import List

import numpy as np
import torch

from . import ops

AGENT_RESOLUTION = (128, 128)                    # agent.py:83
MINEREC_ORIGINAL_HEIGHT_PX = 720                 # data_loader.py:20
CAMERA_SCALER = 360.0 / 2400.0                   # run_inverse_dynamics_model.py:77

# recorded key name -> env action key (run_inverse_dynamics_model.py:17-38); keys outside the table are ignored
_KEY_TO_ACTION = {"key.keyboard." + k: v for k, v in (
    ("escape", "ESC"), ("s", "back"), ("q", "drop"), ("w", "forward"), ("e", "inventory"), ("space", "jump"), ("a", "left"),
    ("d", "right"), ("left.shift", "sneak"), ("left.control", "sprint"), ("f", "swapHands"))}
_KEY_TO_ACTION.update({f"key.keyboard.{i}": f"hotbar.{i}" for i in range(1, 10)})
_MOUSE_TO_ACTION = ((0, "attack"), (1, "use"), (2, "pickItem"))               # :113-121
# key order of the reference's NOOP_ACTION template (:41-66)
_ACTION_KEYS = (["ESC", "back", "drop", "forward"] + [f"hotbar.{i}" for i in range(1, 10)]
                + ["inventory", "jump", "left", "right", "sneak", "sprint", "swapHands", "camera", "attack", "use", "pickItem"])


def json_action_to_env_action(json_action):
    """run_inverse_dynamics_model.py:80-125: one recorded step -> (MineRL env action dict, is_null_action).

    Same result as the reference, key order included.  Two of its quirks are kept: the camera entry is an INTEGER numpy array
    (the reference assigns dy * 0.15, dx * 0.15 into np.array([0, 0]), which truncates toward zero), and a step is "null" only
    if no mapped key, no mouse motion and none of the three mouse buttons is present.  (The reference's |camera| > 180 reset
    sits in the dx == dy == 0 branch, where the camera is already zero: it never fires and is not reproduced.)"""
    pressed = {_KEY_TO_ACTION[k] for k in json_action["keyboard"]["keys"] if k in _KEY_TO_ACTION}
    mouse = json_action["mouse"]
    pressed.update(name for button, name in _MOUSE_TO_ACTION if button in mouse["buttons"])
    moved = mouse["dx"] != 0 or mouse["dy"] != 0
    camera = np.array([int(mouse["dy"] * CAMERA_SCALER), int(mouse["dx"] * CAMERA_SCALER)])
    action = {k: (camera if k == "camera" else int(k in pressed)) for k in _ACTION_KEYS}
    return action, not (pressed or moved)


@dataclass
class ClipSteps:
    """What the loader decides per recorded step, for the steps it keeps."""
    keep: np.ndarray          # int64 [K]: indices (into the recording = into the decoded frames) of the non-null steps
    actions: List[dict]       # K env actions
    cursor_state: np.ndarray  # int32 [K, 3]: (GUI open, cursor x, cursor y) in frame pixels


def clip_steps(json_data, frame_height):
    """data_loader.py:77-118 for a whole recording (a list of the jsonl file's step dicts): the stuck-attack workaround, hotbar
    tracking, json_action_to_env_action, the null-action filter and the cursor position scaled to the video's height."""
    keep, actions, cursor = [], [], []
    attack_is_stuck = False
    last_hotbar = 0
    scale = frame_height / MINEREC_ORIGINAL_HEIGHT_PX
    for i, step_data in enumerate(json_data):
        step_data = copy.deepcopy(step_data)      # the reference edits the step in place; the caller's list stays untouched here
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
        gui = 1 if step_data["isGuiOpen"] else 0
        cx = cy = 0
        if gui:
            cx = int(step_data["mouse"]["x"] * scale)
            cy = int(step_data["mouse"]["y"] * scale)
            if cx < 0 or cy < 0:
                raise ValueError(f"step {i}: negative cursor position ({cx}, {cy}); the reference's slice arithmetic fails on it as well")
        keep.append(i)
        actions.append(action)
        cursor.append((gui, cx, cy))
    return ClipSteps(np.asarray(keep, dtype=np.int64), actions, np.asarray(cursor, dtype=np.int32).reshape(-1, 3))


class ClipFrameProcessor:
    """Cursor compositing + BGR->RGB + cv2.INTER_LINEAR resize of a batch of decoded frames in ONE HIP launch
    (data_loader.py:113-122).  `cursor_bgra`: the cursor sprite as cv2.imread(CURSOR_FILE, IMREAD_UNCHANGED) returns it
    (uint8 [16+,16+,4] BGRA; cropped to 16 x 16, alpha / 255.0 in fp64 like data_loader.py:53-57)."""

    def __init__(self, cursor_bgra, device="cuda", resolution=AGENT_RESOLUTION):
        cursor_bgra = np.asarray(cursor_bgra)
        if cursor_bgra.dtype != np.uint8 or cursor_bgra.ndim != 3 or cursor_bgra.shape[2] != 4:
            raise ValueError("cursor_bgra must be uint8 [h, w, 4] (BGRA)")
        cursor_bgra = cursor_bgra[:16, :16, :]
        self.device = torch.device(device)
        self.resolution = (int(resolution[0]), int(resolution[1]))          # (width, height) like cv2's dsize
        self.cursor_bgr = torch.from_numpy(np.ascontiguousarray(cursor_bgra[:, :, :3])).to(self.device)
        self.cursor_alpha = torch.from_numpy(np.ascontiguousarray(cursor_bgra[:, :, 3] / 255.0)).to(self.device)   # fp64

    def __call__(self, frames_bgr, cursor_state=None, out=None):
        """frames_bgr uint8 [K,H,W,3] on the device (the KEPT frames, BGR); cursor_state int32 [K,3] (ClipSteps.cursor_state) or
        None -> uint8 RGB [K,128,128,3]."""
        if cursor_state is not None and not torch.is_tensor(cursor_state):
            cursor_state = torch.from_numpy(np.ascontiguousarray(cursor_state, dtype=np.int32))
        if cursor_state is not None:
            if int(cursor_state[:, 1:].min()) < 0 if cursor_state.numel() else False:
                raise ValueError("negative cursor position")
            cursor_state = cursor_state.to(self.device)
        w, h = self.resolution
        return ops.clip_frames(frames_bgr, cursor_state, self.cursor_bgr if cursor_state is not None else None,
                               self.cursor_alpha if cursor_state is not None else None, out_hw=(h, w), out=out)

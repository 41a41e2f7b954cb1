"""CameraHierarchicalMapping / IDMActionMapping with the reference's interface (lib/action_mapping.py:11-234), the
joint <-> factored conversion computed on the GPU (vpt_action_from_factored / vpt_action_to_factored): the joint
button index is the mixed-radix number of the per-group choices, so no 8641 x 20 lookup table is needed.
numpy in -> numpy out, torch in -> CUDA torch out; there is no CPU arithmetic path."""
import itertools
from collections import OrderedDict
from typing import Dict

import torch

from .. import ops
from .actions import Buttons, _back, _to_device
from .types import Discrete, TensorType


class ActionMapping:
    BUTTONS_GROUPS = OrderedDict(
        hotbar=["none"] + [f"hotbar.{i}" for i in range(1, 10)],
        fore_back=["none", "forward", "back"],
        left_right=["none", "left", "right"],
        sprint_sneak=["none", "sprint", "sneak"],
        use=["none", "use"],
        drop=["none", "drop"],
        attack=["none", "attack"],
        jump=["none", "jump"],
    )

    def __init__(self, n_camera_bins: int = 11, device="cuda"):
        assert n_camera_bins % 2 == 1, "n_camera_bins should be odd"
        self.n_camera_bins = n_camera_bins
        self.camera_null_bin = n_camera_bins // 2
        self.device = device


class IDMActionMapping(ActionMapping):
    """Identity mapping (lib/action_mapping.py:106-123)."""

    def from_factored(self, ac: Dict) -> Dict:
        return ac

    def to_factored(self, ac: Dict) -> Dict:
        return ac

    def get_action_space_update(self):
        return {"buttons": TensorType(shape=(len(Buttons.ALL),), eltype=Discrete(2)),
                "camera": TensorType(shape=(2,), eltype=Discrete(self.n_camera_bins))}

    def get_zero_action(self):
        raise NotImplementedError()


class CameraHierarchicalMapping(ActionMapping):
    BUTTONS_GROUPS = ActionMapping.BUTTONS_GROUPS.copy()
    BUTTONS_GROUPS["camera"] = ["none", "camera"]
    BUTTONS_COMBINATIONS = list(itertools.product(*BUTTONS_GROUPS.values())) + ["inventory"]
    BUTTONS_COMBINATION_TO_IDX = {comb: i for i, comb in enumerate(BUTTONS_COMBINATIONS)}
    BUTTONS_IDX_TO_COMBINATION = {i: comb for i, comb in enumerate(BUTTONS_COMBINATIONS)}

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.camera_null_idx = self.camera_null_bin * self.n_camera_bins + self.camera_null_bin
        self._null_action = {"buttons": self.BUTTONS_COMBINATION_TO_IDX[tuple("none" for _ in range(len(self.BUTTONS_GROUPS)))]}

    def from_factored(self, ac: Dict) -> Dict:
        """{buttons [B,20], camera [B,2]} -> {buttons [B,1], camera [B,1]} joint indices."""
        assert ac["camera"].ndim == 2, f"bad camera label, {ac['camera']}"
        assert ac["buttons"].ndim == 2, f"bad buttons label, {ac['buttons']}"
        b, was_np = _to_device(ac["buttons"], torch.int64, self.device)
        c, _ = _to_device(ac["camera"], torch.int64, self.device)
        jb, jc = ops.action_from_factored(b, c, self.n_camera_bins)
        return dict(buttons=_back(jb[:, None], was_np), camera=_back(jc[:, None], was_np))

    def to_factored(self, ac: Dict) -> Dict:
        """{buttons [..., 1], camera [..., 1]} joint indices -> {buttons [..., 20], camera [..., 2]}."""
        assert ac["camera"].shape[-1] == 1
        assert ac["buttons"].shape[-1] == 1
        jb, was_np = _to_device(ac["buttons"], torch.int64, self.device)
        jc, _ = _to_device(ac["camera"], torch.int64, self.device)
        lead = jb.shape[:-1]
        b, c = ops.action_to_factored(jb.reshape(-1), jc.reshape(-1), self.n_camera_bins)
        return dict(buttons=_back(b.reshape(*lead, len(Buttons.ALL)), was_np), camera=_back(c.reshape(*lead, 2), was_np))

    def get_action_space_update(self):
        return {"camera": TensorType(shape=(1,), eltype=Discrete(self.n_camera_bins ** 2)),
                "buttons": TensorType(shape=(1,), eltype=Discrete(len(self.BUTTONS_COMBINATIONS)))}

    def get_zero_action(self):
        return self._null_action

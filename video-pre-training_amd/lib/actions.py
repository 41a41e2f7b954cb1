"""ActionTransformer / CameraQuantizer with the reference's interface (lib/actions.py:8-178), computed on the GPU
(vpt_camera_discretize / vpt_camera_undiscretize in libvpt_hip.so).  numpy in -> numpy out, torch in -> CUDA torch out;
there is no CPU arithmetic path."""
import numpy as np
import torch

from .. import ops


class Buttons:
    ATTACK, BACK, FORWARD, JUMP, LEFT, RIGHT = "attack", "back", "forward", "jump", "left", "right"
    SNEAK, SPRINT, USE, DROP, INVENTORY = "sneak", "sprint", "use", "drop", "inventory"
    ALL = [ATTACK, BACK, FORWARD, JUMP, LEFT, RIGHT, SNEAK, SPRINT, USE, DROP, INVENTORY] + [f"hotbar.{i}" for i in range(1, 10)]


class QuantizationScheme:
    LINEAR = "linear"
    MU_LAW = "mu_law"


def _to_device(x, dtype, device):
    """-> (contiguous CUDA tensor, was_numpy)."""
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype).contiguous(), False
    return torch.as_tensor(np.asarray(x), dtype=dtype).to(device).contiguous(), True


def _back(t, was_numpy):
    return t.cpu().numpy() if was_numpy else t


class CameraQuantizer:
    def __init__(self, camera_maxval, camera_binsize, quantization_scheme=QuantizationScheme.LINEAR, mu=5, device="cuda"):
        if quantization_scheme not in (QuantizationScheme.LINEAR, QuantizationScheme.MU_LAW):
            raise ValueError(f"quantization_scheme must be 'linear' or 'mu_law', got {quantization_scheme!r}")
        self.camera_maxval, self.camera_binsize, self.quantization_scheme, self.mu = camera_maxval, camera_binsize, quantization_scheme, mu
        self.device = device

    @property
    def _mu_law(self):
        return self.quantization_scheme == QuantizationScheme.MU_LAW

    def discretize(self, xy):
        t, was_np = _to_device(xy, torch.float64, self.device)
        return _back(ops.camera_discretize(t, self.camera_maxval, self.camera_binsize, self.mu, self._mu_law), was_np)

    def undiscretize(self, xy):
        t, was_np = _to_device(xy, torch.int64, self.device)
        if not self._mu_law:      # the reference's linear scheme stays in integer arithmetic (lib/actions.py:101)
            return _back(t * self.camera_binsize - self.camera_maxval, was_np)
        return _back(ops.camera_undiscretize(t, self.camera_maxval, self.camera_binsize, self.mu, True), was_np)


class ActionTransformer:
    """Transforms actions between the policy's arrays and the MineRL env format (lib/actions.py:111-178)."""

    def __init__(self, camera_maxval=10, camera_binsize=2, camera_quantization_scheme="linear", camera_mu=5, device="cuda"):
        self.camera_maxval, self.camera_binsize = camera_maxval, camera_binsize
        self.camera_quantization_scheme, self.camera_mu = camera_quantization_scheme, camera_mu
        self.quantizer = CameraQuantizer(camera_maxval=camera_maxval, camera_binsize=camera_binsize,
                                         quantization_scheme=camera_quantization_scheme, mu=camera_mu, device=device)

    def camera_zero_bin(self):
        return self.camera_maxval // self.camera_binsize

    def discretize_camera(self, xy):
        return self.quantizer.discretize(xy)

    def undiscretize_camera(self, pq):
        return self.quantizer.undiscretize(pq)

    def numpy_to_dict(self, acs):
        assert acs["buttons"].shape[-1] == len(Buttons.ALL), (
            f"Mismatched actions: {acs}; expected {len(Buttons.ALL)}:\n(  {Buttons.ALL})")
        out = {name: acs["buttons"][..., i] for (i, name) in enumerate(Buttons.ALL)}
        out["camera"] = self.undiscretize_camera(acs["camera"])
        return out

    def policy2env(self, acs):
        return self.numpy_to_dict(acs)

    def env2policy(self, acs):
        nbatch = acs["camera"].shape[0]
        dummy = np.zeros((nbatch,))
        return {"camera": self.discretize_camera(acs["camera"]),
                "buttons": np.stack([acs.get(k, dummy) for k in Buttons.ALL], axis=-1)}

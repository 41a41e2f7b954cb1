"""Minimal value types with the gym3.types surface the reference's policy code touches
(lib/action_head.py:263-275, lib/action_mapping.py:227-231).  Real gym3 objects are accepted wherever
these are: only `.shape`, `.eltype.n`, `.size` and dict iteration are used."""


class ValType:
    pass


class Discrete(ValType):
    def __init__(self, n):
        self.n = int(n)


class Real(ValType):
    pass


class TensorType(ValType):
    def __init__(self, eltype, shape):
        self.eltype = eltype
        self.shape = tuple(shape)

    @property
    def size(self):
        n = 1
        for s in self.shape:
            n *= s
        return n


class DictType(ValType, dict):
    def __init__(self, **kw):
        dict.__init__(self, **kw)


def minecraft_action_space(n_buttons: int = 8641, n_camera: int = 121):
    """The space CameraHierarchicalMapping.get_action_space_update() yields (lib/action_mapping.py:227-231):
    11 camera bins -> 121 joint camera actions, 8641 joint button combinations; insertion order camera, buttons."""
    return DictType(camera=TensorType(Discrete(n_camera), (1,)), buttons=TensorType(Discrete(n_buttons), (1,)))


def idm_action_space(n_buttons: int = 20, n_camera_bins: int = 11):
    """IDMActionMapping.get_action_space_update() (lib/action_mapping.py:110-115): 20 binary buttons, 2 x 11 camera bins."""
    return DictType(buttons=TensorType(Discrete(2), (n_buttons,)), camera=TensorType(Discrete(n_camera_bins), (2,)))

"""Minimal stand-ins for gym3.types used by lib/action_head.py:9, lib/action_mapping.py:7, lib/policy.py:7."""


class ValType:
    pass


class Discrete(ValType):
    def __init__(self, n):
        self.n = int(n)

    def __eq__(self, other):
        return isinstance(other, Discrete) and other.n == self.n


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

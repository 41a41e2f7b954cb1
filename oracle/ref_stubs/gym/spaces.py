import numpy as np


class Discrete:
    def __init__(self, n):
        self.n = n

    def __eq__(self, other):
        return isinstance(other, Discrete) and self.n == other.n


class Box:
    def __init__(self, low, high, shape):
        self.low, self.high, self.shape = low, high, tuple(shape)

    def __eq__(self, other):
        return (isinstance(other, Box) and self.shape == other.shape
                and np.all(self.low == other.low) and np.all(self.high == other.high))


class Dict:
    def __init__(self, spaces):
        self.spaces = dict(spaces)

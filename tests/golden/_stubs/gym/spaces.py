"""Minimal gym.spaces used by the reference for observation-space bookkeeping only."""


class Discrete:
    def __init__(self, n):
        self.n = n
        self.shape = ()


class Box:
    def __init__(self, low=None, high=None, shape=None, dtype=None):
        self.low, self.high, self.dtype = low, high, dtype
        self.shape = shape


class Tuple:
    def __init__(self, spaces):
        self.spaces = list(spaces)
        self.shape = None

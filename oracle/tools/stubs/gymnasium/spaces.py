class Discrete:
    def __init__(self, n):
        self.n = n
        self.shape = ()


class Box:
    def __init__(self, low, high, dtype=None):
        self.low, self.high, self.dtype, self.shape = low, high, dtype, low.shape

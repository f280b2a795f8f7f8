"""`gymnasium` when it is installed, otherwise the few names ACEnv needs.

The reference imports gymnasium 0.28.1 (pyproject.toml:40).  This image does not ship it, so the
stand-ins below carry just the attributes the reference's callers read: `Discrete.n`, `.shape`,
`Box.low/.high/.dtype/.shape`, `.sample()` and `.contains()`.
"""
import numpy as np

try:  # pragma: no cover - exercised only where gymnasium exists
    from gymnasium import Env
    from gymnasium.spaces import Box, Discrete

    HAVE_GYMNASIUM = True
except ImportError:
    HAVE_GYMNASIUM = False

    class Env:
        """Placeholder base class with gymnasium.Env's no-op surface."""

        metadata = {"render_modes": []}
        render_mode = None
        spec = None

        def close(self):
            pass

        @property
        def unwrapped(self):
            return self

    class Discrete:
        def __init__(self, n, seed=None, start=0):
            self.n, self.start, self.shape, self.dtype = int(n), int(start), (), np.int64
            self._rng = np.random.default_rng(seed)

        def sample(self):
            return int(self.start + self._rng.integers(self.n))

        def contains(self, x):
            return isinstance(x, (int, np.integer)) and self.start <= int(x) < self.start + self.n

        __contains__ = contains

        def __repr__(self):
            return f"Discrete({self.n})"

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            low, high = np.asarray(low), np.asarray(high)
            if shape is not None:
                low, high = np.broadcast_to(low, shape), np.broadcast_to(high, shape)
            self.dtype = np.dtype(dtype)
            self.low, self.high = low.astype(self.dtype), high.astype(self.dtype)
            self.shape = self.low.shape
            self._rng = np.random.default_rng(seed)

        def sample(self):
            return self._rng.integers(self.low, self.high, endpoint=True).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        __contains__ = contains

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

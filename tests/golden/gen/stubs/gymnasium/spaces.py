"""Stub spaces: shape/dtype containers with contains() and sample()."""
import numpy as np


class Space:
    def __init__(self, shape=None, dtype=None, seed=None):
        self.shape = shape
        self.dtype = None if dtype is None else np.dtype(dtype)
        self._rng = np.random.default_rng(seed)

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)


class Discrete(Space):
    def __init__(self, n, seed=None, start=0):
        super().__init__((), np.int64, seed)
        self.n = int(n)
        self.start = int(start)

    def contains(self, x):
        if isinstance(x, (bool, np.bool_)):
            return False
        if isinstance(x, (int, np.integer)):
            v = int(x)
        elif isinstance(x, np.ndarray) and x.shape == () and np.issubdtype(x.dtype, np.integer):
            v = int(x)
        else:
            return False
        return self.start <= v < self.start + self.n

    def sample(self):
        return int(self.start + self._rng.integers(self.n))


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        if shape is None:
            shape = np.shape(low)
        super().__init__(tuple(shape), dtype, seed)
        self.low = np.broadcast_to(np.asarray(low, dtype=np.float64), self.shape)
        self.high = np.broadcast_to(np.asarray(high, dtype=np.float64), self.shape)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return self._rng.uniform(lo, hi).astype(self.dtype)


class MultiDiscrete(Space):
    def __init__(self, nvec, dtype=np.int64, seed=None):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        super().__init__(self.nvec.shape, dtype, seed)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= 0) and np.all(x < self.nvec))

    def sample(self):
        return (self._rng.random(self.nvec.shape) * self.nvec).astype(self.dtype)


class MultiBinary(Space):
    def __init__(self, n, seed=None):
        self.n = n
        super().__init__((n,) if np.isscalar(n) else tuple(n), np.int8, seed)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all((x == 0) | (x == 1)))

    def sample(self):
        return self._rng.integers(0, 2, size=self.shape, dtype=np.int8)


class Dict(Space):
    def __init__(self, spaces=None, seed=None, **kw):
        super().__init__(None, None, seed)
        self.spaces = dict(spaces or {}, **kw)

    def __getitem__(self, k):
        return self.spaces[k]

    def contains(self, x):
        return isinstance(x, dict) and all(k in x and s.contains(x[k]) for k, s in self.spaces.items())

    def sample(self):
        return {k: s.sample() for k, s in self.spaces.items()}


class Tuple(Space):
    def __init__(self, spaces, seed=None):
        super().__init__(None, None, seed)
        self.spaces = tuple(spaces)

    def sample(self):
        return tuple(s.sample() for s in self.spaces)

"""Spaces for the vector façade: gymnasium's when it is importable, else attribute-compatible minimal
stand-ins (the build and GPU boxes have no gymnasium and no network)."""
import numpy as np

try:  # pragma: no cover - gymnasium is absent in the build image
    import gymnasium as _gym
    from gymnasium import spaces as _sp
    from gymnasium.vector.utils import batch_space as _batch_space
    HAVE_GYMNASIUM = True
    Box, Discrete, MultiDiscrete = _sp.Box, _sp.Discrete, _sp.MultiDiscrete
    VectorEnvBase = _gym.vector.VectorEnv

    def batch_space(space, n):
        return _batch_space(space, n)

except Exception:  # ModuleNotFoundError in this image
    HAVE_GYMNASIUM = False

    class _Space:
        def __init__(self, shape, dtype, seed=None):
            self.shape = tuple(shape)
            self.dtype = np.dtype(dtype)
            self._rng = np.random.default_rng(seed)

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)

        def __repr__(self):
            return f"{type(self).__name__}(shape={self.shape}, dtype={self.dtype})"

    class Box(_Space):
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            if shape is None:
                shape = np.shape(low)
            super().__init__(shape, dtype, seed)
            self.low = np.broadcast_to(np.asarray(low, dtype=np.float64), self.shape)
            self.high = np.broadcast_to(np.asarray(high, dtype=np.float64), self.shape)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1.0)
            hi = np.where(np.isfinite(self.high), self.high, 1.0)
            return self._rng.uniform(lo, hi).astype(self.dtype)

    class Discrete(_Space):
        def __init__(self, n, seed=None, start=0):
            super().__init__((), np.int64, seed)
            self.n, self.start = int(n), int(start)

        def contains(self, x):
            try:
                v = int(x)
            except (TypeError, ValueError):
                return False
            return v == x and self.start <= v < self.start + self.n

        def sample(self):
            return int(self.start + self._rng.integers(self.n))

    class MultiDiscrete(_Space):
        def __init__(self, nvec, dtype=np.int64, seed=None):
            self.nvec = np.asarray(nvec, dtype=np.int64)
            super().__init__(self.nvec.shape, dtype, seed)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= 0) and np.all(x < self.nvec))

        def sample(self):
            return (self._rng.random(self.nvec.shape) * self.nvec).astype(self.dtype)

    class VectorEnvBase:
        """Attribute surface of gymnasium.vector.VectorEnv (gymnasium 1.x)."""
        metadata = {}
        spec = None
        render_mode = None
        closed = False

        def close(self, **kwargs):
            if not self.closed:
                self.close_extras(**kwargs)
                self.closed = True

        def close_extras(self, **kwargs):
            pass

        @property
        def unwrapped(self):
            return self

        def __del__(self):
            try:
                self.close()
            except Exception:
                pass

    def batch_space(space, n):
        if isinstance(space, Box):
            return Box(np.broadcast_to(space.low, (n,) + space.shape), np.broadcast_to(space.high, (n,) + space.shape),
                       (n,) + space.shape, space.dtype)
        if isinstance(space, Discrete):
            return MultiDiscrete(np.full((n,), space.n, dtype=np.int64))
        if isinstance(space, MultiDiscrete):
            return MultiDiscrete(np.broadcast_to(space.nvec, (n,) + space.nvec.shape).copy())
        raise TypeError(space)

"""Build-owned minimal stand-in for the `gymnasium` package (NOT the real library).

`gymnasium` is not installed in the build container and there is no network, while every
hot-path module of the reference imports it at top level.  This stub provides only what
those modules touch at import/step time so the *reference's own Python* can be executed
unmodified to produce golden vectors (tests/golden/gen/*.py).  It is used by the fixture
generators only; it is never imported by the product or on the GPU box.

Seeding follows gymnasium >= 0.26 as published: `Env.reset(seed=s)` sets
`self._np_random = numpy.random.Generator(PCG64(SeedSequence(s)))`.
"""
import numpy as _np

from . import spaces  # noqa: F401
from .envs.registration import register  # noqa: F401
from .utils import seeding  # noqa: F401
from . import core  # noqa: F401


class Env:
    metadata = {}
    render_mode = None
    _np_random = None

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self._np_random, _ = seeding.np_random(seed)
        return None

    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random, _ = seeding.np_random()
        return self._np_random

    @np_random.setter
    def np_random(self, value):
        self._np_random = value

    def close(self):
        pass


core.Env = Env


def make(*a, **k):  # pragma: no cover - never used by the generators
    raise NotImplementedError("stub gymnasium has no registry")

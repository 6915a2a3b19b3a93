"""Stub of gymnasium.utils.seeding.np_random (published gymnasium >= 0.26 behaviour)."""
import numpy as np


def np_random(seed=None):
    ss = np.random.SeedSequence(seed)
    return np.random.Generator(np.random.PCG64(ss)), ss.entropy

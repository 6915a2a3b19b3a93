"""Shared helpers for the golden-vector generators (run in the BUILD container only).

The generators import the reference's Python from /root/reference (read-only, never copied)
under the build-owned stub `gymnasium`/`pygame` packages in ./stubs, drive it with recorded
action streams and write small .npz fixtures into tests/golden/.  The GPU box never runs
these scripts: /root/reference does not exist there.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.dirname(HERE)
REFERENCE = os.environ.get("CGE_REFERENCE", "/root/reference")

M64 = (1 << 64) - 1
GOLD = 0x9E3779B97F4A7C15
C_T = 0xD1342543DE82EF95


def mix64(z):
    """SplitMix64 finaliser (Steele/Lea/Flood) on python ints."""
    z &= M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def action_hash(a_seed, env, t, j=0):
    """Counter hash used as the synthetic action source everywhere (numpy / C oracle / HIP):
    u = mix64(mix64(a_seed + env*GOLD) + t*C_T + j);  action = ((u >> 32) * n) >> 32."""
    return mix64(mix64(a_seed + env * GOLD) + t * C_T + j)


def hash_action(a_seed, env, t, n, j=0):
    return ((action_hash(a_seed, env, t, j) >> 32) * n) >> 32


def hash_actions_np(a_seed, env_ids, t, n, j=0):
    """Vectorised numpy version of hash_action (uint64 wraparound arithmetic)."""
    with np.errstate(over="ignore"):
        e = np.asarray(env_ids, dtype=np.uint64)

        def mix(z):
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return z ^ (z >> np.uint64(31))

        h = mix(np.uint64(a_seed) + e * np.uint64(GOLD))
        u = mix(h + np.uint64((t * C_T + j) & M64))
        return (((u >> np.uint64(32)) * np.uint64(n)) >> np.uint64(32)).astype(np.int32)


def use_stubs():
    sys.dont_write_bytecode = True
    stubs = os.path.join(HERE, "stubs")
    if stubs not in sys.path:
        sys.path.insert(0, stubs)


def add_reference_dir(*parts):
    p = os.path.join(REFERENCE, *parts)
    if p not in sys.path:
        sys.path.insert(1, p)
    return p


class RunningHash:
    """sha256 over obs bytes + float64 reward bytes + flag bytes, SURVEY.md section 8c recipe."""

    def __init__(self):
        self.h = hashlib.sha256()

    def obs(self, obs):
        self.h.update(np.ascontiguousarray(obs).tobytes())

    def step(self, obs, reward, terminated, truncated):
        self.h.update(np.ascontiguousarray(obs).tobytes())
        self.h.update(np.float64(reward).tobytes())
        self.h.update(bytes([int(bool(terminated)), int(bool(truncated))]))

    def hexdigest(self):
        return self.h.hexdigest()


def versions():
    import platform
    return dict(python=platform.python_version(), numpy=np.__version__)

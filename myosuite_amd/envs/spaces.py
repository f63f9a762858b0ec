"""Minimal gym-space shim (gymnasium is absent from this image).

Mirrors the attributes of ``gym.spaces.Box`` that the reference and its agents
touch: ``low``, ``high``, ``shape``, ``dtype``, ``sample()``, ``contains()``
(myosuite/envs/env_base.py:143-155, 214-218).
"""
from __future__ import annotations

import numpy as np


class Box:
    def __init__(self, low, high, dtype=np.float32, seed=None):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        assert self.low.shape == self.high.shape
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)
        self._rng = np.random.default_rng(seed)

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


def batch_space(space: Box, n: int) -> Box:
    return Box(np.tile(space.low, (n, 1)), np.tile(space.high, (n, 1)), dtype=space.dtype)

"""Positional embeddings (``sorrel/observation/embedding.py:8-46``).

The embedding of a location is a pure function of (y, x) and the world size, so the batched
form is a ``[H, W, 2 * (scale[0] + scale[0])]`` table built once on the host in float64 with the
reference's arithmetic and gathered on the device by agent position."""
from __future__ import annotations

import numpy as np
import torch


def positional_embedding(location, world, scale) -> np.ndarray:
    """sin / cos of the two coordinates at ``scale[0]`` doubling frequencies each (the reference
    uses ``scale[0]`` for both axes, ``embedding.py:31,39``)."""
    x, y = location[0:2]
    h, w = world.height, world.width
    out = []
    for i in range(scale[0]):
        f = 2 * np.pi * (2**i) / h
        out += [np.sin(f * x), np.cos(f * x)]
    for j in range(scale[0]):
        f = 2 * np.pi * (2**j) / w
        out += [np.sin(f * y), np.cos(f * y)]
    return np.array(out)


def positional_embedding_table(world, scale, dtype=torch.float32) -> torch.Tensor:
    """``[H, W, 4 * scale[0]]`` on the world's device; float64 values rounded once to ``dtype``
    (what the reference's float32 replay rows hold)."""
    tab = np.zeros((world.height, world.width, 4 * scale[0]), dtype=np.float64)
    for y in range(world.height):
        for x in range(world.width):
            tab[y, x] = positional_embedding((y, x), world, scale)
    return torch.from_numpy(tab).to(dtype).to(world.device)

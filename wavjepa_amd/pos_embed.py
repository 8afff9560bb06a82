"""Fixed sin-cos positions (reference wavjepa/pos_embed.py:75-93): [sin(p w_j) | cos(p w_j)], w_j = 10000^(-j/(D/2)),
float64 math, sin half first (not interleaved)."""
from __future__ import annotations

import numpy as np


def get_1d_sincos_pos_embed_from_grid(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    if embed_dim % 2:
        raise ValueError("embed_dim must be even")
    half = embed_dim // 2
    omega = 1.0 / 10000 ** (np.arange(half, dtype=np.float64) / (embed_dim / 2.0))
    ang = np.einsum("m,d->md", np.asarray(pos, dtype=np.float64).reshape(-1), omega)
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)


def get_1d_sincos_pos_embed(embed_dim: int, length: int) -> np.ndarray:
    return get_1d_sincos_pos_embed_from_grid(embed_dim, np.arange(length, dtype=np.float64))

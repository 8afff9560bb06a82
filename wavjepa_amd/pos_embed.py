"""Fixed sin-cos positions (reference wavjepa/pos_embed.py:75-93): [sin(p w_j) | cos(p w_j)], w_j = 10000^(-j/(D/2)),
float64 math, sin half first (not interleaved); and the binaural table of pos_embed.py:122-151."""
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


def get_binaural_pos_embed(embed_dim: int, time_steps: int = 100) -> np.ndarray:
    """Positions for a binaural clip whose two channels are flattened channel-major (left tokens, then right tokens):
    [2 * time_steps, embed_dim].  The first embed_dim / 2 features code the time step (the same for both channels), the second
    half codes the channel: zeros for the left one, the sin-cos code of position 0 for the right one
    (reference wavjepa/pos_embed.py:122-151)."""
    if embed_dim % 2:
        raise ValueError("embed_dim must be even")
    half = embed_dim // 2
    time_embed = get_1d_sincos_pos_embed(half, time_steps)
    left = np.concatenate([time_embed, np.zeros((time_steps, half))], axis=1)
    right = np.concatenate([time_embed, np.tile(get_1d_sincos_pos_embed(half, 1), (time_steps, 1))], axis=1)
    return np.concatenate([left, right], axis=0)

"""Span masking (the branch of fairseq's compute_mask_indices the reference reaches, wavjepa/audio_masking.py:5-194:
static span length, overlap allowed, idc_select_ver=1, num_mask_ver=2).  CPU / NumPy, runs in data-loader workers."""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch


def compute_mask_indices(shape: Tuple[int, int], padding_mask: Optional[torch.Tensor], mask_prob: float, mask_length: int,
                         mask_type: str = "static", min_masks: int = 0, rng: Optional[np.random.Generator] = None,
                         **unused) -> torch.Tensor:
    """Returns a bool tensor of `shape` squeezed (like the reference) with True inside masked spans."""
    if mask_type != "static" or padding_mask is not None:
        raise NotImplementedError("only the static, unpadded branch used by the WavJEPA maskers is implemented")
    bsz, n = shape
    out = np.zeros((bsz, n), dtype=bool)
    rows = []
    for i in range(bsz):
        g = rng if rng is not None else np.random.default_rng(None)   # fresh OS-entropy generator per row, as upstream
        num = max(min_masks, int(mask_prob * n / float(mask_length) + g.random()))
        if num == 0:
            raise ValueError("this should never happens")
        min_len = mask_length
        if n - min_len <= num:
            min_len = n - num - 1
        starts = g.choice(n - min_len, num, replace=False)
        idx = (starts[:, None] + np.arange(mask_length)[None, :]).reshape(-1)
        rows.append(np.unique(idx[idx < n]))
    keep = min(len(r) for r in rows)       # require_same_masks: drop extras so every row masks the same count
    for i, r in enumerate(rows):
        if len(r) > keep:
            g = rng if rng is not None else np.random.default_rng(None)
            r = g.choice(r, keep, replace=False)
        out[i, r] = True
    return torch.from_numpy(out).squeeze()

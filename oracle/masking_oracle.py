"""CPU oracle for mask generation and the mask-index gather.  TEST INFRASTRUCTURE ONLY.

Restates the branch of fairseq-style span masking that the reference reaches
(reference wavjepa/audio_masking.py:5-194 with mask_type="static", no_overlap=False, idc_select_ver=1,
num_mask_ver=2, batch of one) and the two maskers built on it (reference wavjepa/masking.py:66-128,
:150-207).  The reference draws a *fresh* `np.random.default_rng(None)` per call (OS entropy); here the
generator factory is an argument so that `tests/golden/make_golden.py` can pin both sides to the same
seeded sequence and compare masks bit for bit.
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np

RngFactory = Callable[[], np.random.Generator]


def _default_factory() -> np.random.Generator:
    return np.random.default_rng(None)


def span_mask(n: int, prob: float, span: int, new_rng: RngFactory = _default_factory) -> np.ndarray:
    """One row of span masking: bool [n], True inside a span.

    num_spans = int(prob * n / span + U[0,1));  every span has length `span`;  starts are drawn without
    replacement from range(n - min_len) where min_len = span (or n - num_spans - 1 when n - span <= num_spans);
    spans are clipped at n.  Zero spans is an error in the reference (audio_masking.py:105-107).
    """
    rng = new_rng()
    num = int(prob * n / float(span) + rng.random())
    num = max(0, num)
    if num == 0:
        raise ValueError("this should never happens")
    min_len = span
    if n - min_len <= num:
        min_len = n - num - 1
    starts = rng.choice(n - min_len, num, replace=False)
    idx = (starts[:, None] + np.arange(span)[None, :]).reshape(-1)
    idx = np.unique(idx[idx < n])
    mask = np.zeros(n, dtype=bool)
    mask[idx] = True
    return mask


def time_inverse_block_masks(batch_size: int, n_times: int, in_channels: int = 1, *, groups: int = 4,
                             context_prob: float = 0.65, context_len: int = 10, target_prob: float = 0.25,
                             target_len: int = 10, ratio_cutoff: float = 0.1,
                             new_rng: RngFactory = _default_factory) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """AudioSet masker.  Returns (ctx_mask [B,T] True = NOT context, target [B,G,T] True = target,
    visible_mask [B,G,T] = ctx_mask XOR target, i.e. False on context U that group's targets)."""
    T = n_times // in_channels
    ctx_mask = np.zeros((batch_size, T), dtype=bool)
    target = np.zeros((batch_size, groups, T), dtype=bool)
    for b in range(batch_size):
        tg = np.zeros((groups, T), dtype=bool)
        while True:
            context = ~span_mask(T, context_prob, context_len, new_rng)
            for g in range(groups):
                tg[g] = span_mask(T, target_prob, target_len, new_rng)
            context = context & ~tg.any(axis=0)
            if context.sum() / T >= ratio_cutoff:
                break
        target[b] = tg
        ctx_mask[b] = ~context
    vis = np.logical_xor(ctx_mask[:, None, :], target)
    return ctx_mask, target, vis


def _drop_short_runs(mask: np.ndarray, min_len: int) -> np.ndarray:
    out = mask.copy()
    n = len(mask)
    i = 0
    while i < n:
        j = i
        while j < n and mask[j] == mask[i]:
            j += 1
        if mask[i] and (j - i) < min_len:
            out[i:j] = False
        i = j
    return out


def speech_masks(batch_size: int, n_times: int, in_channels: int = 1, *, groups: int = 4, target_prob: float = 0.1,
                 target_len: int = 10, ratio_cutoff: float = 0.5, min_context_len: int = 5,
                 new_rng: RngFactory = _default_factory) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """LibriSpeech masker: context = complement of all targets with True-runs shorter than min_context_len removed."""
    T = n_times // in_channels
    ctx_mask = np.zeros((batch_size, T), dtype=bool)
    target = np.zeros((batch_size, groups, T), dtype=bool)
    for b in range(batch_size):
        while True:
            tg = np.zeros((groups, T), dtype=bool)
            for g in range(groups):
                tg[g] = span_mask(T, target_prob, target_len, new_rng)
            context = _drop_short_runs(~tg.any(axis=0), min_context_len)
            if context.sum() / T >= ratio_cutoff:
                break
        target[b] = tg
        ctx_mask[b] = ~context
    vis = np.logical_xor(ctx_mask[:, None, :], target)
    return ctx_mask, target, vis


def gather_rows(x: np.ndarray, ctx_mask: np.ndarray) -> np.ndarray:
    """x [B,T,D], ctx_mask [B,T] (True = dropped) -> rows of x at ~ctx_mask in (b, t) row-major order.
    Pure copy: the HIP gather must be bit-exact against this (reference jepa.py:399)."""
    B, T, D = x.shape
    keep = np.flatnonzero(~ctx_mask.reshape(-1))
    return x.reshape(B * T, D)[keep]


def scatter_rows_fill(rows: np.ndarray, ctx_mask: np.ndarray, fill: np.ndarray) -> np.ndarray:
    """Inverse of gather_rows with `fill` [D] written at dropped positions (reference jepa.py:425-428)."""
    B, T = ctx_mask.shape
    D = rows.shape[-1]
    out = np.broadcast_to(fill.reshape(1, D), (B * T, D)).copy()
    keep = np.flatnonzero(~ctx_mask.reshape(-1))
    out[keep] = rows
    return out.reshape(B, T, D)

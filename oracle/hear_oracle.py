"""CPU oracle for the HEAR-API inference wrapper.  TEST INFRASTRUCTURE ONLY.

Restates reference hear_api/runtime.py:12-35,98-155 and hear_api/runtime_natjepa.py:90-93,139-147 (window split, per-window normalisation, key-padding
mask for padded tokens, cut-off, timestamps) on top of `jepa_oracle.audio_representation`.
"""
from __future__ import annotations

from typing import Tuple

import torch

from . import jepa_oracle as J


def window_normalize(x: torch.Tensor) -> torch.Tensor:
    mean = x.mean(dim=(-2, -1), keepdim=True)
    std = x.std(dim=(-2, -1), keepdim=True)
    return (x - mean) / (std + 1e-5)


def padding_plan(cur_frames: int, unit_frames: int, sr: int, output_steps: int, target_length: int) -> Tuple[int, int, int, int]:
    """Returns (pad_frames, n_windows_for_mask, total_output_steps, cut_off) exactly as the reference computes
    them, including its integer-seconds quirk: process_seconds is passed as target_length // sr (=2 for 2.01 s)."""
    pad_frames = unit_frames - (cur_frames % unit_frames)
    total = cur_frames + pad_frames
    process_seconds = target_length // sr
    n_chunks = int((total / sr) / process_seconds)
    total_output_steps = output_steps * n_chunks
    output_sr = int(output_steps / process_seconds)
    pad_steps = int((pad_frames / sr) * output_sr)
    return pad_frames, n_chunks, total_output_steps, total_output_steps - pad_steps


def timestamp_embeddings(P, audio: torch.Tensor, *, sr: int = 16000, process_seconds: float = 2.01,
                         spec=J.WAVJEPA_CONV_SPEC, enc_heads: int = 12, mode: str = "fp32", channel_streams: int = 1):
    """audio [B, C, n] (already loudness-normalised / channel-fixed) -> (emb [B, steps, D], ts [B, steps] ms).

    channel_streams > 1 restates reference hear_api/runtime_natjepa.py:90-93,139-147 (the multi-channel model: every channel is a
    token stream of its own, "B (C S)"): a window contributes total_patches // C steps, the window's key-padding mask is repeated
    per stream, and the embedding is the mean over the streams."""
    B = audio.shape[0]
    n_in = audio.shape[-1]
    unit = int(process_seconds * sr)
    steps = J.conv_token_count(unit, spec)          # per mono stream (= total_patches // C for the channel extractor)
    target_length = int(sr * process_seconds)
    pad_frames, _, total_steps, cut_off = padding_plan(n_in, unit, sr, steps, target_length)
    audio = torch.nn.functional.pad(audio, (0, pad_frames))
    mask = torch.zeros((B, total_steps), dtype=torch.bool)
    mask[..., cut_off:] = True
    outs = []
    for i in range(audio.shape[-1] // unit):
        win = audio[..., i * unit:(i + 1) * unit]
        m = mask[..., i * steps:(i + 1) * steps]
        if m.shape[-1] < steps:  # the reference's mask can be shorter than the window grid; pad as 'masked'
            m = torch.nn.functional.pad(m, (0, steps - m.shape[-1]), value=True)
        if channel_streams > 1:
            e = J.audio_representation(P, window_normalize(win), m.repeat(1, channel_streams), spec=spec, enc_heads=enc_heads, mode=mode)
            outs.append(e.view(B, channel_streams, steps, e.shape[-1]).mean(dim=1))
            continue
        outs.append(J.audio_representation(P, window_normalize(win), m, spec=spec, enc_heads=enc_heads, mode=mode))
    x = torch.cat(outs, dim=1)[:, :cut_off, :]
    step_ms = (n_in / sr) / x.shape[1] * 1000
    ts = torch.tensor([step_ms * i for i in range(x.shape[1])]).unsqueeze(0).repeat(B, 1)
    return x, ts

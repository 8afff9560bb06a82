"""HEAR-2021 API wrapper for the multi-channel (WavJEPA-Nat) model: same surface as reference hear_api/runtime_natjepa.py:38-155
(`RuntimeNatJEPA(...).get_timestamp_embeddings(audio)`, `get_scene_embeddings`).

What differs from `RuntimeJEPA` (reference runtime_natjepa.py:90-93,139-147): the extractor emits `in_channels` token streams per
window ("B (C S)", channel-major), so a window contributes `total_patches // in_channels` steps; the key-padding mask of those steps
is repeated for every channel stream, and the window's embedding is the mean over the channel streams.

Upstream passes `in_channels=` / `is_spectrogram=` on to `JEPA(...)` and reads `model.in_channels`; its JEPA has neither (the keyword
ends in `nn.Module.__init__`), so the channel count is taken from the extractor here -- the behaviour the code states, runnable."""
from __future__ import annotations

import torch

from .runtime import RuntimeJEPA, normalize


class RuntimeNatJEPA(RuntimeJEPA):
    def __init__(self, in_channels, weights, is_spectrogram, process_seconds, extractor, model_size, sr, **kwargs) -> None:
        if int(getattr(extractor, "in_channels", in_channels)) != int(in_channels):
            raise ValueError(f"extractor built for {extractor.in_channels} channel(s), runtime asked for {in_channels}")
        super().__init__(in_channels=in_channels, weights=weights, is_spectrogram=is_spectrogram, process_seconds=process_seconds,
                         extractor=extractor, model_size=model_size, sr=sr, **kwargs)

    def steps_per_window(self, window_tokens: int) -> int:
        if window_tokens % self.in_channels:
            raise ValueError(f"{window_tokens} tokens per window do not split into {self.in_channels} channel streams")
        return window_tokens // self.in_channels

    def window_embedding(self, window: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        C = self.in_channels
        emb = self.model.get_audio_representation(normalize(window), mask.repeat(1, C))       # "B E -> B (C E)"
        B, _, D = emb.shape
        return emb.view(B, C, self.output_steps, D).mean(dim=1)                              # "B (C S) E -> B C S E", mean over C

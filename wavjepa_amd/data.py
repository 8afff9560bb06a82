"""Synthetic input source for benchmarking / smoke training: white-noise 10 s sources generated on the device and
pre-generated CPU masks that are cycled (so the NumPy masker -- ~0.6 ms per clip -- is not on the timed path).

Stands in for reference data_modules/WebAudioDataModule.py:43-142 (webdataset shards -> resample -> RMS-normalise ->
10 s pad -> masker); the batch layout is the same: (audio [B,1,L_full] f32, ctx [B,S,T], tgt [B,S,G,T], vis [B,S,G,T])."""
from __future__ import annotations

from typing import List, Tuple

import torch



class SyntheticAudioSource:
    def __init__(self, masker, *, batch_size: int = 32, samples_per_audio: int = 8, n_tokens: int = 200, in_channels: int = 1,
                 sr: int = 16000, seconds: float = 10.0, seed: int = 42, n_mask_sets: int = 16, device=None):
        self.B, self.S, self.T, self.C = batch_size, samples_per_audio, n_tokens, in_channels
        self.L_full = int(sr * seconds)
        self.device = device
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed)
        self.mask_sets: List[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = []
        for _ in range(n_mask_sets):
            ctx, tgt, vis = masker(batch_size=batch_size * samples_per_audio, n_times=n_tokens, in_channels=in_channels)
            self.mask_sets.append((ctx.view(batch_size, samples_per_audio, -1), tgt.view(batch_size, samples_per_audio, *tgt.shape[1:]),
                                   vis.view(batch_size, samples_per_audio, *vis.shape[1:])))
        self._i = 0

    def next_batch(self):
        audio = torch.randn(self.B, self.C, self.L_full, generator=self.gen, device=self.device, dtype=torch.float32)
        ctx, tgt, vis = self.mask_sets[self._i % len(self.mask_sets)]
        self._i += 1
        return audio, ctx, tgt, vis

    def __iter__(self):
        while True:
            yield self.next_batch()


class NatSceneSource(SyntheticAudioSource):
    """WavJEPA-Nat input (BASELINE config 4): mono white-noise sources are turned into BINAURAL scenes on the device -- source RIR
    convolution, `n_noise` noise positions convolved with their own RIRs and summed, segmental-SNR mix -- before the usual crops
    (the reference does this in the `on_after_batch_transfer` hook of its denoiser stage, wavjepa/denoiser.py:217-249, through
    data_modules/scene_module/generate_scenes_batch.py; as wired there it keeps receiver channel 0 only, here both channels of the
    RIRs are used, which is what a 2-channel front-end consumes).  RIRs are synthetic: exponentially decaying Gaussian noise,
    `rir_seconds` long, drawn once.  Masks are made for `in_channels` = 2 channel streams."""

    def __init__(self, masker, *, rir_seconds: float = 0.5, n_noise: int = 2, sr: int = 16000, **kw):
        super().__init__(masker, in_channels=2, sr=sr, **kw)
        L = int(rir_seconds * sr)
        decay = torch.exp(-torch.arange(L, device=self.device) / (0.12 * L))
        self.source_rir = torch.randn(self.B, 2, L, generator=self.gen, device=self.device) * decay
        self.noise_rirs = torch.randn(self.B, n_noise, 2, L, generator=self.gen, device=self.device) * decay
        self.source_rir[:, :, 0] += 4.0                      # direct path

    def next_batch(self):
        from . import scene
        src = torch.randn(self.B, self.L_full, generator=self.gen, device=self.device)
        noise = torch.randn(self.B, self.L_full, generator=self.gen, device=self.device)
        snr = torch.rand(self.B, generator=self.gen, device=self.device) * 35.0 + 5.0
        length = torch.randint(self.L_full // 4, self.L_full, (self.B,), generator=self.gen, device=self.device)
        start = ((self.L_full - length).float() * torch.rand(self.B, generator=self.gen, device=self.device)).long()
        conv, agg = scene.process_audio(self.source_rir, self.noise_rirs, src, noise)
        audio = scene.add_noise(conv, agg, snr, start, length)                      # [B, 2, L_full]
        ctx, tgt, vis = self.mask_sets[self._i % len(self.mask_sets)]
        self._i += 1
        return audio, ctx, tgt, vis

"""Waveform pre-processing of the HEAR wrapper (reference hear_api/feature_helper.py:6-88): RMS-normalise to -14 dBFS,
fix the channel count, pad to the longest clip, move to the GPU."""
from __future__ import annotations

import torch


def normalize_audio(audio_data: torch.Tensor, target_dBFS: float = -14.0) -> torch.Tensor:
    rms = torch.sqrt(torch.mean(audio_data ** 2))
    if rms == 0:
        return audio_data
    gain_dB = target_dBFS - 20 * torch.log10(rms)
    return audio_data * (10 ** (gain_dB / 20))


class FeatureExtractor(torch.nn.Module):
    def __init__(self, in_channels: int) -> None:
        super().__init__()
        self.in_channels = in_channels

    def _fix_channels(self, audio: torch.Tensor) -> torch.Tensor:
        c = audio.shape[0]
        if c == self.in_channels:
            return audio
        if c == 1 and self.in_channels in (2, 4):
            return audio.repeat(self.in_channels, 1)
        if c == 2 and self.in_channels == 1:
            return audio.mean(dim=0, keepdim=True)
        if c == 4 and self.in_channels == 1:
            return audio[:1]
        if c == 4 and self.in_channels == 2:
            return audio[:1].repeat(2, 1)
        raise Exception("Unknowm channel count")

    def _wav2feature(self, waveforms):
        feats = []
        for audio in waveforms:
            if audio.ndim == 2 and audio.shape[0] > 100:
                audio = audio.transpose(1, 0)
            if audio.ndim == 1:
                audio = audio.unsqueeze(0)
            feats.append(self._fix_channels(normalize_audio(audio, -14.0)))
        # pad_sequence pads along dim 0, so the reference effectively requires equal lengths per call; keep its layout
        return torch.nn.utils.rnn.pad_sequence(feats, batch_first=True)

    def forward(self, x):
        x = self._wav2feature(x)
        return x.cuda() if torch.cuda.is_available() else x

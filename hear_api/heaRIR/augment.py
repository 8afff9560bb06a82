"""reference hear_api/heaRIR/augment.py: wraps a scene iterator and applies one spatial scene per clip."""
from typing import Optional

import torch

from .iterators import SceneIterator
from .scene_module import generate_scene


class Augmenter:
    def __init__(self, spatial_scene_iter: Optional[SceneIterator], sr: int, snr: Optional[int]):
        self.spatial_scene_iter = spatial_scene_iter
        self.sr = sr
        self.snr = snr

    def augment(self, audio: torch.Tensor, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """audio: normalised 1-D clip (GPU); noise: optional normalised 1-D clip.  Returns [channels, len(audio)]."""
        n_in = audio.shape[-1]
        if self.spatial_scene_iter:
            source_rir, noise_rirs, _ = next(self.spatial_scene_iter)
            source_rir = source_rir.to(audio.device)
            noise_rirs = [r.to(audio.device) for r in noise_rirs]
            if source_rir.shape[-1] > n_in:      # a RIR longer than the clip: zero-extend the clip first
                audio = torch.nn.functional.pad(audio, (0, source_rir.shape[-1] - n_in), value=0, mode="constant")
            audio = generate_scene(source_rir=source_rir, noise_rirs=[] if noise is None else noise_rirs, source=audio, noise=noise,
                                   snr=self.snr, sr=self.sr)
        if audio.ndim == 1:
            audio = torch.unsqueeze(audio, 0)
        return audio[:, :n_in]

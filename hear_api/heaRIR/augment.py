"""Evaluation-time augmentation of the HEAR wrapper: every clip is placed in the next spatial scene of an iterator (source RIR + noise
RIRs) and, when a noise clip is given, mixed with it at a fixed SNR (reference hear_api/heaRIR/augment.py:8-61: class `Augmenter`,
`augment(audio, noise=None)`).  Output: [channels, len(audio)] -- the scene's reverberant tail is cut, a mono pass-through gains a
channel axis."""
from typing import List, Optional, Tuple

import torch

from .iterators import SceneIterator
from .scene_module import generate_scene


class Augmenter:
    def __init__(self, spatial_scene_iter: Optional[SceneIterator], sr: int, snr: Optional[int]):
        self.spatial_scene_iter, self.sr, self.snr = spatial_scene_iter, sr, snr

    def _next_scene(self, device) -> Tuple[torch.Tensor, List[torch.Tensor]]:
        source_rir, noise_rirs, _ = next(self.spatial_scene_iter)
        return source_rir.to(device), [r.to(device) for r in noise_rirs]

    def augment(self, audio: torch.Tensor, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """audio: loudness-normalised 1-D clip; noise: optional loudness-normalised 1-D clip."""
        keep = audio.shape[-1]
        out = audio
        if self.spatial_scene_iter:
            rir, noise_rirs = self._next_scene(audio.device)
            shortfall = rir.shape[-1] - keep
            if shortfall > 0:                       # a RIR longer than the clip: the clip is zero-extended to the RIR's length first
                out = torch.nn.functional.pad(out, (0, shortfall))
            out = generate_scene(source_rir=rir, noise_rirs=noise_rirs if noise is not None else [], source=out, noise=noise,
                                 snr=self.snr, sr=self.sr)
        return (out if out.ndim > 1 else out[None])[:, :keep]

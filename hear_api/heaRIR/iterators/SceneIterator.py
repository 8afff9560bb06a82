"""reference hear_api/heaRIR/iterators/SceneIterator.py: endless random draws from a directory of scene descriptions.

Each `*.json` under `scenes` holds {"sampled_regions": [{"region": {"scene": {"source": {"rir": {"binaural_rir_path": ...,
"ambisonic_rir_path": ...}, "azimuth": ..., "elevation": ...}, "noise": [{"rir": {...}}, ...]}}}, ...]}; the RIR arrays are `.npy`
files looked up by BASENAME in `rir_data_dir`, shaped [channels, taps] at 32 kHz and padded / cut to 2 s."""
import glob
import json
import os
import threading
from random import randrange

import numpy as np
import torch

RIR_SR = 32000


def preprocess_rirs(reverb: torch.Tensor, sr: int) -> torch.Tensor:
    want = 2 * sr
    have = reverb.shape[1]
    if have < want:
        return torch.nn.functional.pad(reverb, (0, want - have), "constant", 0)
    return reverb[:, :want]


class SceneIterator:
    def __init__(self, rir_data_dir: str, scenes: str, with_noise: bool = True, ambisonic: bool = False):
        self.scenes = []
        for path in glob.glob(f"{scenes}/*.json"):
            with open(path) as fh:
                self.scenes.extend(json.load(fh)["sampled_regions"])
        self.max_len = len(self.scenes)
        self.rir_data_dir = rir_data_dir
        self.with_noise = with_noise
        self.ambisonic = ambisonic
        self._lock = threading.RLock()

    def __iter__(self):
        return self

    def _load(self, rir_entry: dict) -> torch.Tensor:
        key = "ambisonic_rir_path" if self.ambisonic else "binaural_rir_path"
        path = os.path.join(self.rir_data_dir, os.path.basename(rir_entry[key]))
        return preprocess_rirs(torch.tensor(np.load(path)).float(), RIR_SR)

    def __next__(self):
        with self._lock:
            scene = self.scenes[randrange(self.max_len)]["region"]["scene"]
            src = scene["source"]
            source_rir = self._load(src["rir"])
            noise_rirs = [self._load(n["rir"]) for n in scene["noise"]] if self.with_noise else []
        return source_rir, noise_rirs, [src["azimuth"], src["elevation"]]

"""reference hear_api/heaRIR/iterators/NoiseIterator.py: endless random draws of (waveform [channels, samples] float32, sr) from the
`*.wav` files of a directory (the reference reads them with torchaudio.load; here PCM wav through scipy, scaled to [-1, 1) the same
way)."""
import glob
from random import randrange

import numpy as np
import torch


def load_wav(path: str):
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    if data.dtype.kind == "i":
        data = data.astype(np.float32) / float(1 << (8 * data.dtype.itemsize - 1))
    elif data.dtype.kind == "u":                     # 8-bit PCM is unsigned
        data = (data.astype(np.float32) - 128.0) / 128.0
    data = np.asarray(data, dtype=np.float32)
    data = data[None, :] if data.ndim == 1 else data.T
    return torch.from_numpy(np.ascontiguousarray(data)), int(sr)


class NoiseIterator:
    def __init__(self, noise_dir: str):
        self.noise_files = glob.glob(f"{noise_dir}/*.wav")
        self.max_len = len(self.noise_files)

    def __iter__(self):
        self.index = randrange(self.max_len)
        return self

    def __next__(self):
        out = load_wav(self.noise_files[self.index])
        self.index = randrange(self.max_len)
        return out

"""Data module of the denoiser stage (reference data_modules/WebAudioDataModuleDenoiser.py) on the shard reader of this package.

Same constructor and batch as the reference: every sample is

    (audio [10 s at 32 kHz], source_rir [C, L] | None, noise [10 s] | None, noise_length, noise_start_idx, noise_rirs [n, C, L] | None, snr | None)

and a batch stacks them field by field (None fields stay lists of None, which is what `generate_scene`'s `x[0] is None` tests look at);
`Denoiser.on_after_batch_transfer` turns it into (generated, clean) crops on the GPU.  RIR sets and noise clips come from `.npy`
members of their own tar shards (the reference's RIRDataManager / NoiseDataManager: resampled shards + a 100-sample shuffle buffer;
here every DataLoader worker keeps its own two streams instead of two extra loader processes feeding queues).  Per sample
(reference :203-248): channel 0 -> resample to 32 kHz -> -14 dBFS, 10 s; rirs[0] is the source RIR, rirs[1:] the noise RIRs; the noise
clip is normalised to -14 dBFS, cut at random to the clip length with a 0.2 s fade-out when longer, faded in and out when shorter and
then placed at a random offset; SNR uniform in [snr_low, snr_high]."""
import io
import os
import random
import warnings
from typing import Iterator, List, Optional

import numpy as np
import torch

from .. import audio_io
from ..resample import KAISER_BEST, resample_waveform_cpu
from .dataset_functions import pre_process, pre_process_noise
from .WebAudioDataModule import _Base, expand_shards, iterate_shard, raw_samples


def fade_noise(noise: torch.Tensor, audio: torch.Tensor, sr: int) -> torch.Tensor:
    """reference data_modules/scene_module/generate_scenes.py:132-154: a noise clip longer than the audio is cut at a random position
    and faded out over 0.2 s; a shorter one is faded in and out."""
    n = int(0.2 * sr)
    noise = noise.clone()
    if noise.shape[-1] > audio.shape[-1]:
        start = torch.randint(0, noise.shape[-1] - audio.shape[-1], (1,)).item()
        noise = noise[start:start + audio.shape[-1]]
        noise[noise.shape[0] - n:] *= torch.linspace(1.0, 0.0, n)
        return noise
    noise[:n] *= torch.linspace(0.0, 1.0, n)
    noise[noise.shape[0] - n:] *= torch.linspace(1.0, 0.0, n)
    return noise


def npy_stream(shards: List[str], rng: random.Random, shuffle: int = 100) -> Iterator[torch.Tensor]:
    """Endless stream of the `.npy` members of `shards` as float tensors (shards drawn with replacement, small shuffle buffer)."""
    buf: List[torch.Tensor] = []
    while True:
        shard = shards[rng.randrange(len(shards))]
        for raw in iterate_shard(shard):
            if "npy" not in raw:
                continue
            item = torch.from_numpy(np.load(io.BytesIO(raw["npy"]))).float()
            if len(buf) < shuffle:
                buf.append(item)
                continue
            j = rng.randrange(len(buf))
            buf[j], item = item, buf[j]
            yield item
        if buf and len(buf) < shuffle:           # a corpus smaller than the buffer
            rng.shuffle(buf)
            yield from buf
            buf = []


def collate(samples):
    out = []
    for k in range(len(samples[0])):
        col = [s[k] for s in samples]
        if any(c is None for c in col):
            out.append(col)
        elif isinstance(col[0], torch.Tensor):
            out.append(torch.stack(col))
        else:
            out.append(torch.tensor(col))
    return tuple(out)


class WebAudioDataModuleDenoiser(_Base):
    sr: int = 32000
    in_channels: int = 1
    NUM_WORKERS: int = 16
    PREFETCH_FACTOR: int = 2
    SHUFFLE: int = 1000
    SHUFFLE_INITIAL: int = 100
    VERIFY_MD5_CLIPS: int = 64
    MAX_SHARD_FAILURES: int = 16
    MAX_SAMPLE_FAILURES: int = 1000

    def __init__(self, data_dir: str, rir_dir: str, noise_dir: str, batch_size: int = 32, with_noise: bool = False, with_rir: bool = False,
                 nr_samples_per_audio: int = 16, nr_time_points: int = 100, cache_size: int = 1000, snr_low: float = -5.0, snr_high: float = 5.0,
                 seed: int = 0, rank: Optional[int] = None, world_size: Optional[int] = None, **kwargs):
        super().__init__()
        self.data_dir, self.rir_dir, self.noise_dir = data_dir, rir_dir, noise_dir
        self.batch_size = batch_size
        self.nr_samples_per_audio = nr_samples_per_audio
        self.cache_size = cache_size
        self.nr_time_points = nr_time_points
        self.snr_low, self.snr_high = snr_low, snr_high
        self.with_noise, self.with_rir = with_noise, with_rir
        self.seed = seed
        self.rank = int(os.environ.get("RANK", 0)) if rank is None else rank
        self.world_size = int(os.environ.get("WORLD_SIZE", 1)) if world_size is None else world_size
        self.audio_train = None

    def _augment_sample(self, sample, rir_loader, noise_loader):
        audio, audio_sr = sample
        audio = audio[0, :] if audio.ndim > 1 else audio
        if audio_sr != self.sr:
            audio = resample_waveform_cpu(audio, audio_sr, self.sr, resampling_method="sinc_interp_kaiser", **KAISER_BEST)
        audio = pre_process(audio, self.sr).squeeze(0)
        noise = noise_rirs = snr = source_rir = None
        noise_start_idx, noise_length = 0, 0
        if self.with_rir:
            rirs = next(rir_loader)
            source_rir = rirs[0]
        if self.with_noise:
            if self.with_rir:
                noise_rirs = rirs[1:]
            noise = pre_process_noise(next(noise_loader))
            noise = fade_noise(noise, audio, self.sr)
            noise_length = noise.shape[-1]
            if audio.shape[-1] > noise.shape[-1]:
                noise_start_idx = torch.randint(0, audio.shape[-1] - noise.shape[-1], (1,)).item()
                placed = torch.zeros_like(audio)
                placed[noise_start_idx:noise_start_idx + noise.shape[-1]] = noise
                noise = placed
            snr = torch.distributions.uniform.Uniform(self.snr_low, self.snr_high).sample().item()
        return audio, source_rir, noise, noise_length, noise_start_idx, noise_rirs, snr

    def _batches(self, worker: int, n_workers: int):
        rng = random.Random(f"{self.seed}/{self.rank}/{self.world_size}/{worker}/{n_workers}")
        torch.manual_seed(rng.randrange(1 << 31))            # the per-sample draws (fade position, offset, SNR) use torch's generator
        shards = expand_shards(self.data_dir)
        rir_loader = npy_stream(expand_shards(self.rir_dir), rng) if self.with_rir else None
        noise_loader = npy_stream(expand_shards(self.noise_dir), rng) if self.with_noise else None
        # the shuffle buffer holds RAW samples (webdataset shuffles before decode / map): a prepared item is 10 s of 32 kHz fp32 audio
        # + zero-padded noise + RIR sets, 2.6 GB per worker at 1000 items; decoding, resampling and augmenting happen on the sample
        # popped from the buffer.  Any per-sample failure is reported and skipped (wds.warn_and_continue).
        batch, decoded, failed_in_a_row = [], 0, 0
        for raw in raw_samples(shards, rng, self.SHUFFLE, self.SHUFFLE_INITIAL, self.MAX_SHARD_FAILURES):
            try:
                item = self._augment_sample(audio_io.decode_flac(raw["flac"], verify_md5=decoded < self.VERIFY_MD5_CLIPS), rir_loader,
                                            noise_loader)
            except Exception as e:                           # noqa: BLE001
                warnings.warn(f"{raw.get('__key__')}: {e!r}; skipped")
                failed_in_a_row += 1                         # a corpus in which nothing decodes raises instead of warning for ever
                if failed_in_a_row >= self.MAX_SAMPLE_FAILURES:
                    raise RuntimeError(f"{failed_in_a_row} samples in a row failed to decode / prepare (no decodable .flac member in "
                                       f"{len(shards)} shard(s)?); last error: {e!r}") from e
                continue
            decoded += 1
            failed_in_a_row = 0
            batch.append(item)
            if len(batch) == self.batch_size:
                yield collate(batch)
                batch = []

    def setup(self, stage: str):
        if stage == "fit":
            if not expand_shards(self.data_dir):
                raise FileNotFoundError(f"no shards match {self.data_dir!r}")
            self.audio_train = True

    def train_dataloader(self):
        from torch.utils.data import DataLoader
        if self.audio_train is None:
            self.setup("fit")
        kw = dict(prefetch_factor=self.PREFETCH_FACTOR) if self.NUM_WORKERS > 0 else {}
        return DataLoader(_DenoiserBatches(self), batch_size=None, pin_memory=False, num_workers=self.NUM_WORKERS, **kw)


class _DenoiserBatches(torch.utils.data.IterableDataset):
    def __init__(self, dm: WebAudioDataModuleDenoiser):
        super().__init__()
        self.dm = dm

    def __iter__(self):
        info = torch.utils.data.get_worker_info()
        worker, n_workers = (info.id, info.num_workers) if info is not None else (0, 1)
        return self.dm._batches(worker, n_workers)

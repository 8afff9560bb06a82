"""Shard-based audio data module (reference data_modules/WebAudioDataModule.py) without webdataset / torchaudio / Lightning.

Same constructor, `setup("fit")` / `train_dataloader()` and batch layout as the reference:

    (audio [B, 1, 10 s * sr] fp32, context_mask [B, S, T], target_indices [B, S, G, T], ctx_and_target_masks [B, S, G, T])

Pipeline per worker (reference :86-121): shards drawn WITH replacement (`resampled=True`, split by rank and worker through the
seed) -> members grouped by key -> shuffle buffer (1000) of the RAW `{ext: bytes}` samples, as webdataset shuffles before
`.decode()` / `.map()` (a compressed 10 s clip is ~0.2 MB where the decoded, resampled, masked item is 0.64 MB and costs 15 ms
of CPU: a buffer of prepared items was ~40 GB per rank at 16 workers and delayed the first batch by 1000 decodes); it starts
yielding once it holds SHUFFLE_INITIAL = 100 samples, webdataset's `initial` -> decode the `.flac` member (native decoder,
wavjepa_amd/audio_io.py; the STREAMINFO MD5 is verified for the first VERIFY_MD5_CLIPS clips of every worker; undecodable
members and any per-sample failure are reported and skipped, as `wds.warn_and_continue` does) -> channel 0 -> kaiser-sinc
resampling to `sr` when the file's rate differs -> RMS -14 dBFS, pad / cut to 10 s -> masks from the masker -> batches of
`batch_size`.  MAX_SHARD_FAILURES unreadable shards in a row raise instead of spinning on warnings.  Several data
directories are mixed with `mixing_weights` (webdataset's RandomMix: a source is drawn with probability ~ its weight for every
batch).  Workers are the torch DataLoader's processes, as upstream.
"""
import glob
import os
import random
import re
import tarfile
import warnings
from typing import Iterator, List, Optional, Sequence, Tuple, Union

import torch

from .. import audio_io
from ..resample import KAISER_BEST, resample_waveform_cpu
from .dataset_functions import pre_process

try:                                         # pragma: no cover - depends on the environment
    import pytorch_lightning as _pl
    _Base = _pl.LightningDataModule
except Exception:                            # noqa: BLE001
    _Base = object


def expand_shards(spec: Union[str, Sequence[str]]) -> List[str]:
    """Shard list of a webdataset-style spec: brace ranges (`shard-{000..012}.tar`), globs, directories, or a list of those."""
    if not isinstance(spec, str):
        out: List[str] = []
        for s in spec:
            out += expand_shards(s)
        return out
    m = re.search(r"\{(\d+)\.\.(\d+)\}", spec)
    if m:
        lo, hi, width = int(m.group(1)), int(m.group(2)), len(m.group(1))
        out = []
        for i in range(lo, hi + 1):
            out += expand_shards(spec[:m.start()] + str(i).zfill(width) + spec[m.end():])
        return out
    if os.path.isdir(spec):
        return sorted(glob.glob(os.path.join(spec, "*.tar")))
    if any(ch in spec for ch in "*?["):
        return sorted(glob.glob(spec))
    return [spec]


def iterate_shard(path: str) -> Iterator[dict]:
    """Samples of one tar shard: consecutive members that share a key (path without extension) form one sample {ext: bytes}."""
    current_key, sample = None, {}
    with tarfile.open(path, "r:*") as tf:
        for member in tf:
            if not member.isfile():
                continue
            folder, base = os.path.split(member.name)
            stem, _, ext = base.partition(".")           # webdataset's rule: the extension starts at the FIRST dot of the file name
            key = os.path.join(folder, stem)
            if key != current_key:
                if sample:
                    yield sample
                current_key, sample = key, {"__key__": key}
            sample[ext.lower()] = tf.extractfile(member).read()
        if sample:
            yield sample


def raw_samples(shards: List[str], rng: random.Random, shuffle: int, initial: int = 100, max_shard_failures: int = 16) -> Iterator[dict]:
    """Endless stream of RAW `{ext: bytes}` samples that carry a .flac member -- shards drawn with replacement -- through
    webdataset's shuffle stage: the buffer takes two samples per sample it gives until it holds `shuffle` of them, and gives (a
    uniformly drawn one) as soon as it holds `initial`.  `max_shard_failures` shards in a row without a readable sample raise."""
    def stream() -> Iterator[dict]:
        failures = 0
        while True:
            shard = shards[rng.randrange(len(shards))]
            got = 0
            try:
                for raw in iterate_shard(shard):
                    if "flac" in raw:
                        got += 1
                        yield raw
            except (tarfile.TarError, OSError) as e:
                warnings.warn(f"{shard}: {e!r}; shard skipped")
            failures = 0 if got else failures + 1
            if failures >= max_shard_failures:
                raise RuntimeError(f"{failures} shards in a row gave no readable .flac sample (last: {shard})")

    buf: List[dict] = []
    initial = min(initial, max(shuffle, 1))
    src = stream()
    for raw in src:
        buf.append(raw)
        if len(buf) < shuffle:
            buf.append(next(src))
        if len(buf) >= initial:
            j = rng.randrange(len(buf))
            buf[j], buf[-1] = buf[-1], buf[j]
            yield buf.pop()


class WebAudioDataModule(_Base):
    TARGET_SECONDS: int = 10
    SHUFFLE: int = 1000
    SHUFFLE_INITIAL: int = 100
    VERIFY_MD5_CLIPS: int = 64
    MAX_SHARD_FAILURES: int = 16
    MAX_SAMPLE_FAILURES: int = 1000          # consecutive samples that fail to decode / prepare before the stream raises (= SHUFFLE)
    NUM_WORKERS: int = 16
    PREFETCH_FACTOR: int = 2

    def __init__(self, masker, data_dirs, mixing_weights: Optional[Sequence[float]], batch_size: int = 96, nr_samples_per_audio: int = 16,
                 nr_time_points: int = 100, cache_size: int = 1000, in_channels: int = 1, sr: int = 16000, seed: int = 0, rank: Optional[int] = None,
                 world_size: Optional[int] = None, **kwargs):
        super().__init__()
        self.data_dirs = data_dirs
        self.mixing_weights = mixing_weights
        self.batch_size = batch_size
        self.nr_samples_per_audio = nr_samples_per_audio
        self.cache_size = cache_size
        self.nr_time_points = nr_time_points
        self.masker = masker
        self.sr = sr
        self.in_channels = in_channels
        self.seed = seed
        self.rank = int(os.environ.get("RANK", 0)) if rank is None else rank
        self.world_size = int(os.environ.get("WORLD_SIZE", 1)) if world_size is None else world_size
        self.audio_train = None

    # ------------------------------------------------------------------------------------------------ per-sample work
    def _retrieve_sample(self, sample: Tuple[torch.Tensor, int]):
        """(waveform [channels, samples], sample rate) -> (audio [1, 10 s], context_mask, target_indices, ctx_and_target_masks)"""
        audio, audio_sr = sample
        audio = audio[0, :] if audio.ndim > 1 else audio
        if audio_sr != self.sr:
            audio = resample_waveform_cpu(audio, audio_sr, self.sr, resampling_method="sinc_interp_kaiser", **KAISER_BEST)
        audio = pre_process(audio, self.sr)
        ctx, tgt, vis = self.masker(batch_size=self.nr_samples_per_audio, n_times=self.nr_time_points, in_channels=self.in_channels)
        return audio, ctx, tgt, vis

    def _samples(self, shards: List[str], rng: random.Random, shuffle: int) -> Iterator[tuple]:
        """Endless stream of prepared samples: raw samples popped from the shuffle buffer, then decoded and prepared."""
        decoded, failed_in_a_row = 0, 0
        for raw in raw_samples(shards, rng, shuffle, self.SHUFFLE_INITIAL, self.MAX_SHARD_FAILURES):
            try:
                item = self._retrieve_sample(audio_io.decode_flac(raw["flac"], verify_md5=decoded < self.VERIFY_MD5_CLIPS))
            except Exception as e:                                   # noqa: BLE001  (wds.warn_and_continue: any per-sample failure)
                warnings.warn(f"{raw.get('__key__')}: {e!r}; skipped")
                # a corpus in which NOTHING decodes (wrong bit depth, unsupported streams, every clip over the size bound) must not
                # spin on warnings for ever while the trainer waits for its first batch: readable shards do not count as progress
                failed_in_a_row += 1
                if failed_in_a_row >= self.MAX_SAMPLE_FAILURES:
                    raise RuntimeError(f"{failed_in_a_row} samples in a row failed to decode / prepare (no decodable .flac member in "
                                       f"{len(shards)} shard(s)?); last error: {e!r}") from e
                continue
            decoded += 1
            failed_in_a_row = 0
            yield item

    def _batches(self, worker: int, n_workers: int) -> Iterator[tuple]:
        rng = random.Random(f"{self.seed}/{self.rank}/{self.world_size}/{worker}/{n_workers}")
        if self.mixing_weights is None:
            streams = [self._samples(expand_shards(self.data_dirs), rng, self.SHUFFLE)]
            weights = [1.0]
        else:
            dirs = [self.data_dirs] if isinstance(self.data_dirs, str) else list(self.data_dirs)
            streams = [self._samples(expand_shards(d), rng, self.SHUFFLE) for d in dirs]
            weights = list(self.mixing_weights)
        while True:
            src = rng.choices(range(len(streams)), weights=weights)[0]
            items = [next(streams[src]) for _ in range(self.batch_size)]
            yield tuple(torch.stack([it[k] for it in items]) for k in range(4))

    # ------------------------------------------------------------------------------------------------ Lightning-style surface
    def setup(self, stage: str):
        if stage == "fit":
            shards = expand_shards(self.data_dirs)
            if not shards:
                raise FileNotFoundError(f"no shards match {self.data_dirs!r}")
            self.audio_train = shards

    def train_dataloader(self):
        """As the reference (:134-142): a torch DataLoader over an iterable of ready batches, NUM_WORKERS worker PROCESSES (each with
        its own shard stream, seeded by rank and worker id), pinned memory when a GPU is present."""
        from torch.utils.data import DataLoader
        if self.audio_train is None:
            self.setup("fit")
        kw = dict(prefetch_factor=self.PREFETCH_FACTOR) if self.NUM_WORKERS > 0 else {}
        return DataLoader(_ShardBatches(self), batch_size=None, pin_memory=torch.cuda.is_available(), num_workers=self.NUM_WORKERS, **kw)


class _ShardBatches(torch.utils.data.IterableDataset):
    def __init__(self, dm: WebAudioDataModule):
        super().__init__()
        self.dm = dm

    def __iter__(self):
        info = torch.utils.data.get_worker_info()
        worker, n_workers = (info.id, info.num_workers) if info is not None else (0, 1)
        return self.dm._batches(worker, n_workers)

#!/usr/bin/env python3
"""Throughput of the real data path on the host cores (no GPU needed): one synthetic 10 s / 32 kHz / 16-bit mono FLAC clip (fixed
order-2 prediction, Rice coded: the bitstream shape libFLAC produces at low compression levels) repeated in a tar shard, then
(a) the native decoder alone, (b) decode + 32 -> 16 kHz kaiser-sinc resampling + RMS normalisation + masks per clip on one thread,
(c) the data module with N DataLoader worker processes."""
import io
import json
import os
import sys
import tarfile
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import flac_encoder as E  # noqa: E402
from wavjepa_amd import audio_io as A  # noqa: E402
from wavjepa_amd.data_modules import WebAudioDataModule  # noqa: E402
from wavjepa_amd.masking import TimeInverseBlockMasker  # noqa: E402

rng = np.random.default_rng(0)
n = 320000
t = np.arange(n)
pcm = np.round(9000 * np.sin(2 * np.pi * 330 * t / 32000) + 900 * rng.standard_normal(n)).astype(np.int64)[:, None]
t0 = time.time()
flac = E.encode(pcm, 32000, 16, blocksize=4096, subframes=dict(kind="fixed", order=2, porder=3))
enc_s = time.time() - t0
reps = 20
t0 = time.time()
for _ in range(reps):
    A.decode_flac(flac)
dec = (time.time() - t0) / reps
masker = TimeInverseBlockMasker(4, 0.65, 10, 0.25, 10, 0.1)
with tempfile.TemporaryDirectory() as d:
    with tarfile.open(os.path.join(d, "shard-000.tar"), "w") as tf:
        for i in range(64):
            ti = tarfile.TarInfo(f"clip{i:04d}.flac")
            ti.size = len(flac)
            tf.addfile(ti, io.BytesIO(flac))

    class DM(WebAudioDataModule):
        SHUFFLE = 16

    res = {}
    for workers in (1, 2, 4, 8):
        DM.NUM_WORKERS = workers
        dm = DM(masker, d, None, batch_size=8, nr_samples_per_audio=8, nr_time_points=200, sr=16000)
        it = iter(dm.train_dataloader())
        next(it)
        t0 = time.time()
        k = 24
        for _ in range(k):
            next(it)
        res[workers] = round(k * 8 / (time.time() - t0), 1)
        del it
print(json.dumps({"clip": "10 s, 32 kHz, 16-bit mono FLAC (%d bytes, %.2f bits/sample)" % (len(flac), 8 * len(flac) / n),
                  "decode_ms_per_clip": round(dec * 1e3, 2), "decode_realtime_factor": round(10.0 / dec, 0),
                  "sources_per_s_by_worker_processes": res, "host_threads": os.cpu_count(), "encoder_s_per_clip_python": round(enc_s, 1)}))

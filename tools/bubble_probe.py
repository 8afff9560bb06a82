#!/usr/bin/env python3
"""Is the main stream waiting for the HOST between the front-end and the student's first kernel?  Two events recorded on the main stream
around engine._conv_rows (host-side NumPy list building + one upload): with the host running ahead of the GPU their distance is the copy's
~0.1 ms; with the host just in time it is the host's own time in that function.  Unprofiled, steady state."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from wavjepa_amd.data import SyntheticAudioSource  # noqa: E402
from wavjepa_amd.masking import TimeInverseBlockMasker  # noqa: E402
from wavjepa_amd.trainer import StepRunner  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build_model(dev, seed=42)
model.trainer.max_steps = 375000
masker = TimeInverseBlockMasker(4, 0.65, 10, 0.25, 10, 0.1)
src = SyntheticAudioSource(masker, batch_size=32, samples_per_audio=8, n_tokens=model.total_patches, seed=42, n_mask_sets=64, device=dev)
runner = StepRunner(model, gradient_clip_val=5.0)
eng = model._ensure_engine()
evs, host = [], []
orig = eng._conv_rows


def wrapped(plan):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    r = orig(plan)
    host.append((time.perf_counter() - t0) * 1e3)
    e1.record()
    evs.append((e0, e1))
    return r


eng._conv_rows = wrapped
for i in range(30):
    runner.step(src.next_batch(), i)
torch.cuda.synchronize()
# _conv_rows is called twice per step (forward: builds and uploads; backward: cached) -- report the forward calls (the longer host times)
gaps = [a.elapsed_time(b) for a, b in evs]
pairs = sorted(zip(host, gaps), reverse=True)[:12]
print("host ms in _conv_rows / GPU-side gap on the main stream around it (12 longest host times of 30 steps):")
for h, g in pairs:
    print(f"  host {h:6.2f} ms   stream gap {g:6.2f} ms")

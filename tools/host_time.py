#!/usr/bin/env python3
"""How long does the HOST need to issue one training step (Python, NumPy index lists, ~550 ctypes launches)?  After a device
synchronisation the launches of the next steps return long before the GPU has run them, so the wall time of `runner.step` over a few steps
(each started behind a fresh synchronisation) is host time.  Compared with the GPU's step time it says how far the host is from being the bound."""
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
for i in range(6):
    runner.step(src.next_batch(), i)
torch.cuda.synchronize()
host, total = [], []
for i in range(6, 14):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    runner.step(src.next_batch(), i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3)
    total.append((t2 - t0) * 1e3)
host.sort(); total.sort()
print(f"host time to issue one step: median {host[len(host) // 2]:.1f} ms (min {host[0]:.1f}, max {host[-1]:.1f}); "
      f"issue + GPU completion of a lone step: median {total[len(total) // 2]:.1f} ms; cores: {os.cpu_count()}")
# where it goes: cProfile of three steps
import cProfile, pstats, io
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for i in range(14, 17):
    runner.step(src.next_batch(), i)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print("\n".join(s.getvalue().splitlines()[:60]))

#!/usr/bin/env python3
"""Scene-augmentation micro-benchmark at the denoiser stage's sizes (GPU box): 32 sources of 10 s at 32 kHz, one source RIR and two
noise RIRs of 1.5 s, segmental-SNR mix.  HIP-event timing (median of 7), HBM roofline of the dominant kernel against the
ALGORITHMIC bytes (every input read once, every output written once), and the oracle (numpy, one core) on a bounded sample."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavjepa_amd import scene  # noqa: E402

dev = torch.device("cuda:0")
B, T, L, NN = 32, 320000, 48000, 2


def timeit(fn, n=7):
    ts = []
    for r in range(n + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


g = torch.Generator().manual_seed(0)
src = torch.randn(B, T, generator=g).to(dev)
noise = torch.randn(B, T, generator=g).to(dev)
srir = (torch.randn(B, 2, L, generator=g) * torch.exp(-torch.arange(L) / 6000.0)).to(dev)
nrir = (torch.randn(B, NN, 2, L, generator=g) * torch.exp(-torch.arange(L) / 9000.0)).to(dev)
length = torch.randint(T // 4, T, (B,), generator=g).to(dev)
start = torch.zeros(B, dtype=torch.int64, device=dev)
snr = (torch.rand(B, generator=g) * 20 - 5).to(dev)

t_conv = timeit(lambda: scene.convolve_with_rir(src, srir[:, :1]))
t_scene = timeit(lambda: scene.generate_scene(srir, nrir, src, noise, length, start, snr))
conv_bytes = B * T * 4 * 2 + B * L * 4
res = {"workload": f"{B} sources x {T} samples, RIR {L} taps, {NN} noise positions", "rir_convolve_ms": round(t_conv, 4),
       "rir_convolve_algorithmic_GBps": round(conv_bytes / t_conv / 1e6, 1), "rir_convolve_frac_of_8TBps": round(conv_bytes / t_conv / 1e6 / 8000, 4),
       "generate_scene_ms": round(t_scene, 4), "sources_per_s": round(B / t_scene * 1e3, 1)}
if "--cpu" in sys.argv:
    from oracle import scene_oracle as S
    n = 2
    t0 = time.time()
    S.generate_scene(srir[:n].cpu().numpy(), nrir[:n].cpu().numpy(), src[:n].cpu().numpy(), noise[:n].cpu().numpy(),
                     length[:n].cpu().numpy(), start[:n].cpu().numpy(), snr[:n].cpu().numpy(), np.float32)
    dt = time.time() - t0
    res["cpu_oracle_sources_per_s"] = round(n / dt, 2)
    res["cpu_oracle_sample"] = f"{n} sources, numpy rfft of length T + L - 1, 1 core"
print(json.dumps(res))

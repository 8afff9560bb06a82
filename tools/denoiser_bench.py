#!/usr/bin/env python3
"""Denoiser stage at full size (GPU box): WavJEPA-base student + frozen WavJEPA-base teacher, 16 sources of 10 s at 32 kHz x 8 crops
= 128 clean + 128 generated clips of 2.01 s per step; times the batch hook (scene + resampling + crops) and the training step
(teacher inference, student forward on 256 clips, backward, AdamW) with HIP events, median of 7."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavjepa_amd.denoiser import Denoiser  # noqa: E402
from wavjepa_amd.extractors import ConvFeatureExtractor  # noqa: E402
from wavjepa_amd.jepa import JEPA  # noqa: E402
from wavjepa_amd.types import TransformerEncoderCFG, TransformerLayerCFG  # noqa: E402

dev = torch.device("cuda:0")
SPEC = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)]
B, S, T32, L, NN = 16, 8, 320000, 48000, 2


def timeit(fn, n=7):
    ts = []
    for r in range(n + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if r > 1:
            ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


torch.manual_seed(0)
den = Denoiser(ConvFeatureExtractor(conv_layers_spec=SPEC, in_channels=1), TransformerLayerCFG.create(), TransformerEncoderCFG.create(),
               nr_samples_per_audio=S, alpha=0.0).to(dev)
tea = JEPA(feature_extractor=ConvFeatureExtractor(conv_layers_spec=SPEC, in_channels=1), transformer_encoder_cfg=TransformerEncoderCFG.create(),
           transformer_encoder_layers_cfg=TransformerLayerCFG.create(), transformer_decoder_cfg=TransformerEncoderCFG.create(),
           transformer_decoder_layers_cfg=TransformerLayerCFG.create(d_model=384), process_audio_seconds=2.01).to(dev)
den._set_teacher(tea)
opt = den.configure_optimizers()["optimizer"]
opt.max_grad_norm = 5.0
g = torch.Generator().manual_seed(1)
batch = (torch.randn(B, T32, generator=g), torch.randn(B, 2, L, generator=g) * torch.exp(-torch.arange(L) / 6000.0),
         torch.randn(B, T32, generator=g), torch.randint(T32 // 4, T32, (B,), generator=g), torch.zeros(B, dtype=torch.long),
         torch.randn(B, NN, 2, L, generator=g) * torch.exp(-torch.arange(L) / 9000.0), torch.rand(B, generator=g) * 20 - 5)
batch = tuple(t.to(dev) for t in batch)
clips = {}


def hook():
    clips["b"] = den.on_after_batch_transfer(batch, 0)


def step():
    out = den.training_step(clips["b"], 0)
    out["loss"].backward()
    opt.step()
    clips["loss"] = out["loss"].detach()


t_hook = timeit(hook)
t_step = timeit(step)
print(json.dumps({"workload": f"denoiser stage, WavJEPA-base student + frozen WavJEPA-base teacher, {B} sources x {S} crops = {B * S} clean + "
                              f"{B * S} generated clips of 2.01 s (scene: {L}-tap source RIR + {NN} noise RIRs at 32 kHz, kaiser-sinc to 16 kHz)",
                  "batch_hook_ms": round(t_hook, 3), "train_step_ms": round(t_step, 2), "clip_pairs_per_s": round(B * S / (t_hook + t_step) * 1e3, 1),
                  "final_loss": round(float(clips["loss"]), 5), "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}))

"""Per-clip scene generation for evaluation (reference hear_api/heaRIR/scene_module/generate_scenes.py) over the HIP scene kernels.

Same function names and argument meaning as the reference: a 1-D `source`, a source RIR [C, L] (or [L]), a list of noise RIRs, a
1-D noise clip, a scalar SNR.  The convolutions run in `wj_rir_convolve` (one clip = a batch of one), the mix in `wj_snr_mix` over
the whole clip (torchaudio `add_noise` semantics: energies over all valid samples).  GPU tensors only.
"""
from pathlib import Path

import numpy as np
import torch

from wavjepa_amd import ops
from wavjepa_amd import scene as _scene


def _ramp(n: int, up: bool, device) -> torch.Tensor:
    r = torch.linspace(0.0, 1.0, n, device=device)
    return r if up else r.flip(0)


def apply_fadein(audio: torch.Tensor, sr: int, duration: float = 0.10) -> torch.Tensor:
    """reference :11-20 -- linear fade over the first `duration` seconds, in place."""
    n = int(duration * sr)
    audio[:n] = audio[:n] * _ramp(n, True, audio.device)
    return audio


def apply_fadeout(audio: torch.Tensor, sr: int, duration: float = 0.10) -> torch.Tensor:
    """reference :23-33 -- linear fade over the last `duration` seconds, in place."""
    n = int(duration * sr)
    end = audio.shape[0]
    audio[end - n:end] = audio[end - n:end] * _ramp(n, False, audio.device)
    return audio


def load_rir(path: str):
    """reference :36-42"""
    assert Path(path).exists(), f"Path {path} does not exist"
    try:
        return torch.tensor(np.load(path))
    except Exception as e:                       # noqa: BLE001
        print(f"Error loading RIR file: {e}")


def convolve_with_rir(waveform: torch.Tensor, rir: torch.Tensor) -> torch.Tensor:
    """reference :45-60 -- 1-D waveform, rir [C, L] or [L] -> [C, T] (full convolution cut to the input length)."""
    assert waveform.ndim == 1, "No Stero sounds are accepted, cast the sound to mono or collables the first dimension!"
    if rir.ndim == 1:
        rir = rir.unsqueeze(0)
    return _scene._conv(waveform.unsqueeze(0), rir.unsqueeze(0).to(waveform.device))[0]


def add_noise(waveform: torch.Tensor, noise: torch.Tensor, snr: torch.Tensor, lengths=None) -> torch.Tensor:
    """reference :63-138 (torchaudio.functional.add_noise): y = x + a n with a from the energies of the first `lengths` samples (all
    when None).  waveform / noise [..., L], snr [...], lengths [...] or None."""
    if not (waveform.ndim - 1 == noise.ndim - 1 == snr.ndim and (lengths is None or lengths.ndim == snr.ndim)):
        raise ValueError("Input leading dimensions don't match.")
    L = waveform.size(-1)
    if noise.size(-1) != L:
        raise ValueError(f"Length dimensions of waveform and noise don't match (got {L} and {noise.size(-1)}).")
    ops.require_gpu()
    lead = waveform.shape[:-1]
    B = int(np.prod(lead)) if len(lead) else 1
    x = _scene._f32_cuda(waveform, "waveform").reshape(B, 1, L)
    n = _scene._f32_cuda(noise, "noise").reshape(B, 1, L)
    dev = x.device
    length = torch.full((B,), L, dtype=torch.int32, device=dev) if lengths is None else lengths.reshape(B).to(dev, torch.int32)
    start = torch.zeros(B, dtype=torch.int32, device=dev)
    out = torch.empty_like(x)
    ws = torch.empty(ops.workspace_bytes("wj_snr_mix", B=B, C=1, T=L) // 4, device=dev, dtype=torch.float32)
    ops.snr_mix(x, n, out, snr.reshape(B).to(dev, torch.float32).contiguous(), start, length.contiguous(), ws, B=B, C=1, T=L)
    return out.reshape(waveform.shape)


def fade_noise(noise_source: torch.Tensor, audio_source: torch.Tensor, sr: int) -> torch.Tensor:
    """reference :141-152 -- noise longer than the clip: cut + fade out; otherwise fade in and out (0.2 s)."""
    if noise_source.shape[-1] > audio_source.shape[-1]:
        noise_source = noise_source[: audio_source.shape[-1]]
        return apply_fadeout(noise_source, sr=sr, duration=0.2)
    noise_source = apply_fadein(noise_source, sr=sr, duration=0.2)
    return apply_fadeout(noise_source, sr=sr, duration=0.2)


def aggregate_noise(noise_rirs, noise_source: torch.Tensor) -> torch.Tensor:
    """reference :155-167 -- sum over the noise positions of convolve_with_rir(noise, rir_i)  ([C, T])."""
    agg = None
    for rir in noise_rirs:
        if rir.ndim == 1:
            rir = rir.unsqueeze(0)
        agg = _scene._conv(noise_source.unsqueeze(0), rir.unsqueeze(0).to(noise_source.device), out=agg, accumulate=agg is not None)
    return agg[0]


def process_audio(source_rir, noise_rirs, audio_source, noise_source, sr):
    """reference :170-191 -- a noise clip shorter than the source lands at a random offset (np.random.randint, as upstream)."""
    assert source_rir is not None, "No source RIR is provided"
    assert len(noise_rirs) > 0, "No noise RIRs are provided"
    input_length = audio_source.shape[-1]
    noise_source = fade_noise(noise_source, audio_source, sr)
    convolved_source = convolve_with_rir(audio_source, source_rir)
    agg_noise = aggregate_noise(noise_rirs, noise_source)[:, :input_length]
    if convolved_source.shape[1] > agg_noise.shape[1]:
        start = np.random.randint(0, input_length - agg_noise.shape[1])
        placed = torch.zeros_like(convolved_source)
        placed[:, start:start + agg_noise.shape[1]] = agg_noise
        return convolved_source, placed
    return convolved_source, agg_noise


def generate_scene(source_rir, noise_rirs, source, noise, snr, sr):
    """reference :194-203"""
    if len(noise_rirs) > 0:
        source, noise = process_audio(source_rir, noise_rirs, audio_source=source, noise_source=noise, sr=sr)
        snr = torch.tensor([snr], device=source.device, dtype=torch.float32).expand(source.shape[0])
        return add_noise(source, noise, snr)
    return convolve_with_rir(source, source_rir)

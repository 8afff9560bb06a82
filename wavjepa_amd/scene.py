"""Scene augmentation on the GPU: RIR convolution + segmental-SNR noise mixing.

Host-side mirror of the reference's `data_modules/scene_module/generate_scenes_batch.py` -- same function names, argument
meaning, shapes and the four cases of `generate_scene` (:152-188) -- over the HIP entry points `wj_rir_convolve` / `wj_snr_mix`
(wavjepa_amd/csrc/scene.hip).  GPU only: CPU tensors raise (no fallback); all arithmetic is fp32 as in the reference.
"""
from typing import Optional

import torch

from . import ops

FFT_SIZE = 0   # 0 = library default (8192); tests set 1024 to cross many blocks / partitions at small sizes


def _f32_cuda(t: torch.Tensor, what: str) -> torch.Tensor:
    if not t.is_cuda:
        raise ops._abi.WavJepaHipError(f"scene augmentation runs on the GPU only: {what} is on {t.device}")
    return t.float().contiguous()


def _conv(x: torch.Tensor, h: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """x [B, T], h [B, C, L] (any strides with a contiguous last dim) -> [B, C, T]."""
    ops.require_gpu()
    x = _f32_cuda(x, "waveform")
    if not h.is_cuda:
        raise ops._abi.WavJepaHipError(f"scene augmentation runs on the GPU only: rir is on {h.device}")
    if h.dtype != torch.float32 or h.stride(-1) != 1:
        h = h.float().contiguous()
    B, T = x.shape
    _, C, L = h.shape
    y = out if out is not None else torch.empty(B, C, T, device=x.device, dtype=torch.float32)
    dims = dict(B=B, C=C, T=T, L=L, fft_size=FFT_SIZE)
    ws = torch.empty(ops.workspace_bytes("wj_rir_convolve", **dims), device=x.device, dtype=torch.uint8)
    ops.rir_convolve(x, h, y, ws, h_stride_b=h.stride(0), h_stride_c=h.stride(1), accumulate=accumulate, **dims)
    return y


def convolve_with_rir(waveform: torch.Tensor, rir: torch.Tensor) -> torch.Tensor:
    """reference :12-44 -- waveform [B, T], rir [B, C, L] -> [B, C, T]: full convolution per channel, cut to the input length."""
    assert waveform.shape[0] == rir.shape[0], "Not compatible for this operation"
    return _conv(waveform, rir)


def aggregate_noise(noise_rirs: torch.Tensor, noise_source: torch.Tensor) -> torch.Tensor:
    """reference :47-71 -- noise_rirs [B, n, C, L], noise_source [B, T] -> sum over the n noise positions, [B, C, T]."""
    B, n = noise_rirs.shape[:2]
    agg = None
    for i in range(n):
        agg = _conv(noise_source, noise_rirs[:, i], out=agg, accumulate=i > 0)
    if agg is None:
        agg = torch.zeros(B, noise_rirs.shape[2], noise_source.shape[-1], device=noise_source.device)
    return agg


def process_audio(source_rir, noise_rirs, audio_source, noise_source):
    """reference :74-106."""
    assert source_rir is not None, "No source RIR is provided"
    assert len(noise_rirs) > 0, "No noise RIRs are provided"
    input_length = audio_source.shape[-1]
    convolved_source = convolve_with_rir(audio_source, source_rir)
    agg_noise = aggregate_noise(noise_rirs, noise_source)[:, :, :input_length]
    return convolved_source, agg_noise


def add_noise(source: torch.Tensor, noise: torch.Tensor, snr, start_idx, real_noise_length) -> torch.Tensor:
    """reference :108-150 -- source / noise [B, C, T]; snr [B] (or [B, 1]) tensor or float; start / length [B] tensors or ints."""
    ops.require_gpu()
    B, C, T = source.shape
    dev = source.device
    source = _f32_cuda(source, "source")
    noise = _f32_cuda(noise, "noise")
    if not isinstance(start_idx, torch.Tensor):
        start_idx = torch.tensor([start_idx] * B)
    if not isinstance(real_noise_length, torch.Tensor):
        real_noise_length = torch.tensor([real_noise_length] * B)
    if not isinstance(snr, torch.Tensor):
        snr = torch.full((B,), float(snr))
    start = start_idx.to(device=dev, dtype=torch.int32).contiguous()
    length = real_noise_length.to(device=dev, dtype=torch.int32).contiguous()
    snr_t = snr.to(device=dev, dtype=torch.float32).reshape(B).contiguous()
    out = torch.empty_like(source)
    ws = torch.empty(ops.workspace_bytes("wj_snr_mix", B=B, C=C, T=T) // 4, device=dev, dtype=torch.float32)
    ops.snr_mix(source, noise, out, snr_t, start, length, ws, B=B, C=C, T=T)
    return out


def generate_scene(source_rir, noise_rirs, source, noise, real_noise_length, noise_start_idx, snr):
    """reference :152-188 -- the four cases; output [B, 1, T] (first receiver channel) whenever a RIR or noise is applied."""
    if source_rir[0] is not None and noise[0] is not None:
        convolved_source, agg_noise = process_audio(source_rir[:, [0], :], noise_rirs[:, :, [0], :], audio_source=source,
                                                    noise_source=noise)
        return add_noise(convolved_source, agg_noise, snr, noise_start_idx, real_noise_length)
    elif source_rir[0] is not None and noise[0] is None:
        return convolve_with_rir(source, source_rir[:, [0], :])
    elif source_rir[0] is None and noise[0] is not None:
        return add_noise(source, noise, snr, noise_start_idx, real_noise_length)
    else:
        return source

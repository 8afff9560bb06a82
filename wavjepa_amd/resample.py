"""Kaiser-windowed sinc resampling on the GPU (the reference's `resample`, wavjepa/denoiser.py:29-42 and
data_modules/WebAudioDataModule.py:50-60: torchaudio.functional.resample(audio, 32000, 16000, lowpass_filter_width=64,
rolloff=0.9475937167399596, resampling_method="sinc_interp_kaiser", beta=14.769656459379492)).

torchaudio is a third-party dependency of the reference that is neither vendored there nor installed here; its published
algorithm has two steps, restated below: (1) a [new/gcd][2 * width + orig/gcd] table of windowed-sinc taps
(`_get_sinc_resample_kernel`); (2) a strided convolution of the zero-padded signal with that table, interleaving the phases and
cutting to ceil(new * L / orig) samples (`_apply_sinc_resample_kernel`) -- the HIP kernel `wj_resample_fir`.  GPU only: CPU tensors
raise.

In which precision the table is evaluated: torchaudio's `resample` hands the WAVEFORM's dtype to `_get_sinc_resample_kernel`, and both
reference call sites pass float32 waveforms (WebAudioDataModule.py:50-58 decodes to float32; denoiser.py:33-41 resamples the float32
batch), so there the tap times, the Kaiser window (torch.i0) and the sinc are all computed in float32 -- NOT in float64 and rounded
once (torchaudio only does that when no dtype is given).  `sinc_resample_kernel(..., dtype=torch.float32)` follows that op sequence
with torch's own float32 kernels and is what float32 waveforms get here; `dtype=None` is torchaudio's dtype-less form (float64
evaluation, one rounding).  The two tables differ by <= 1e-6 of the largest tap (tests/test_oracle_golden.py pins both); bit equality
with torchaudio's output cannot be claimed for either: its float32 table depends on the device's sin / i0 kernels (CPU loader workers vs
GPU batches), and torchaudio is not available to record.
"""
import math
from functools import lru_cache

import numpy as np
import torch

from . import ops

KAISER_BEST = dict(lowpass_filter_width=64, rolloff=0.9475937167399596, beta=14.769656459379492)


def _kernel_in_dtype(orig: int, new: int, lowpass_filter_width: int, rolloff: float, resampling_method: str, beta, dtype) -> np.ndarray:
    """The table evaluated in `dtype` with torch's CPU kernels, in the op order torchaudio publishes for a waveform of that dtype."""
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=dtype)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=dtype)[:, None, None] / new + idx
    t *= base_freq
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    if resampling_method == "sinc_interp_hann":
        window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    elif resampling_method == "sinc_interp_kaiser":
        beta_tensor = torch.tensor(14.769656459379492 if beta is None else float(beta))
        window = torch.i0(beta_tensor * torch.sqrt(1 - (t / lowpass_filter_width) ** 2)) / torch.i0(beta_tensor)
    else:
        raise ValueError(f"Invalid resampling method: {resampling_method}")
    t *= math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0).to(t), t.sin() / t)
    kernels *= window * scale
    return kernels[:, 0, :].to(torch.float32).numpy()


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99,
                         resampling_method: str = "sinc_interp_hann", beta=None, dtype=None):
    """-> (kernel float32 [new/gcd][2 * width + orig/gcd], width, orig/gcd, new/gcd).
    dtype=None: evaluated in float64, rounded once (torchaudio without a dtype); torch.float32: evaluated in float32, as torchaudio
    does for a float32 waveform (the reference's call sites)."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    if dtype is not None and dtype != torch.float64:
        return _kernel_in_dtype(orig, new, lowpass_filter_width, rolloff, resampling_method, beta, dtype), width, orig, new
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t = np.clip(t * base_freq, -lowpass_filter_width, lowpass_filter_width)
    if resampling_method == "sinc_interp_hann":
        window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    elif resampling_method == "sinc_interp_kaiser":
        b = 14.769656459379492 if beta is None else float(beta)
        window = np.i0(b * np.sqrt(1 - (t / lowpass_filter_width) ** 2)) / np.i0(b)
    else:
        raise ValueError(f"Invalid resampling method: {resampling_method}")
    t = t * math.pi
    scale = base_freq / orig
    with np.errstate(invalid="ignore", divide="ignore"):
        kern = np.where(t == 0, 1.0, np.sin(t) / t)
    kern = kern * window * scale
    return kern.astype(np.float32), width, orig, new


@lru_cache(maxsize=8)
def _kernel_on(device_index: int, orig_freq: int, new_freq: int, width_param: int, rolloff: float, method: str, beta):
    kern, width, orig, new = sinc_resample_kernel(orig_freq, new_freq, width_param, rolloff, method, beta, dtype=torch.float32)
    return torch.from_numpy(kern).to(torch.device("cuda", device_index)), width, orig, new


def resample_waveform(waveform: torch.Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99,
                      resampling_method: str = "sinc_interp_hann", beta=None) -> torch.Tensor:
    """torchaudio.functional.resample semantics for [..., time] float tensors on the GPU."""
    ops.require_gpu()
    if not waveform.is_cuda:
        raise ops._abi.WavJepaHipError(f"resampling runs on the GPU only: the waveform is on {waveform.device}")
    if orig_freq == new_freq:
        return waveform
    kern, width, orig, new = _kernel_on(waveform.device.index or 0, int(orig_freq), int(new_freq), int(lowpass_filter_width), float(rolloff),
                                        resampling_method, beta)
    shape = waveform.shape
    x = waveform.reshape(-1, shape[-1]).float().contiguous()
    B, L = x.shape
    L_out = int(math.ceil(new * L / orig))
    y = torch.empty(B, L_out, device=x.device, dtype=torch.float32)
    ops.resample_fir(x, kern, y, B=B, L_in=L, L_out=L_out, orig=orig, nw=new, width=width)
    return y.view(shape[:-1] + (L_out,))


def resample(audio: torch.Tensor, resample_sr: int, original_sr: int = 32000) -> torch.Tensor:
    """reference wavjepa/denoiser.py:29-42 ("kaiser best")."""
    return resample_waveform(audio, original_sr, resample_sr, resampling_method="sinc_interp_kaiser", **KAISER_BEST)


def resample_waveform_cpu(waveform: torch.Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99,
                          resampling_method: str = "sinc_interp_hann", beta=None) -> torch.Tensor:
    """The same algorithm on a CPU tensor, for the loader workers (one clip at a time, any rate pair; the reference resamples there
    too: WebAudioDataModule.py:50-60).  Strided windows x table, float32."""
    if orig_freq == new_freq:
        return waveform
    kern, width, orig, new = sinc_resample_kernel(int(orig_freq), int(new_freq), int(lowpass_filter_width), float(rolloff),
                                                  resampling_method, beta, dtype=torch.float32)
    shape = waveform.shape
    x = waveform.reshape(-1, shape[-1]).to(torch.float32).numpy()
    L = shape[-1]
    pad = np.pad(x, ((0, 0), (width, width + orig)))
    if new * orig <= 8:
        # few phases and a short stride (32 -> 16 kHz: one phase, stride 2): overlap-add FFT correlation of the whole clip per phase,
        # then the stride -- 7x faster than gathering 274-tap windows for a matrix product
        from scipy.signal import oaconvolve
        out = np.stack([oaconvolve(pad, kern[p][None, ::-1], mode="valid", axes=1)[:, ::orig] for p in range(new)], axis=-1)
        out = out.reshape(x.shape[0], -1).astype(np.float32)
    else:
        win = np.lib.stride_tricks.sliding_window_view(pad, kern.shape[1], axis=1)[:, ::orig]  # [B, frames, taps] (a view)
        out = np.matmul(win, kern.T).reshape(x.shape[0], -1)
    target = int(math.ceil(new * L / orig))
    return torch.from_numpy(np.ascontiguousarray(out[:, :target])).view(shape[:-1] + (target,))

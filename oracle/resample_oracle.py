"""CPU oracle of the reference's `resample` (wavjepa/denoiser.py:29-42 = torchaudio.functional.resample with the "kaiser best"
parameters).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED against torchaudio itself: torchaudio (a pip dependency of the reference, version not pinned there -- its
requirements list just `torchaudio`) is neither vendored in /root/reference nor installed here, so this file restates the
algorithm torchaudio publishes (`_get_sinc_resample_kernel` + `_apply_sinc_resample_kernel`, torchaudio/functional/functional.py)
in plain numpy loops, and is anchored on the reference's call sites (parameters, 32 kHz -> 16 kHz) and on known answers that follow
from the definition (tests/test_oracle_golden.py::test_resample_oracle_known_answers: unit DC gain, band-limited sinusoids keep
amplitude and phase, output length ceil(new * L / orig), energy above the new Nyquist rejected).
"""
import math

import numpy as np


def kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99, method="sinc_interp_hann", beta=None):
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    taps = 2 * width + orig
    k = np.zeros((new, taps))
    for p in range(new):
        for j in range(taps):
            t = (-p / new + (j - width) / orig) * base
            t = min(max(t, -lowpass_filter_width), lowpass_filter_width)
            if method == "sinc_interp_hann":
                win = math.cos(t * math.pi / lowpass_filter_width / 2) ** 2
            else:
                b = 14.769656459379492 if beta is None else beta
                win = float(np.i0(b * math.sqrt(1 - (t / lowpass_filter_width) ** 2)) / np.i0(b))
            tp = t * math.pi
            k[p, j] = (1.0 if tp == 0 else math.sin(tp) / tp) * win * base / orig
    return k, width, orig, new


def resample(x: np.ndarray, orig_freq: int, new_freq: int, lowpass_filter_width=64, rolloff=0.9475937167399596,
             method="sinc_interp_kaiser", beta=14.769656459379492) -> np.ndarray:
    """x [..., L] -> [..., ceil(new * L / orig)]  (float64)"""
    if orig_freq == new_freq:
        return x.astype(np.float64)
    k, width, orig, new = kernel(orig_freq, new_freq, lowpass_filter_width, rolloff, method, beta)
    shape = x.shape
    x2 = x.reshape(-1, shape[-1]).astype(np.float64)
    L = shape[-1]
    pad = np.concatenate([np.zeros((x2.shape[0], width)), x2, np.zeros((x2.shape[0], width + orig))], axis=1)
    taps = k.shape[1]
    frames = (pad.shape[1] - taps) // orig + 1
    win = np.lib.stride_tricks.sliding_window_view(pad, taps, axis=1)[:, ::orig][:, :frames]     # [B, frames, taps]
    out = np.einsum("bft,pt->bfp", win, k).reshape(x2.shape[0], -1)
    target = int(math.ceil(new * L / orig))
    return out[:, :target].reshape(shape[:-1] + (target,))

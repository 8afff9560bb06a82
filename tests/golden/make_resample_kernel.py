#!/usr/bin/env python3
"""Writes tests/golden/resample_kernel.npz: windowed-sinc tap tables of torchaudio.functional.resample's closed form
(`_get_sinc_resample_kernel`: taps[p][j] = sinc(pi t) * w(t) * base/orig with t = clamp((-p/new + (j - width)/orig) * base,
+-lowpass_filter_width), base = min(orig, new) * rolloff, width = ceil(lowpass_filter_width * orig / base), w = Kaiser
I0(beta sqrt(1 - (t / lpw)^2)) / I0(beta) or Hann cos^2) for the reference's call sites (WebAudioDataModule.py:50-60 and
wavjepa/denoiser.py:29-42: lowpass_filter_width 64, rolloff 0.9475937167399596, beta 14.769656459379492).

torchaudio itself is neither vendored in the reference nor installed in this image (and there is no network), so its own output
could not be recorded.  What this script pins instead is the closed form, evaluated along a path that shares no code with
wavjepa_amd/resample.py or oracle/resample_oracle.py: tap times as exact rationals (fractions.Fraction), the Bessel function from
scipy.special.i0 (Cephes; the product and the oracle use numpy.i0), float64 throughout.  The product's float32 table and the
oracle's float64 table are held to these numbers by tests/test_oracle_golden.py::test_resample_tap_tables_match_the_pinned_closed_form.
"""
import math
import os
from fractions import Fraction

import numpy as np
from scipy.special import i0

LPW, ROLLOFF, BETA = 64, 0.9475937167399596, 14.769656459379492
PAIRS = [(32000, 16000, "kaiser"), (48000, 16000, "kaiser"), (24000, 16000, "kaiser"), (16000, 32000, "kaiser"), (44100, 22050, "hann")]


def table(orig_freq: int, new_freq: int, method: str) -> np.ndarray:
    g = math.gcd(orig_freq, new_freq)
    orig, new = orig_freq // g, new_freq // g
    lpw = LPW if method == "kaiser" else 6
    rolloff = Fraction(ROLLOFF) if method == "kaiser" else Fraction(0.99)
    base = min(orig, new) * rolloff                                  # exact rational
    width = math.ceil(lpw * orig / base)
    out = np.zeros((new, 2 * width + orig))
    for p in range(new):
        for j in range(2 * width + orig):
            t = (Fraction(-p, new) + Fraction(j - width, orig)) * base
            t = min(max(t, -lpw), lpw)
            tf = float(t)
            if method == "kaiser":
                w = float(i0(BETA * math.sqrt(max(0.0, 1.0 - (tf / lpw) ** 2))) / i0(BETA))
            else:
                w = math.cos(tf * math.pi / lpw / 2) ** 2
            s = 1.0 if t == 0 else math.sin(math.pi * tf) / (math.pi * tf)
            out[p, j] = s * w * float(base) / orig
    return out


def table_f32(orig_freq: int, new_freq: int, method: str) -> np.ndarray:
    """The same closed form with every intermediate rounded to float32, in the order torchaudio evaluates it for a float32 waveform
    (tap index / orig + phase / new, times base, clamp, window, times pi, sin(t) / t, times window * scale): numpy float32 scalars
    and scipy's Bessel function -- no torch, no code shared with the product.  The product evaluates the same sequence with torch's
    float32 kernels; the two may differ in the last place of sin / i0 (the test allows 4 float32 ulps of the largest tap)."""
    f = np.float32
    g = math.gcd(orig_freq, new_freq)
    orig, new = orig_freq // g, new_freq // g
    lpw = LPW if method == "kaiser" else 6
    base = min(orig, new) * (ROLLOFF if method == "kaiser" else 0.99)          # a Python float, as in torchaudio
    width = math.ceil(lpw * orig / base)
    out = np.zeros((new, 2 * width + orig), dtype=np.float32)
    i0b = f(i0(f(BETA)))
    for p in range(new):
        for j in range(2 * width + orig):
            t = f(f(f(-p) / f(new)) + f(f(j - width) / f(orig)))
            t = f(t * f(base))
            t = f(min(max(t, f(-lpw)), f(lpw)))
            if method == "kaiser":
                r = f(t / f(lpw))
                w = f(f(i0(f(f(BETA) * f(np.sqrt(f(f(1) - f(r * r))))))) / i0b)
            else:
                c = f(np.cos(f(f(f(t * f(math.pi)) / f(lpw)) / f(2))))
                w = f(c * c)
            tp = f(t * f(math.pi))
            s = f(1) if tp == 0 else f(f(np.sin(tp)) / tp)
            out[p, j] = f(s * f(w * f(base / orig)))
    return out


def main() -> None:
    out = {}
    for o, n, m in PAIRS:
        k = table(o, n, m)
        out[f"{o}_{n}_{m}"] = k
        k32 = table_f32(o, n, m)
        out[f"f32:{o}_{n}_{m}"] = k32
        assert np.abs(k32 - k).max() < 2e-6 * np.abs(k).max() + 2e-7, (o, n, np.abs(k32 - k).max())   # float32 evaluation vs the exact form
        # sanity that follows from the definition: every output phase has (nearly) unit DC gain when the pass band covers DC
        assert abs(k.sum(axis=1) - 1.0).max() < 2e-3, (o, n, k.sum(axis=1))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resample_kernel.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()

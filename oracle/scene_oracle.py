"""CPU oracle for the scene augmentation.  TEST INFRASTRUCTURE ONLY -- never on the product path.

Restates, in numpy, the algorithm of the reference's `data_modules/scene_module/generate_scenes_batch.py`:

  * convolve_with_rir ........ :12-44   per channel c: fftconvolve(waveform[b], rir[b][c], "full")[..., :T]
  * aggregate_noise .......... :47-71   sum over the noise positions i of convolve_with_rir(noise, noise_rirs[:, i])
  * process_audio ............ :74-106
  * add_noise ................ :108-150  segmental SNR: norms over the window [start, start + length) only
  * generate_scene ........... :152-188  four cases

`torchaudio.functional.fftconvolve` (a third-party dependency, not vendored in the reference and not installed here) is, per
torchaudio's published definition, `irfft(rfft(x, n) * rfft(y, n), n)` with `n = len(x) + len(y) - 1`; the oracle computes the
same linear convolution that way in float64 (or in float32 with `dtype=np.float32` to mimic the reference's own rounding).

Parity pinning: `tests/test_oracle_golden.py::test_scene_oracle_matches_reference_fixture` checks this file against
`tests/golden/scene.npz`, produced by running the reference's module itself in the build container (`tests/golden/make_golden.py
scene`, with that fftconvolve definition supplied for the missing dependency).
"""
import numpy as np


def fftconvolve_full(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    n = x.shape[-1] + y.shape[-1] - 1
    return np.fft.irfft(np.fft.rfft(x, n=n) * np.fft.rfft(y, n=n), n=n)


def convolve_with_rir(waveform: np.ndarray, rir: np.ndarray, dtype=np.float64) -> np.ndarray:
    """waveform [B, T], rir [B, C, L] -> [B, C, T]"""
    assert waveform.shape[0] == rir.shape[0], "Not compatible for this operation"
    B, T = waveform.shape
    out = np.empty((B, rir.shape[1], T), dtype=dtype)
    for b in range(B):
        for c in range(rir.shape[1]):
            out[b, c] = fftconvolve_full(waveform[b].astype(dtype), rir[b, c].astype(dtype))[:T]
    return out


def aggregate_noise(noise_rirs: np.ndarray, noise_source: np.ndarray, dtype=np.float64) -> np.ndarray:
    """noise_rirs [B, n, C, L], noise_source [B, T] -> [B, C, T]"""
    B, T = noise_source.shape
    agg = np.zeros((B, noise_rirs.shape[2], T), dtype=dtype)
    for i in range(noise_rirs.shape[1]):
        agg += convolve_with_rir(noise_source, noise_rirs[:, i], dtype)
    return agg


def add_noise(source: np.ndarray, noise: np.ndarray, snr, start_idx, real_noise_length, dtype=np.float64) -> np.ndarray:
    """source / noise [B, C, T]; snr, start_idx, real_noise_length: [B] arrays or scalars."""
    B, C, T = source.shape
    source, noise = source.astype(dtype), noise.astype(dtype)
    start = np.broadcast_to(np.asarray(start_idx), (B,)).reshape(B, 1, 1)
    length = np.broadcast_to(np.asarray(real_noise_length), (B,)).reshape(B, 1, 1)
    t = np.arange(T).reshape(1, 1, T)
    mask = (t >= start) & (t < start + length)
    norm_x = np.sqrt(((source * mask) ** 2).sum(-1, keepdims=True))
    norm_n = np.sqrt(((noise * mask) ** 2).sum(-1, keepdims=True))
    snr_t = np.broadcast_to(np.asarray(snr, dtype=dtype).reshape(-1), (B,)).reshape(B, 1, 1)
    a = np.sqrt(norm_x ** 2 / (norm_n ** 2 + 1e-9) * 10.0 ** (-snr_t / 10.0))
    return source + a * noise


def generate_scene(source_rir, noise_rirs, source, noise, real_noise_length, noise_start_idx, snr, dtype=np.float64):
    if source_rir is not None and noise is not None:
        conv = convolve_with_rir(source, source_rir[:, [0], :], dtype)
        agg = aggregate_noise(noise_rirs[:, :, [0], :], noise, dtype)[:, :, :source.shape[-1]]
        return add_noise(conv, agg, snr, noise_start_idx, real_noise_length, dtype)
    if source_rir is not None:
        return convolve_with_rir(source, source_rir[:, [0], :], dtype)
    if noise is not None:
        return add_noise(source, noise, snr, noise_start_idx, real_noise_length, dtype)
    return source


# ----------------------------------------------------------------------------------------------------------------------
# Evaluation-time twin: reference hear_api/heaRIR/scene_module/generate_scenes.py (per clip; 1-D source / noise)
#   apply_fadein / apply_fadeout :11-33, add_noise :63-138 (= torchaudio.functional.add_noise), fade_noise :141-152,
#   aggregate_noise :155-167, process_audio :170-191, generate_scene :194-203
# ----------------------------------------------------------------------------------------------------------------------
def fade(audio: np.ndarray, sr: int, duration: float, fade_in: bool) -> np.ndarray:
    out = audio.astype(np.float64).copy()
    n = int(duration * sr)
    ramp = np.linspace(0.0, 1.0, n)
    if fade_in:
        out[:n] *= ramp
    else:
        out[out.shape[0] - n:] *= ramp[::-1]
    return out


def fade_noise(noise: np.ndarray, audio: np.ndarray, sr: int) -> np.ndarray:
    if noise.shape[-1] > audio.shape[-1]:
        return fade(noise[: audio.shape[-1]], sr, 0.2, False)
    return fade(fade(noise, sr, 0.2, True), sr, 0.2, False)


def add_noise_full(waveform: np.ndarray, noise: np.ndarray, snr, lengths=None) -> np.ndarray:
    """[..., L] tensors; energies over the first `lengths` samples (all when None)."""
    x, n = waveform.astype(np.float64), noise.astype(np.float64)
    L = x.shape[-1]
    if lengths is not None:
        m = np.arange(L) < np.asarray(lengths)[..., None]
        es, en = ((x * m) ** 2).sum(-1), ((n * m) ** 2).sum(-1)
    else:
        es, en = (x ** 2).sum(-1), (n ** 2).sum(-1)
    scale = 10.0 ** ((10.0 * (np.log10(es) - np.log10(en)) - np.asarray(snr, dtype=np.float64)) / 20.0)
    return x + scale[..., None] * n


def hear_generate_scene(source_rir: np.ndarray, noise_rirs, source: np.ndarray, noise, snr: float, sr: int, rng=np.random):
    """source 1-D, source_rir [C, L], noise_rirs list of [C, L], noise 1-D (or None with an empty noise_rirs) -> [C, T]"""
    conv = convolve_with_rir(source[None], source_rir[None])[0]
    if len(noise_rirs) == 0:
        return conv
    T = source.shape[-1]
    nz = fade_noise(noise, source, sr)
    agg = sum(convolve_with_rir(nz[None], r[None])[0] for r in noise_rirs)[:, :T]
    if conv.shape[1] > agg.shape[1]:
        start = rng.randint(0, T - agg.shape[1])
        placed = np.zeros_like(conv)
        placed[:, start:start + agg.shape[1]] = agg
        agg = placed
    return add_noise_full(conv, agg, np.full(conv.shape[0], snr))

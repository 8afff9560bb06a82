"""Audio file decoding for the real data path: the job `wds.torch_audio` / torchaudio.load does for the reference
(data_modules/WebAudioDataModule.py:104-109).  FLAC through the native decoder `libwavjepa_io.so` (csrc/flac_decode.cpp, C ABI,
called without the GIL from loader threads), PCM / float WAV through scipy.  Returns what torchaudio.load returns:
(float32 tensor [channels, samples] scaled to [-1, 1), sample_rate)."""
import ctypes
import hashlib
import io
import os
from typing import Tuple

import numpy as np
import torch

_LIB = None
ERRORS = {-1: "not a FLAC stream / malformed", -2: "truncated stream", -3: "CRC mismatch", -4: "unsupported feature", -5: "output too small"}


class FlacStreamInfo(ctypes.Structure):
    _fields_ = [("sample_rate", ctypes.c_int32), ("channels", ctypes.c_int32), ("bits_per_sample", ctypes.c_int32),
                ("min_block", ctypes.c_int32), ("max_block", ctypes.c_int32), ("total_samples", ctypes.c_int64), ("md5", ctypes.c_uint8 * 16)]


class AudioDecodeError(RuntimeError):
    pass


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libwavjepa_io.so")
        if not os.path.exists(path):
            raise AudioDecodeError(f"{path} is missing: build it with `python -m wavjepa_amd.build`")
        lib = ctypes.CDLL(path)
        lib.wj_flac_info.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.POINTER(FlacStreamInfo)]
        lib.wj_flac_info.restype = ctypes.c_int
        lib.wj_flac_decode.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
        lib.wj_flac_decode.restype = ctypes.c_int64
        _LIB = lib
    return _LIB


def flac_info(data: bytes) -> FlacStreamInfo:
    si = FlacStreamInfo()
    rc = _lib().wj_flac_info(data, len(data), ctypes.byref(si))
    if rc:
        raise AudioDecodeError(f"FLAC: {ERRORS.get(rc, rc)}")
    return si


MAX_SAMPLES = 1 << 26      # per channel (23 minutes at 48 kHz): what a training clip can reasonably hold; STREAMINFO is untrusted input


def decode_flac_pcm(data: bytes, verify_md5: bool = False, max_samples: int = MAX_SAMPLES) -> Tuple[np.ndarray, FlacStreamInfo]:
    """-> (int32 PCM [samples, channels], stream info).  Frame CRCs are always checked; `verify_md5` also checks the decoded samples
    against the STREAMINFO signature (skipped when the encoder left it zero).  The output is sized from STREAMINFO's sample count
    only up to `max_samples` (a corrupt 36-bit field must not ask for terabytes); a stream of unknown length is sized by a counting
    pass of the decoder (a silent clip compresses far below any bytes-based bound)."""
    si = flac_info(data)
    cap = int(si.total_samples)
    if cap == 0:
        cap = int(_lib().wj_flac_decode(data, len(data), None, 0))
        if cap < 0:
            raise AudioDecodeError(f"FLAC: {ERRORS.get(cap, cap)}")
    if cap > max_samples:
        raise AudioDecodeError(f"FLAC: {cap} samples per channel exceed the limit of {max_samples}")
    pcm = np.empty((max(cap, 1), si.channels), dtype=np.int32)
    n = _lib().wj_flac_decode(data, len(data), pcm.ctypes.data, cap)
    if n < 0:
        raise AudioDecodeError(f"FLAC: {ERRORS.get(int(n), int(n))}")
    pcm = pcm[:n]
    if verify_md5 and any(si.md5):
        nbytes = (si.bits_per_sample + 7) // 8
        raw = pcm.astype("<i4").tobytes()
        packed = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 4)[:, :nbytes].tobytes()
        if hashlib.md5(packed).digest() != bytes(si.md5):
            raise AudioDecodeError("FLAC: MD5 of the decoded samples does not match STREAMINFO")
    return pcm, si


def decode_flac(data: bytes, verify_md5: bool = False, max_samples: int = MAX_SAMPLES) -> Tuple[torch.Tensor, int]:
    pcm, si = decode_flac_pcm(data, verify_md5, max_samples)
    scale = np.float32(1.0 / float(1 << (si.bits_per_sample - 1)))
    wav = np.ascontiguousarray(pcm.T.astype(np.float32) * scale)
    return torch.from_numpy(wav), int(si.sample_rate)


def decode_wav(data: bytes) -> Tuple[torch.Tensor, int]:
    from scipy.io import wavfile
    sr, x = wavfile.read(io.BytesIO(data))
    if x.dtype.kind == "i":
        x = x.astype(np.float32) / float(1 << (8 * x.dtype.itemsize - 1))
    elif x.dtype.kind == "u":
        x = (x.astype(np.float32) - 128.0) / 128.0
    x = np.asarray(x, dtype=np.float32)
    x = x[None, :] if x.ndim == 1 else x.T
    return torch.from_numpy(np.ascontiguousarray(x)), int(sr)


def decode_audio(data: bytes, extension: str, **kw) -> Tuple[torch.Tensor, int]:
    ext = extension.lower().lstrip(".")
    if ext == "flac":
        return decode_flac(data, **kw)
    if ext == "wav":
        return decode_wav(data)
    raise AudioDecodeError(f"no decoder for .{ext} members")

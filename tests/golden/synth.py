"""Deterministic synthetic tensors shared by the golden generator and the tests (pure numpy, no RNG state).

Values come from a splitmix64 hash of (seed, element index), so every machine regenerates exactly the same
weights / inputs without storing them.  Used where a fixture would otherwise be too large to commit
(the base model has 196 M parameters).
"""
from __future__ import annotations

import zlib
from typing import Dict, Sequence, Tuple

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def hash_uniform(n: int, seed: int) -> np.ndarray:
    """n float64 values in [-1, 1)."""
    with np.errstate(over="ignore"):
        x = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return (x >> np.uint64(11)).astype(np.float64) / float(1 << 53) * 2.0 - 1.0


def hash_normal(n: int, seed: int) -> np.ndarray:
    """n approximately-normal float64 values (sum of 4 uniforms, unit variance)."""
    u = hash_uniform(4 * n, seed).reshape(4, n)
    return u.sum(axis=0) * np.sqrt(3.0 / 4.0)


def name_seed(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF


def synth_tensor(name: str, shape: Sequence[int], seed: int) -> np.ndarray:
    """Parameter-like values by tensor role (so a deep post-norm stack stays well conditioned)."""
    n = int(np.prod(shape))
    s = name_seed(name, seed)
    if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name.endswith("norm.weight") \
            or name.endswith("feature_norms.weight") or name.endswith("cnn.0.2.weight"):
        v = 1.0 + 0.1 * hash_uniform(n, s)
    elif name.endswith(".bias") or name.endswith("in_proj_bias"):
        v = 0.02 * hash_uniform(n, s)
    elif "cnn." in name:
        fan_in = int(np.prod(shape[1:]))
        v = np.sqrt(2.0 / fan_in) * hash_normal(n, s)
    elif name == "mask_token":
        v = 0.02 * hash_normal(n, s)
    else:
        fan_in = shape[-1]
        v = (1.0 / np.sqrt(fan_in)) * hash_uniform(n, s) * np.sqrt(3.0) * 0.6
    return v.reshape(shape).astype(np.float32)


def synth_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int) -> Dict[str, np.ndarray]:
    """All tensors except the fixed sin-cos tables (callers fill `pos_encoding_*`), teacher = copy of student."""
    out = {}
    for name, shape in shapes.items():
        if name.startswith("pos_encoding_") or name.startswith("teacher_encoder."):
            continue
        out[name] = synth_tensor(name, shape, seed)
    for name in list(out):
        if name.startswith("encoder."):
            out["teacher_" + name] = out[name].copy()
    return out


def synth_audio(n: int, channels: int, length: int, seed: int) -> np.ndarray:
    return hash_normal(n * channels * length, name_seed("audio", seed)).reshape(n, channels, length).astype(np.float32)


def jepa_shapes(*, conv_spec, in_channels: int, d_enc: int, enc_layers: int, d_dec: int, dec_layers: int,
                n_tokens: int) -> Dict[str, Tuple[int, ...]]:
    """state_dict names/shapes of the reference JEPA module (measured names)."""
    S: Dict[str, Tuple[int, ...]] = {}
    S["mask_token"] = (1, 1, d_dec)
    S["pos_encoding_encoder"] = (1, n_tokens, d_enc)
    S["pos_encoding_decoder"] = (1, n_tokens, d_dec)
    cin = in_channels
    for i, (dim, k, _s) in enumerate(conv_spec):
        S[f"extract_audio.cnn.{i}.0.weight"] = (dim, cin, k)
        if i == 0:
            S["extract_audio.cnn.0.2.weight"] = (dim,)
            S["extract_audio.cnn.0.2.bias"] = (dim,)
        cin = dim
    c_out = conv_spec[-1][0]
    S["feature_norms.weight"] = (c_out,)
    S["feature_norms.bias"] = (c_out,)

    def stack(prefix, d, layers):
        for i in range(layers):
            p = f"{prefix}.layers.{i}."
            S[p + "self_attn.in_proj_weight"] = (3 * d, d)
            S[p + "self_attn.in_proj_bias"] = (3 * d,)
            S[p + "self_attn.out_proj.weight"] = (d, d)
            S[p + "self_attn.out_proj.bias"] = (d,)
            S[p + "linear1.weight"] = (4 * d, d)
            S[p + "linear1.bias"] = (4 * d,)
            S[p + "linear2.weight"] = (d, 4 * d)
            S[p + "linear2.bias"] = (d,)
            S[p + "norm1.weight"] = (d,)
            S[p + "norm1.bias"] = (d,)
            S[p + "norm2.weight"] = (d,)
            S[p + "norm2.bias"] = (d,)
        S[f"{prefix}.norm.weight"] = (d,)
        S[f"{prefix}.norm.bias"] = (d,)

    stack("encoder", d_enc, enc_layers)
    if c_out != d_enc:
        S["post_extraction_mapper.weight"] = (d_enc, c_out)
        S["post_extraction_mapper.bias"] = (d_enc,)
    stack("decoder", d_dec, dec_layers)
    S["decoder_to_encoder_mapper.weight"] = (d_enc, d_dec)
    S["decoder_to_encoder_mapper.bias"] = (d_enc,)
    S["encoder_to_decoder_mapper.weight"] = (d_dec, d_enc)
    S["encoder_to_decoder_mapper.bias"] = (d_dec,)
    stack("teacher_encoder", d_enc, enc_layers)
    return S

"""Per-file preparation steps of the loader (reference data_modules/dataset_functions.py), on CPU tensors like upstream (they run
in loader workers on one clip at a time; the per-batch work -- crops, normalisation, scene augmentation -- is on the GPU)."""
import torch
import torch.nn.functional as F


def pad_or_truncate(feature: torch.Tensor, target_length: int) -> torch.Tensor:
    """[C, n] -> [C, target_length]: zero-pad at the end or cut (reference :4-30)."""
    n = feature.shape[1]
    if n < target_length:
        return F.pad(feature, (0, target_length - n))
    return feature[:, :target_length]


def pad_or_truncate_batch(feature: torch.Tensor, target_length: int) -> torch.Tensor:
    """[B, C, n] -> [B, C, target_length] (reference :34-60)."""
    n = feature.shape[-1]
    if n < target_length:
        return F.pad(feature, (0, target_length - n))
    return feature[:, :, :target_length]


def instance_normalize(feature: torch.Tensor) -> torch.Tensor:
    """(x - mean) / (std + 1e-8) over the whole tensor (reference :63-88)."""
    return (feature - feature.mean()) / (feature.std() + 1e-8)


def normalize_audio(audio_data: torch.Tensor, target_dBFS: float = -14.0) -> torch.Tensor:
    """Scale to an RMS level of `target_dBFS`; silence is returned unchanged (reference :90-98)."""
    rms = torch.sqrt(torch.mean(audio_data ** 2))
    if rms == 0:
        return audio_data
    gain_db = target_dBFS - 20 * torch.log10(rms)
    return audio_data * 10 ** (gain_db / 20)


def pre_process(waveform: torch.Tensor, sr: int) -> torch.Tensor:
    """RMS -14 dBFS, [1, samples], exactly 10 s (reference :100-111)."""
    waveform = normalize_audio(waveform, -14.0).reshape(1, -1)
    return pad_or_truncate(waveform, sr * 10)


def pre_process_noise(waveform: torch.Tensor) -> torch.Tensor:
    return normalize_audio(waveform, -14.0)

"""HEAR-2021 API wrapper around the MI355X JEPA module (same surface as reference hear_api/runtime.py:12-155:
`RuntimeJEPA(...).get_timestamp_embeddings(audio)`, `get_scene_embeddings`, attributes `sample_rate`,
`scene_embedding_size`, `timestamp_embedding_size`)."""
from __future__ import annotations

import torch

from wavjepa_amd.jepa import JEPA
from wavjepa_amd.types import TransformerEncoderCFG, TransformerLayerCFG

from .feature_helper import FeatureExtractor


def normalize(audio: torch.Tensor) -> torch.Tensor:
    mean = audio.mean(dim=(-2, -1), keepdim=True)
    std = audio.std(dim=(-2, -1), keepdim=True)
    return (audio - mean) / (std + 1e-5)


def calculate_padding_mask(pad_frames, total_frames, sr, output_steps, process_seconds, model, B):
    """Key-padding mask over the token grid of all windows + the cut-off index (reference runtime.py:19-35, including
    its integer-seconds arithmetic)."""
    n_chunks = int((total_frames / sr) / process_seconds)
    total_output_steps = output_steps * n_chunks
    mask = torch.zeros((B, total_output_steps), dtype=torch.bool, device=model.device)
    output_sr = int(output_steps / process_seconds)
    pad_steps = int((pad_frames / sr) * output_sr)
    mask[..., total_output_steps - pad_steps:] = True
    return mask, total_output_steps - pad_steps


def strip_compile_prefixes(state_dict):
    """Checkpoints written with torch.compile carry an `._orig_mod` infix (reference runtime.py:63-75)."""
    return {k.replace("._orig_mod", ""): v for k, v in state_dict.items()}


class RuntimeJEPA(torch.nn.Module):
    def __init__(self, in_channels, weights, is_spectrogram, process_seconds, extractor, model_size, sr, **kwargs) -> None:
        super().__init__()
        self.sample_rate = sr
        self.in_channels = int(in_channels)
        self.model = JEPA(feature_extractor=extractor, transformer_encoder_cfg=TransformerEncoderCFG.create(),
                          transformer_encoder_layers_cfg=TransformerLayerCFG.create(), transformer_decoder_cfg=TransformerEncoderCFG.create(),
                          transformer_decoder_layers_cfg=TransformerLayerCFG.create(d_model=384), resample_sr=self.sample_rate,
                          size=model_size, process_audio_seconds=process_seconds)
        if weights is not None:
            self.model.load_state_dict(strip_compile_prefixes(weights["state_dict"]), strict=False)
        self.embedding_size = self.model.encoder_embedding_dim
        self.scene_embedding_size = self.embedding_size
        self.timestamp_embedding_size = self.embedding_size
        self.unit_frames = int(process_seconds * self.sample_rate)
        self.output_steps = self.steps_per_window(self.model.extract_audio.total_patches(self.unit_frames))
        if torch.cuda.is_available():
            self.model.cuda()
        self.model.eval()
        self.feature_extractor = FeatureExtractor(in_channels=in_channels)

    # the two points where the multi-channel runtime (runtime_natjepa.py) differs from this one
    def steps_per_window(self, window_tokens: int) -> int:
        """Embedding steps one window contributes to the output (here: every token of the window)."""
        return window_tokens

    def window_embedding(self, window: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        """[B, C, unit_frames] + key-padding mask over the window's `output_steps` -> [B, output_steps, D]."""
        return self.model.get_audio_representation(normalize(window), mask)

    def to_feature(self, batch_audio):
        return self.feature_extractor(batch_audio)

    def get_scene_embeddings(self, audio):
        embeddings, _ = self.get_timestamp_embeddings(audio)
        return torch.mean(embeddings, dim=1)

    def get_timestamp_embeddings(self, audio):
        B = audio.shape[0]
        audio = self.to_feature(audio)
        input_audio_len = audio.shape[-1]
        if audio.ndim != 3:
            raise ValueError("audio input tensor must be 2D with shape (n_sounds, n_channels, num_samples)")
        cur_frames = audio.shape[-1]
        pad_frames = self.unit_frames - (cur_frames % self.unit_frames)
        if pad_frames > 0:
            audio = torch.nn.functional.pad(audio, (0, pad_frames), mode="constant")
        padding_mask, cut_off = calculate_padding_mask(pad_frames=pad_frames, total_frames=audio.shape[-1], sr=self.sample_rate,
                                                       output_steps=self.output_steps,
                                                       process_seconds=self.model.target_length // self.sample_rate,
                                                       model=self.model, B=B)
        embeddings, mask_idx = [], 0
        for i in range(audio.shape[-1] // self.unit_frames):
            window = audio[..., i * self.unit_frames:(i + 1) * self.unit_frames]
            mask = padding_mask[..., mask_idx:mask_idx + self.output_steps]
            if mask.shape[-1] < self.output_steps:
                mask = torch.nn.functional.pad(mask, (0, self.output_steps - mask.shape[-1]), value=True)
            embeddings.append(self.window_embedding(window, mask))
            mask_idx += self.output_steps
        x = torch.hstack(embeddings)[:, :cut_off, :]
        ts = get_timestamps(self.sample_rate, B, input_audio_len, x)
        assert ts.shape[-1] == x.shape[1]
        return x, ts


def get_timestamps(sample_rate, B, input_audio_len, x):
    sec = input_audio_len / sample_rate
    x_len = x.shape[1]
    step = sec / x_len * 1000
    ts = torch.tensor([step * i for i in range(x_len)]).unsqueeze(0)
    return ts.repeat(B, 1)

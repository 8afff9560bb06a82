"""HEAR-2021 API wrapper around the MI355X JEPA module.  Surface of reference hear_api/runtime.py:12-155:
`RuntimeJEPA(...).get_timestamp_embeddings(audio)` / `get_scene_embeddings(audio)`, attributes `sample_rate`,
`scene_embedding_size`, `timestamp_embedding_size`, and the module-level helpers `normalize`, `calculate_padding_mask`,
`get_timestamps`.

How a clip becomes embeddings (same arithmetic as upstream, organised around a window plan):
  1. loudness-normalise and fix the channel count (feature_helper), zero-pad up to the next multiple of the window length (a clip
     that already is a multiple gets one more, all-padding window, as upstream);
  2. every window is normalised on its own and encoded with a key-padding mask over the embedding steps that lie in the padding;
  3. windows are concatenated and cut at the first padded step; step i is stamped  i * clip_seconds / n_steps  (ms).
Upstream computes the number of mask windows with INTEGER seconds (`target_length // sr`: 2 for 2.01 s windows), so the mask can
be longer or shorter than the window grid; that is kept (`WindowPlan.key_mask` is cut / extended with `True`)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import torch

from wavjepa_amd.jepa import JEPA
from wavjepa_amd.types import TransformerEncoderCFG, TransformerLayerCFG

from .feature_helper import FeatureExtractor


def normalize(audio: torch.Tensor) -> torch.Tensor:
    """Zero mean, unit (unbiased) deviation over channels x time of every window."""
    centred = audio - audio.mean(dim=(-2, -1), keepdim=True)
    return centred / (audio.std(dim=(-2, -1), keepdim=True) + 1e-5)


def calculate_padding_mask(pad_frames, total_frames, sr, output_steps, process_seconds, model, B):
    """(mask [B, steps] with True on the steps that fall into the trailing padding, index of the first such step).
    `steps` = output_steps x the number of `process_seconds` chunks in the padded clip; the padded tail, converted to steps at
    int(output_steps / process_seconds) steps per second, is what gets masked (reference runtime.py:19-35)."""
    steps = output_steps * int(total_frames / sr / process_seconds)
    first_padded = steps - int(pad_frames / sr * int(output_steps / process_seconds))
    start = first_padded if first_padded >= 0 else max(steps + first_padded, 0)       # upstream writes mask[..., first_padded:] = True
    mask = (torch.arange(steps, device=model.device) >= start).expand(B, steps).clone()
    return mask, first_padded


def get_timestamps(sample_rate, B, input_audio_len, x):
    """Start time in ms of every embedding step: the clip's duration spread evenly over the steps (reference runtime.py:145-155)."""
    n_steps = x.shape[1]
    step_ms = input_audio_len / sample_rate / n_steps * 1000
    return torch.tensor([step_ms * i for i in range(n_steps)]).unsqueeze(0).repeat(B, 1)


def strip_compile_prefixes(state_dict):
    """Checkpoints written with torch.compile carry an `._orig_mod` infix (reference runtime.py:63-75)."""
    return {k.replace("._orig_mod", ""): v for k, v in state_dict.items()}


@dataclass
class WindowPlan:
    pad_frames: int          # zeros appended to the clip
    n_windows: int
    cut_off: int             # embedding steps kept
    key_masks: List[torch.Tensor]     # per window: [B, steps_per_window] bool, True = padded step


class RuntimeJEPA(torch.nn.Module):
    def __init__(self, in_channels, weights, is_spectrogram, process_seconds, extractor, model_size, sr, **kwargs) -> None:
        super().__init__()
        self.sample_rate = sr
        self.in_channels = int(in_channels)
        layer, stack = TransformerLayerCFG.create, TransformerEncoderCFG.create
        self.model = JEPA(feature_extractor=extractor, transformer_encoder_cfg=stack(), transformer_encoder_layers_cfg=layer(),
                          transformer_decoder_cfg=stack(), transformer_decoder_layers_cfg=layer(d_model=384),
                          resample_sr=sr, size=model_size, process_audio_seconds=process_seconds)
        if weights is not None:
            self.model.load_state_dict(strip_compile_prefixes(weights["state_dict"]), strict=False)
        self.embedding_size = self.scene_embedding_size = self.timestamp_embedding_size = self.model.encoder_embedding_dim
        self.unit_frames = int(process_seconds * sr)
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

    def window_plan(self, n_frames: int, B: int) -> WindowPlan:
        pad = self.unit_frames - n_frames % self.unit_frames            # a full extra window when n_frames is a multiple
        total = n_frames + pad
        mask, cut = calculate_padding_mask(pad_frames=pad, total_frames=total, sr=self.sample_rate, output_steps=self.output_steps,
                                           process_seconds=self.model.target_length // self.sample_rate, model=self.model, B=B)
        n_windows, S = total // self.unit_frames, self.output_steps
        if mask.shape[-1] < n_windows * S:                              # integer-seconds chunk count fell short of the window grid
            mask = torch.nn.functional.pad(mask, (0, n_windows * S - mask.shape[-1]), value=True)
        return WindowPlan(pad, n_windows, cut, [mask[:, w * S:(w + 1) * S] for w in range(n_windows)])

    def get_timestamp_embeddings(self, audio):
        feats = self.to_feature(audio)
        if feats.ndim != 3:
            raise ValueError("audio input tensor must be 2D with shape (n_sounds, n_channels, num_samples)")
        B, n_in = audio.shape[0], feats.shape[-1]
        plan = self.window_plan(n_in, B)
        feats = torch.nn.functional.pad(feats, (0, plan.pad_frames))
        per_window = [self.window_embedding(feats[..., w * self.unit_frames:(w + 1) * self.unit_frames], plan.key_masks[w])
                      for w in range(plan.n_windows)]
        x = torch.cat(per_window, dim=1)[:, :plan.cut_off]
        ts = get_timestamps(self.sample_rate, B, n_in, x)
        assert ts.shape[-1] == x.shape[1]
        return x, ts

    def get_scene_embeddings(self, audio):
        return self.get_timestamp_embeddings(audio)[0].mean(dim=1)

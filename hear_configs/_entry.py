"""The three functions a HEAR-2021 module exposes (`load_model`, `get_scene_embeddings`, `get_timestamp_embeddings`), built for one
conv spec / window length / runtime class.  reference hear_configs/WavJEPA.py:11-43 and WavJEPA_w2v2.py:11-45 spell the same three
functions out per model; here each model is one call of `hear_entry_points`."""
from __future__ import annotations

from typing import Callable, Sequence, Tuple

import torch


def hear_entry_points(conv_spec: Sequence[Tuple[int, int, int]], process_seconds: float, *, sr: int = 16000, in_channels: int = 1,
                      model_size: str = "base", runtime=None, extractor=None) -> Tuple[Callable, Callable, Callable]:
    def load_model(*args, **kwargs):
        """load_model()  -> randomly initialised runtime;  load_model(path) -> weights from a Lightning checkpoint (`state_dict` key)."""
        from hear_api.runtime import RuntimeJEPA
        from wavjepa_amd.extractors import ConvFeatureExtractor
        weights = torch.load(args[0], weights_only=False, map_location="cpu") if args else None
        ext_cls = extractor or ConvFeatureExtractor
        rt_cls = runtime or RuntimeJEPA
        return rt_cls(in_channels=in_channels, process_seconds=process_seconds, weights=weights, sr=sr, model_size=model_size,
                      is_spectrogram=False, extractor=ext_cls(conv_layers_spec=list(conv_spec), in_channels=in_channels))

    def get_scene_embeddings(audio, model):
        return model.get_scene_embeddings(audio)

    def get_timestamp_embeddings(audio, model):
        return model.get_timestamp_embeddings(audio)

    return load_model, get_scene_embeddings, get_timestamp_embeddings

"""HEAR module contract: load_model / get_scene_embeddings / get_timestamp_embeddings (reference hear_configs/WavJEPA.py:11-43)."""
import torch

from hear_api.runtime import RuntimeJEPA
from wavjepa_amd.extractors import ConvFeatureExtractor

SR = 16000


def load_model(*args, **kwargs):
    weights = None
    if len(args) != 0:
        weights = torch.load(args[0], weights_only=False, map_location="cpu")
    extractor = ConvFeatureExtractor(conv_layers_spec=[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)], in_channels=1)
    return RuntimeJEPA(in_channels=1, process_seconds=2.01, weights=weights, sr=SR, model_size="base", is_spectrogram=False,
                       extractor=extractor)


def get_scene_embeddings(audio, model):
    return model.get_scene_embeddings(audio)


def get_timestamp_embeddings(audio, model):
    return model.get_timestamp_embeddings(audio)

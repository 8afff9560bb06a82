"""HEAR module contract for the 7-layer wav2vec2-spec model on 4.02 s windows (reference hear_configs/WavJEPA_w2v2.py:11-45)."""
import torch

from hear_api.runtime import RuntimeJEPA
from wavjepa_amd.extractors import ConvFeatureExtractor

SR = 16000
W2V2_CONV_SPEC = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] + [(512, 2, 2)]


def load_model(*args, **kwargs):
    weights = None
    if len(args) != 0:
        weights = torch.load(args[0], weights_only=False, map_location="cpu")
    extractor = ConvFeatureExtractor(conv_layers_spec=list(W2V2_CONV_SPEC), in_channels=1)
    return RuntimeJEPA(in_channels=1, process_seconds=4.02, weights=weights, sr=SR, model_size="base", is_spectrogram=False,
                       extractor=extractor)


def get_scene_embeddings(audio, model):
    return model.get_scene_embeddings(audio)


def get_timestamp_embeddings(audio, model):
    return model.get_timestamp_embeddings(audio)

"""HEAR module of the binaural model: every channel through its own conv stack, `RuntimeNatJEPA` (reference
hear_api/runtime_natjepa.py:38-155; upstream ships no hear_configs module for it)."""
from hear_api.runtime_natjepa import RuntimeNatJEPA
from hear_configs._entry import hear_entry_points
from wavjepa_amd.extractors import ConvChannelFeatureExtractor

SR = 16000
CONV_SPEC = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)]
load_model, get_scene_embeddings, get_timestamp_embeddings = hear_entry_points(CONV_SPEC, 2.01, sr=SR, in_channels=2, runtime=RuntimeNatJEPA,
                                                                                extractor=ConvChannelFeatureExtractor)

"""HEAR module of the base model: 6-layer conv spec, 2.01 s windows -> 200 steps (reference hear_configs/WavJEPA.py:11-43)."""
from hear_configs._entry import hear_entry_points

SR = 16000
CONV_SPEC = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)]
load_model, get_scene_embeddings, get_timestamp_embeddings = hear_entry_points(CONV_SPEC, 2.01, sr=SR)

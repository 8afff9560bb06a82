"""HEAR module of the 7-layer wav2vec2-spec model on 4.02 s windows -> 200 steps (reference hear_configs/WavJEPA_w2v2.py:11-45)."""
from hear_configs._entry import hear_entry_points

SR = 16000
W2V2_CONV_SPEC = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] + [(512, 2, 2)]
load_model, get_scene_embeddings, get_timestamp_embeddings = hear_entry_points(W2V2_CONV_SPEC, 4.02, sr=SR)

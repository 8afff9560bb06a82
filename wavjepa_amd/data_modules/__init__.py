"""The real data path (reference data_modules/): shard reader, audio decode, resampling, loudness normalisation,
10 s padding, mask generation and batching for the JEPA step."""
from .WebAudioDataModule import WebAudioDataModule as WebAudioDataModule
from .WebAudioDataModuleDenoiser import WebAudioDataModuleDenoiser as WebAudioDataModuleDenoiser

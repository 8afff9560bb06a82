"""Evaluation-time scene augmentation (reference hear_api/heaRIR): RIR convolution + noise mixing for HEAR clips, on the GPU."""
from .augment import Augmenter as Augmenter

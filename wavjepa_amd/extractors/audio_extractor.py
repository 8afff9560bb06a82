"""Extractor interface (reference wavjepa/extractors/audio_extractor.py:6-20)."""
from abc import ABC, abstractmethod

import torch


class Extractor(ABC):
    embedding_dim: int

    @abstractmethod
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        ...

    @abstractmethod
    def total_patches(self, time: int) -> int:
        ...

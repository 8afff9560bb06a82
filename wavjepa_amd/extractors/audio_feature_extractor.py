"""wav2vec2-style stacked Conv1d waveform encoder on MI355X.

Same constructor, attributes and state_dict names as reference
wavjepa/extractors/audio_feature_extractor.py:13-154 (`cnn.{l}.0.weight`, `cnn.0.2.{weight,bias}`), but the modules in
`self.cnn` are only parameter containers: `forward` runs the HIP path (conv0+GroupNorm+GELU kernel, then one
implicit-GEMM per layer over a channels-last activation) through `wavjepa_amd.engine`.
"""
from __future__ import annotations

from math import prod
from typing import List, Optional, Sequence, Tuple

import torch
from torch import nn

from .audio_extractor import Extractor


class ConvFeatureExtractor(Extractor, nn.Module):
    def __init__(self, *args, conv_layers_spec: Sequence[Tuple[int, int, int]], in_channels: int = 2, dropout: float = 0.0,
                 mode: str = "default", conv_bias: bool = False, depthwise: bool = False, **kwargs):
        nn.Module.__init__(self)
        if mode != "default":
            raise NotImplementedError("only mode='default' (GroupNorm on layer 0) is on the accelerated path")
        if conv_bias or depthwise or dropout != 0.0:
            raise NotImplementedError("conv_bias / depthwise / dropout are not used by the WavJEPA configs")
        self.in_channels = in_channels
        self.depthwise = depthwise
        self.conv_layers_spec = [tuple(int(v) for v in cl) for cl in conv_layers_spec]
        layers, c_in = [], in_channels
        for i, (dim, k, stride) in enumerate(self.conv_layers_spec):
            conv = nn.Conv1d(c_in, dim, k, stride=stride, bias=False)
            nn.init.kaiming_normal_(conv.weight)
            if i == 0:
                block = nn.Sequential(conv, nn.Dropout(p=0.0), nn.GroupNorm(dim, dim, affine=True), nn.GELU())
            else:
                block = nn.Sequential(conv, nn.Dropout(p=0.0), nn.GELU())
            layers.append(block)
            c_in = dim
        self.cnn = nn.Sequential(*layers)
        self.embedding_dim = self.conv_layers_spec[-1][0]
        self._standalone = None

    def total_patches(self, time: int, device: str = "cuda") -> int:
        """Output frames for `time` samples: floor((L - k) / s) + 1 per layer (the reference measures it by a dummy forward)."""
        for _, k, s in self.conv_layers_spec:
            time = (time - k) // s + 1
        return time

    @property
    def receptive_fields(self) -> List[int]:
        rf, out = 1, [1]
        for _, width, stride in reversed(self.conv_layers_spec):
            rf = (rf - 1) * stride + width
            out.append(rf)
        return list(reversed(out))

    def description(self, sfreq: Optional[int] = None, dummy_time: Optional[int] = None) -> str:
        dims, _, strides = zip(*self.conv_layers_spec)
        rf = self.receptive_fields[0]
        ds = prod(strides)
        desc = f"Receptive field: {rf} samples | Downsampled by {ds} | Overlap of {rf - ds} samples"
        if dummy_time is not None:
            desc += f" | {self.total_patches(dummy_time)} encoded samples/trial"
        return desc

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [N, C_in, L] -> tokens [N, T, C] (bf16).  Standalone use of the front-end kernels (no gradient)."""
        from ..standalone import conv_frontend_tokens
        return conv_frontend_tokens(self, x)

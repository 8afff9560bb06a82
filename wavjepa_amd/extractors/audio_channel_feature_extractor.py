"""Per-channel stacked Conv1d waveform encoder (WavJEPA-Nat front-end, BASELINE config 4) on MI355X.

Same constructor, attributes and state_dict names as reference
wavjepa/extractors/audio_channel_feature_extractor.py:13-199 (`cnns.{c}.{l}.0.weight`, `cnns.{c}.0.2.{weight,bias}`; a single
`cnns.0` when `share_weights_over_channels`): every audio channel goes through a MONO conv stack (its own, or the shared one) and
the per-channel token sequences are flattened channel-major, "B (C S)" (:174-179), so a clip yields in_channels * T tokens.
The modules in `self.cnns` are parameter containers only: `wavjepa_amd.engine` runs the channels as independent mono clips of
one batch (the same conv0 / implicit-GEMM kernels as ConvFeatureExtractor) and only the LayerNorm that follows re-orders rows.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
from torch import nn

from .audio_extractor import Extractor


class ConvChannelFeatureExtractor(Extractor, nn.Module):
    def __init__(self, *args, conv_layers_spec: Sequence[Tuple[int, int, int]], in_channels: int = 2, dropout: float = 0.0,
                 mode: str = "default", conv_bias: bool = False, depthwise: bool = False, share_weights_over_channels: bool = False,
                 **kwargs):
        nn.Module.__init__(self)
        if mode != "default":
            raise NotImplementedError("only mode='default' (GroupNorm on layer 0) is on the accelerated path")
        if conv_bias or dropout != 0.0:
            raise NotImplementedError("conv_bias / dropout are not used by the WavJEPA configs")
        # depthwise: every stack starts from ONE input channel, so groups = n_in is a plain convolution in layer 0; deeper layers
        # (groups = dim) would be true depthwise convolutions, which no WavJEPA config selects
        if depthwise:
            raise NotImplementedError("depthwise conv stacks are not on the accelerated path")
        self.in_channels = int(in_channels)
        self.depthwise = depthwise
        self.conv_layers_spec = [tuple(int(v) for v in cl) for cl in conv_layers_spec]
        self.weight_sharing = bool(share_weights_over_channels)
        self.cnns = nn.ModuleList()
        for _ in range(1 if self.weight_sharing else self.in_channels):
            layers, c_in = [], 1
            for i, (dim, k, stride) in enumerate(self.conv_layers_spec):
                conv = nn.Conv1d(c_in, dim, k, stride=stride, bias=False)
                nn.init.kaiming_normal_(conv.weight)
                if i == 0:
                    layers.append(nn.Sequential(conv, nn.Dropout(p=0.0), nn.GroupNorm(dim, dim, affine=True), nn.GELU()))
                else:
                    layers.append(nn.Sequential(conv, nn.Dropout(p=0.0), nn.GELU()))
                c_in = dim
            self.cnns.append(nn.Sequential(*layers))
        self.embedding_dim = self.conv_layers_spec[-1][0]

    def frames_per_channel(self, time: int) -> int:
        for _, k, s in self.conv_layers_spec:
            time = (time - k) // s + 1
        return time

    def total_patches(self, time: int) -> int:
        """Tokens per clip: in_channels * frames (the reference measures it by a dummy forward, :181-186)."""
        return self.in_channels * self.frames_per_channel(time)

    @property
    def receptive_fields(self) -> List[int]:
        rf, out = 1, [1]
        for _, width, stride in reversed(self.conv_layers_spec):
            rf = (rf - 1) * stride + width
            out.append(rf)
        return list(reversed(out))

    def stack_prefix(self, channel: int) -> str:
        """state_dict prefix (relative to this module) of the conv stack channel `channel` runs through."""
        return f"cnns.{0 if self.weight_sharing else channel}."

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [N, C, L] -> tokens [N, C*T, dim] (bf16).  Standalone use of the front-end kernels (no gradient)."""
        from ..standalone import conv_frontend_tokens
        return conv_frontend_tokens(self, x)

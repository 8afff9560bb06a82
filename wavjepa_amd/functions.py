"""Initialisers used by the JEPA module (reference wavjepa/functions.py:9-49)."""
from __future__ import annotations

import math

import torch


def trunc_normal_(tensor: torch.Tensor, mean: float = 0.0, std: float = 1.0, a: float = -2.0, b: float = 2.0) -> torch.Tensor:
    """Truncated normal by inverse-CDF sampling: uniform on [cdf(a), cdf(b)] -> erfinv -> scale/shift -> clamp."""
    def cdf(x: float) -> float:
        return (1.0 + math.erf(x / math.sqrt(2.0))) / 2.0

    with torch.no_grad():
        lo, hi = cdf((a - mean) / std), cdf((b - mean) / std)
        tensor.uniform_(2 * lo - 1, 2 * hi - 1).erfinv_().mul_(std * math.sqrt(2.0)).add_(mean).clamp_(min=a, max=b)
    return tensor

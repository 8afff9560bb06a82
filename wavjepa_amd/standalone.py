"""Stand-alone use of the conv front-end kernels (ConvFeatureExtractor.forward outside a JEPA module)."""
from __future__ import annotations

import torch

from . import ops
from .engine import conv_geometry


@torch.no_grad()
def conv_frontend_tokens(extractor, x: torch.Tensor) -> torch.Tensor:
    """x [N, C_in, L] on the GPU -> tokens bf16 (conv0+GroupNorm+GELU kernel, then one implicit GEMM per layer).
    ConvFeatureExtractor: [N, T, C].  ConvChannelFeatureExtractor: every channel through its mono stack, flattened channel-major
    [N, C_in * T, C] (reference audio_channel_feature_extractor.py:154-179)."""
    ops.require_gpu()
    if not x.is_cuda:
        raise RuntimeError("the conv front-end needs GPU tensors (no CPU fallback)")
    audio = x.to(torch.bfloat16).contiguous()
    N, C_in, n_samples = audio.shape
    if hasattr(extractor, "cnns"):
        outs = [_stack_tokens(extractor.cnns[0 if extractor.weight_sharing else c], extractor.conv_layers_spec, audio, c, 1, C_in * n_samples)
                for c in range(C_in)]
        return torch.stack(outs, dim=1).flatten(1, 2)
    return _stack_tokens(extractor.cnn, extractor.conv_layers_spec, audio, 0, C_in, 0)


def _stack_tokens(cnn, spec, audio: torch.Tensor, first_channel: int, C_in: int, clip_stride: int) -> torch.Tensor:
    """One conv stack over channels [first_channel, first_channel + C_in) of `audio` [N, C, L] bf16 -> [N, T, C_out] bf16."""
    dev, bf = audio.device, torch.bfloat16
    N, n_samples = audio.shape[0], audio.shape[2]
    C = spec[-1][0]
    L, P = conv_geometry(n_samples, spec)
    w0 = cnn[0][0].weight.detach().to(dev, bf).contiguous()
    gn = cnn[0][2]
    gamma, beta = gn.weight.detach().float().contiguous(), gn.bias.detach().float().contiguous()

    def rows(n):
        t = torch.zeros((2 + n + 8) * C, dtype=bf, device=dev)
        return t, t.data_ptr() + 2 * C * 2

    post, post_p = rows(N * P[0])
    stats = torch.empty(2, N, C, device=dev)
    _, k0, s0 = spec[0]
    ws = torch.empty(ops.workspace_bytes("wj_conv0_gn_gelu_fwd", N=N, C_in=C_in, C=C, k=k0, L_out=L[0]) // 4, device=dev)
    ops.conv0_fwd(audio.data_ptr() + first_channel * n_samples * 2, w0, gamma, beta, post_p, stats[0], stats[1], ws, N=N, C_in=C_in,
                  L=n_samples, C=C, k=k0, stride=s0, L_out=L[0], P=P[0], audio_clip_stride=clip_stride)
    keep = [post]
    for l in range(1, len(spec)):
        _, k, s = spec[l]
        w = cnn[l][0].weight.detach().to(dev, torch.float32).contiguous()
        wp = torch.empty(C, k * C, dtype=bf, device=dev)
        ops.conv_weight_layout(w, wp, C_out=C, C_in=C, k=k, mode=0)
        pre, pre_p = rows(N * P[l])
        nxt, nxt_p = rows(N * P[l])
        ops.gemm(post_p, wp, pre_p, C2=nxt_p, M=N * P[l], N=C, K=k * C, lda=s * C, ldb=k * C, ldc=C, epilogue=ops.EPI_CONV_GELU,
                 seg_rows=P[l], seg_valid=L[l])
        keep += [pre, nxt, wp, w]
        post, post_p = nxt, nxt_p
    out = post[2 * C:(2 + N * P[-1]) * C].view(N, P[-1], C)[:, :L[-1]].contiguous()
    torch.cuda.current_stream().synchronize()   # temporaries above must outlive the kernels
    return out

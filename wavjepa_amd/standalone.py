"""Stand-alone use of the conv front-end kernels (ConvFeatureExtractor.forward outside a JEPA module)."""
from __future__ import annotations

import torch

from . import ops
from .engine import conv_geometry


@torch.no_grad()
def conv_frontend_tokens(extractor, x: torch.Tensor) -> torch.Tensor:
    """x [N, C_in, L] on the GPU -> tokens [N, T, C] bf16 (conv0+GroupNorm+GELU kernel, then one implicit GEMM per layer)."""
    ops.require_gpu()
    if not x.is_cuda:
        raise RuntimeError("ConvFeatureExtractor.forward needs GPU tensors (no CPU fallback)")
    spec = extractor.conv_layers_spec
    dev, bf = x.device, torch.bfloat16
    N, C_in, n_samples = x.shape
    C = spec[-1][0]
    L, P = conv_geometry(n_samples, spec)
    audio = x.to(bf).contiguous()
    w0 = extractor.cnn[0][0].weight.detach().to(dev, bf).contiguous()
    gn = extractor.cnn[0][2]
    gamma, beta = gn.weight.detach().float().contiguous(), gn.bias.detach().float().contiguous()

    def rows(n):
        t = torch.zeros((2 + n + 8) * C, dtype=bf, device=dev)
        return t, t.data_ptr() + 2 * C * 2

    post, post_p = rows(N * P[0])
    stats = torch.empty(2, N, C, device=dev)
    _, k0, s0 = spec[0]
    ws = torch.empty(ops.workspace_bytes("wj_conv0_gn_gelu_fwd", N=N, C_in=C_in, C=C, k=k0, L_out=L[0]) // 4, device=dev)
    ops.conv0_fwd(audio, w0, gamma, beta, post_p, stats[0], stats[1], ws, N=N, C_in=C_in, L=n_samples, C=C, k=k0, stride=s0,
                  L_out=L[0], P=P[0])
    keep = [post]
    for l in range(1, len(spec)):
        _, k, s = spec[l]
        w = extractor.cnn[l][0].weight.detach().to(dev, torch.float32).contiguous()
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

#!/usr/bin/env python3
"""Attention / LayerNorm micro-benchmark at the step's shapes (GPU box): the dense teacher shapes and the ragged
student / predictor shapes of a 256-clip AudioSet-masker batch.  HIP-event timing, median of 5 x 3 launches."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavjepa_amd import ops  # noqa: E402
from wavjepa_amd.masking import TimeInverseBlockMasker  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16


def timeit(fn, n=5, reps=3):
    ts = []
    for r in range(n + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) / reps)
    return sorted(ts)[len(ts) // 2]


torch.manual_seed(0)
import random  # noqa: E402
random.seed(0)
np.random.seed(0)
masker = TimeInverseBlockMasker(target_masks_per_context=4, context_mask_prob=0.65, context_mask_length=10, target_prob=0.25,
                                target_length=10, ratio_cutoff=0.1)
ctx, tgt, vis = masker(256, 200, 1)
enc_len = (~ctx).sum(-1).numpy()
dec_len = (~vis).reshape(-1, 200).sum(-1).numpy()


def attention(tag, lens, H, hd, T_dense=None):
    D = H * hd
    B = len(lens)
    if T_dense:                       # dense, no mask (teacher)
        rows, T, off = B * T_dense, T_dense, None
        flops = 4.0 * T_dense * T_dense * D * B
    else:
        off_np = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        rows, T, off = int(off_np[-1]), int(lens.max()), torch.from_numpy(off_np).to(dev)
        flops = 4.0 * D * float((lens.astype(np.float64) ** 2).sum())
    qkv = torch.randn(rows, 3 * D, device=dev).to(bf)
    out = torch.empty(rows, D, device=dev, dtype=bf)
    lse = torch.empty(rows * H, device=dev)
    dout = torch.randn(rows, D, device=dev).to(bf)
    dqkv = torch.empty_like(qkv)
    db = torch.zeros(3 * D, device=dev)
    ws = torch.empty(B, 3 * D, device=dev)
    tf = timeit(lambda: ops.attn_fwd(qkv, out, B=B, T=T, H=H, hd=hd, seq_off=off, lse=lse))
    tb = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, dqkv, B=B, T=T, H=H, hd=hd, seq_off=off, dbias=db, dbias_ws=ws))
    io_f = rows * D * 2 * 4
    io_b = rows * D * 2 * (3 + 1 + 1 + 3)
    print(f"attn {tag:14s} rows {rows:7d} Tmax {T:3d}: fwd {tf * 1e3:7.1f} us ({flops / tf / 1e9:6.1f} TF, {io_f / tf / 1e9:5.2f} TB/s)   "
          f"bwd {tb * 1e3:7.1f} us ({2.5 * flops / tb / 1e9:6.1f} TF, {io_b / tb / 1e9:5.2f} TB/s)")


attention("teacher", np.full(256, 200), 12, 64, T_dense=200)
attention("student ragged", enc_len, 12, 64)
attention("predictor ragged", dec_len, 12, 32)
long_dec = dec_len.copy()
long_dec[0] = 150                  # one sequence beyond 128 tokens moves the whole launch to the 129..192-token variants
attention("predictor, Tmax 150", long_dec, 12, 32)

for tag, M, D in (("teacher", 51200, 768), ("student ragged", int(enc_len.sum()), 768), ("predictor ragged", int(dec_len.sum()), 384)):
    x = torch.randn(M, D, device=dev)
    r = torch.randn(M, D, device=dev).to(bf)
    g = torch.ones(D, device=dev)
    b = torch.zeros(D, device=dev)
    y = torch.empty(M, D, device=dev)
    yb = torch.empty(M, D, device=dev, dtype=bf)
    mean = torch.empty(M, device=dev)
    rstd = torch.empty(M, device=dev)
    dy = torch.randn(M, D, device=dev)
    ds = torch.empty(M, D, device=dev)
    dsb = torch.empty(M, D, device=dev, dtype=bf)
    dg = torch.zeros(D, device=dev)
    dbt = torch.zeros(D, device=dev)
    dbi = torch.zeros(D, device=dev)
    wsx = torch.empty(1536 * 3 * D, device=dev)
    tf = timeit(lambda: ops.layernorm_fwd(x, g, b, M=M, D=D, eps=1e-6, r=r, y_f32=y, y_bf16=yb, mean=mean, rstd=rstd))
    tb = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, M=M, D=D, r=r, ds_f32=ds, ds_bf16=dsb, dgamma=dg, dbeta=dbt, dbias=dbi,
                                          workspace=wsx))
    t0 = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, M=M, D=D, r=r, ds_f32=ds, ds_bf16=dsb))
    bf_bytes = M * D * (4 + 2 + 4 + 2)
    bb_bytes = M * D * (4 + 4 + 2 + 4 + 2)
    print(f"LN   {tag:16s} M {M:7d} D {D}: fwd {tf * 1e3:7.1f} us ({bf_bytes / tf / 1e9:5.2f} TB/s)   bwd {tb * 1e3:7.1f} us "
          f"({bb_bytes / tb / 1e9:5.2f} TB/s; without parameter gradients {t0 * 1e3:7.1f} us)")

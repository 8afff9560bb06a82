#!/usr/bin/env python3
"""Attention / LayerNorm micro-benchmark at the step's shapes (GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavjepa_amd import ops  # noqa: E402

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


for tag, B, H, hd in (("enc", 256, 12, 64), ("dec", 1024, 12, 32)):
    T, D = 200, H * hd
    qkv = torch.randn(B, T, 3 * D, device=dev).to(bf)
    mask = (torch.rand(B, T, device=dev) < 0.6)
    mask[:, 0] = False
    m8 = mask.to(torch.uint8).contiguous()
    out = torch.empty(B, T, D, device=dev, dtype=bf)
    lse = torch.empty(B, H, T, device=dev)
    dout = torch.randn(B, T, D, device=dev).to(bf)
    dqkv = torch.empty_like(qkv)
    db = torch.zeros(3 * D, device=dev); ws = torch.empty(B, 3 * D, device=dev)
    tf = timeit(lambda: ops.attn_fwd(qkv, out, B=B, T=T, H=H, hd=hd, key_mask=m8, lse=lse))
    tb = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, dqkv, B=B, T=T, H=H, hd=hd, key_mask=m8, dbias=db, dbias_ws=ws))
    fl = 4.0 * T * T * D * B
    print(f"attn {tag}: fwd {tf * 1000:7.1f} us ({fl / tf / 1e9:6.1f} TF)   bwd {tb * 1000:7.1f} us ({2.5 * fl / tb / 1e9:6.1f} TF)")

for tag, M, D in (("enc", 51200, 768), ("dec", 204800, 384)):
    x = torch.randn(M, D, device=dev)
    r = torch.randn(M, D, device=dev).to(bf)
    g = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
    y = torch.empty(M, D, device=dev); yb = torch.empty(M, D, device=dev, dtype=bf)
    mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    dy = torch.randn(M, D, device=dev); ds = torch.empty(M, D, device=dev); dsb = torch.empty(M, D, device=dev, dtype=bf)
    dg = torch.zeros(D, device=dev); dbt = torch.zeros(D, device=dev); dbi = torch.zeros(D, device=dev)
    tf = timeit(lambda: ops.layernorm_fwd(x, g, b, M=M, D=D, eps=1e-6, r=r, y_f32=y, y_bf16=yb, mean=mean, rstd=rstd))
    tb = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, M=M, D=D, r=r, ds_f32=ds, ds_bf16=dsb, dgamma=dg, dbeta=dbt, dbias=dbi))
    bf_bytes = M * D * (4 + 2 + 4 + 2); bb_bytes = M * D * (4 + 4 + 2 + 4 + 2)
    print(f"LN   {tag}: fwd {tf * 1000:7.1f} us ({bf_bytes / tf / 1e9:5.2f} TB/s)   bwd {tb * 1000:7.1f} us ({bb_bytes / tb / 1e9:5.2f} TB/s)")

# --- cost of the contended parameter-gradient atomics
for tag, M, D in (("enc", 51200, 768), ("dec", 204800, 384)):
    x = torch.randn(M, D, device=dev); r = torch.randn(M, D, device=dev).to(bf)
    g = torch.ones(D, device=dev)
    mean = torch.zeros(M, device=dev); rstd = torch.ones(M, device=dev)
    dy = torch.randn(M, D, device=dev); ds = torch.empty(M, D, device=dev); dsb = torch.empty(M, D, device=dev, dtype=bf)
    dg = torch.zeros(D, device=dev); dbt = torch.zeros(D, device=dev); dbi = torch.zeros(D, device=dev); wsx = torch.empty(1536 * 3 * D, device=dev)
    t1 = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, M=M, D=D, r=r, ds_f32=ds, ds_bf16=dsb, dgamma=dg, dbeta=dbt, dbias=dbi, workspace=wsx))
    t0 = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, M=M, D=D, r=r, ds_f32=ds, ds_bf16=dsb))
    print(f"LN bwd {tag}: with param-grad atomics {t1 * 1000:7.1f} us, without {t0 * 1000:7.1f} us")
for tag, B, H, hd in (("enc", 256, 12, 64), ("dec", 1024, 12, 32)):
    T, D = 200, H * hd
    qkv = torch.randn(B, T, 3 * D, device=dev).to(bf)
    out = torch.randn(B, T, D, device=dev).to(bf); lse = torch.zeros(B, H, T, device=dev)
    dout = torch.randn(B, T, D, device=dev).to(bf); dqkv = torch.empty_like(qkv); db = torch.zeros(3 * D, device=dev); ws = torch.empty(B, 3 * D, device=dev)
    t1 = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, dqkv, B=B, T=T, H=H, hd=hd, dbias=db, dbias_ws=ws))
    t0 = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, dqkv, B=B, T=T, H=H, hd=hd))
    print(f"attn bwd {tag}: with dbias atomics {t1 * 1000:7.1f} us, without {t0 * 1000:7.1f} us")

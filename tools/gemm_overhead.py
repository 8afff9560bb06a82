#!/usr/bin/env python3
"""Per-tile fixed overhead of the GEMM: time vs K for fixed M, N (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavjepa_amd import ops
dev = torch.device("cuda:0"); bf = torch.bfloat16
M, N = 51200, 2304
for epi, name in ((ops.EPI_BF16, "BF16+bias"), (ops.EPI_BF16, "BF16 nobias")):
    for K in (32, 64, 128, 256, 768, 1536):
        A = torch.randn(M, K, device=dev).to(bf); W = torch.randn(N, K, device=dev).to(bf)
        C = torch.empty(M, N, device=dev, dtype=bf); bias = torch.randn(N, device=dev) if "nobias" not in name else None
        ts = []
        for r in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.gemm(A, W, C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias, epilogue=epi)
            e1.record(); torch.cuda.synchronize()
            if r: ts.append(e0.elapsed_time(e1) / 3)
        t = sorted(ts)[len(ts) // 2]
        print(f"{name:12s} K={K:5d}  {t * 1000:8.1f} us")

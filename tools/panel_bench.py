#!/usr/bin/env python3
"""The predictor's N = 384 GEMM shapes on the persistent 256 x 256 kernel (variant 4) and on the row-panel kernel (variant 5), operands not
cache-resident (a 768-MB fill between launches), interleaved in one process:  python tools/panel_bench.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavjepa_amd import ops  # noqa: E402

dev, bf = torch.device("cuda:0"), torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 87421
junk = torch.empty(768 * 1024 * 1024 // 4, device=dev)
N = 384
for K in (384, 1152, 1536):
    A = torch.randn(M, K, device=dev).to(bf)
    W = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    bias = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev, dtype=bf)
    res = {4: [], 5: []}
    for r in range(14):
        for v in (4, 5):
            junk.fill_(float(r))
            e0, e1 = ops.TimingEvent(), ops.TimingEvent()
            s = torch.cuda.current_stream().cuda_stream
            e0.record(s)
            ops.gemm(A, W, C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias, schedule=v)
            e1.record(s)
            torch.cuda.synchronize()
            if r > 1:
                res[v].append(e0.elapsed_time(e1) * 1e3)
    warm = {}
    for v in (4, 5):
        ts = []
        for r in range(12):
            e0, e1 = ops.TimingEvent(), ops.TimingEvent()
            s = torch.cuda.current_stream().cuda_stream
            e0.record(s)
            ops.gemm(A, W, C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias, schedule=v)
            e1.record(s)
            torch.cuda.synchronize()
            if r > 1:
                ts.append(e0.elapsed_time(e1) * 1e3)
        warm[v] = sorted(ts)[len(ts) // 2]
    med = {v: sorted(t)[len(t) // 2] for v, t in res.items()}
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: persistent {med[4]:7.1f} us ({fl / med[4] / 1e6:6.0f} TFLOP/s)   row panels {med[5]:7.1f} us ({fl / med[5] / 1e6:6.0f} TFLOP/s)"
          f"   warm {warm[4]:.1f} / {warm[5]:.1f} us", flush=True)

#!/usr/bin/env python3
"""Calibration only (never used by the product path): what the vendor BLAS (torch.matmul -> hipBLASLt / rocBLAS) reaches on
the step's GEMM shapes, next to this library's kernel, same process, HIP-event timing."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavjepa_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16
SHAPES = [("teacher qkv", 51200, 2304, 768), ("teacher lin2", 51200, 768, 3072), ("teacher out", 51200, 768, 768),
          ("teacher lin1", 51200, 3072, 768), ("pred qkv", 86691, 1152, 384), ("pred lin1", 86691, 1536, 384),
          ("pred lin2", 86691, 384, 1536), ("pred out", 86691, 384, 384), ("stud qkv", 10048, 2304, 768),
          ("stud lin2", 10048, 768, 3072), ("square 8192", 8192, 8192, 8192)]


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


for tag, M, N, K in SHAPES:
    A = torch.randn(M, K, device=dev).to(bf)
    W = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    bias = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev, dtype=bf)
    t_lib = timeit(lambda: torch.nn.functional.linear(A, W))
    t_ours = timeit(lambda: ops.gemm(A, W, C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias))
    fl = 2.0 * M * N * K
    print(f"{tag:14s} M={M:6d} N={N:5d} K={K:5d}   vendor BLAS {t_lib * 1e3:8.1f} us {fl / t_lib / 1e9:7.1f} TF   "
          f"this library (+bias) {t_ours * 1e3:8.1f} us {fl / t_ours / 1e9:7.1f} TF")

#!/usr/bin/env python3
"""MX fp8 GEMM (wj_gemm_mxfp8) next to the bf16 GEMM on the transformer forward shapes (GPU box): TFLOP/s, interleaved rounds."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavjepa_amd import ops  # noqa: E402

dev, bf = torch.device("cuda:0"), torch.bfloat16
SHAPES = [("teacher qkv 2s", 51200, 2304, 768), ("teacher out 2s", 51200, 768, 768), ("teacher lin1 2s", 51200, 3072, 768),
          ("teacher lin2 2s", 51200, 768, 3072), ("teacher qkv 4s", 102400, 2304, 768), ("teacher lin2 4s", 102400, 768, 3072),
          ("pred lin1", 86317, 1536, 384), ("pred lin2", 86317, 384, 1536), ("square 8192", 8192, 8192, 8192)]
for tag, M, N, K in SHAPES:
    if K % 256:
        print(f"{tag:18s} K={K}: not a multiple of 256, bf16 only")
    x = torch.randn(M, K, device=dev).to(bf)
    w = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    bias = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev, dtype=bf)
    qx, qw = torch.empty(M, K, dtype=torch.uint8, device=dev), torch.empty(N, K, dtype=torch.uint8, device=dev)
    sx = torch.zeros(ops.fp8_scale_dwords(M, K), dtype=torch.int32, device=dev)
    sw = torch.zeros(ops.fp8_scale_dwords(N, K), dtype=torch.int32, device=dev)
    fns = {"bf16": lambda: ops.gemm(x, w, C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias),
           "quantize A": lambda: ops.quantize_mxfp8(x, qx, sx, M=M, K=K, ldx=K, ldq=K, ld_scale=M)}
    ops.quantize_mxfp8(w, qw, sw, M=N, K=K, ldx=K, ldq=K, ld_scale=N)
    if K % 256 == 0:
        fns["mxfp8"] = lambda: ops.gemm_mxfp8(qx, qw, sx, sw, C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, ld_scale_a=M, ld_scale_b=N, bias=bias)
    ts = {k: [] for k in fns}
    for r in range(6):
        for k, fn in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts[k].append(e0.elapsed_time(e1) / 3)
    fl = 2.0 * M * N * K
    med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
    line = f"{tag:18s} M={M:6d} N={N:5d} K={K:5d}: bf16 {med['bf16'] * 1e3:7.1f} us {fl / med['bf16'] / 1e9:7.1f} TF"
    if "mxfp8" in med:
        line += f" | mxfp8 {med['mxfp8'] * 1e3:7.1f} us {fl / med['mxfp8'] / 1e9:7.1f} TF"
    line += f" | quantize A {med['quantize A'] * 1e3:6.1f} us ({3.03 * M * K / med['quantize A'] / 1e9:.2f} TB/s)"
    print(line)

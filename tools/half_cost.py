#!/usr/bin/env python3
"""What does a half-width work item of the persistent GEMM cost?  N = 256 (one full item per 256-row panel), 384 (full + half), 512 (two
full) at the predictor's row count, cold operands, variant 4: cost(half) / cost(full) = (t384 - t256) / (t512 - t256)."""
import os
os.environ.setdefault("WAVJEPA_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wavjepa_amd", "lib", "libwavjepa_hip_lab.so"))  # laboratory build: honours the WJ_* A/B switches, exports the stamp reader
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("WJ_PERSIST_MIN_TILES", "1")
from wavjepa_amd import ops  # noqa: E402

dev, bf = torch.device("cuda:0"), torch.bfloat16
M = 86317
junk = torch.empty(768 * 1024 * 1024 // 4, device=dev)
ops.gemm_set_variant(4)
for K in (384, 1152, 1536):
    res = {}
    for N in (256, 384, 512):
        A = torch.randn(M, K, device=dev).to(bf)
        W = (torch.randn(N, K, device=dev) * 0.05).to(bf)
        C = torch.empty(M, N, device=dev, dtype=bf)
        ts = []
        for r in range(12):
            junk.fill_(float(r))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.gemm(A, W, C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N)
            e1.record()
            torch.cuda.synchronize()
            if r > 1:
                ts.append(e0.elapsed_time(e1) * 1e3)
        res[N] = sorted(ts)[len(ts) // 2]
    full = res[512] - res[256]
    print(f"K={K}: N=256 {res[256]:.1f} us, N=384 {res[384]:.1f} us, N=512 {res[512]:.1f} us -> a second full item per panel costs {full:.1f} us, "
          f"a half item {res[384] - res[256]:.1f} us = {(res[384] - res[256]) / full:.2f} of it")

#!/usr/bin/env python3
"""Timing of the deferred-epilogue GEMM (variant 6) against the persistent kernel (variant 4) on the step's GELU shapes, cold operands
(a 512-MB fill between launches); WJ_PDE_DIAG=1 (laboratory library) = the raw K loop of variant 6 (no GELU, no stores)."""
import os, sys
os.environ.setdefault("WAVJEPA_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wavjepa_amd", "lib", "libwavjepa_hip_lab.so"))
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavjepa_amd import ops
dev = torch.device("cuda:0"); bf = torch.bfloat16
SH = [(51200, 3072, 768, ops.EPI_BIAS_GELU), (86605, 1536, 384, ops.EPI_BIAS_GELU2), (10002, 3072, 768, ops.EPI_BIAS_GELU2),
      (823296, 512, 1536, ops.EPI_CONV_GELU), (51200, 3072, 3072, ops.EPI_BIAS_GELU)]
flush = torch.empty(128 * 1024 * 1024, device=dev)
for (M, N, K, epi) in SH:
    A = torch.randn(M, K, device=dev).to(bf); W = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    C = torch.empty(M, N, device=dev, dtype=bf); C2 = torch.empty(M, N, device=dev, dtype=bf) if epi != ops.EPI_BIAS_GELU else None
    kw = dict(M=M, N=N, K=K, lda=K, ldb=K, ldc=N, epilogue=epi)
    if epi == ops.EPI_CONV_GELU: kw.update(seg_rows=402, seg_valid=400)
    else: kw["bias"] = torch.randn(N, device=dev)
    res = {}
    for v in (4, 6):
        ts = []
        for r in range(6):
            flush.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.gemm(A, W, C, C2=C2, schedule=v, **kw); e1.record(); torch.cuda.synchronize()
            if r: ts.append(e0.elapsed_time(e1) * 1e3)
        res[v] = sorted(ts)[len(ts) // 2]
    it4, it6 = -(-M // 256) * (N // 256), -(-M // 128) * (N // 256)
    r4, r6 = -(-it4 // 256), -(-it6 // 256)
    print(f"M={M} N={N} K={K} epi={epi} diag={os.environ.get('WJ_PDE_DIAG','0')}: v4 {res[4]:7.1f} us ({res[4]/r4:5.2f} us/item over {r4} rounds) | "
          f"v6 {res[6]:7.1f} us ({res[6]/r6:5.2f} us/item over {r6} rounds, {res[6]/r6/(K//64):.3f} us per K tile)", flush=True)

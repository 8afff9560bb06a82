#!/usr/bin/env python3
"""A GEMM as the step sees it (GPU box): operands NOT resident in the Infinity Cache.  Every timed launch follows a 768 MB write to
an unrelated buffer; compared with back-to-back (warm) launches.  usage: gemm_cold.py [variant ...]"""
import os
os.environ.setdefault("WAVJEPA_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wavjepa_amd", "lib", "libwavjepa_hip_lab.so"))  # laboratory build: honours the WJ_* A/B switches, exports the stamp reader
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavjepa_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16
SHAPES = [(51200, 2304, 768, ops.EPI_BF16), (51200, 768, 3072, ops.EPI_BF16), (51200, 768, 768, ops.EPI_BF16),
          (51200, 3072, 768, ops.EPI_BIAS_GELU), (86317, 1536, 384, ops.EPI_BIAS_GELU2), (86317, 1152, 384, ops.EPI_BF16),
          (86317, 384, 1536, ops.EPI_BF16), (86317, 384, 384, ops.EPI_BF16)]
SHAPES += [(86317, 1536, 384, ops.EPI_MUL_GELU_GRAD), (9907, 3072, 768, ops.EPI_MUL_GELU_GRAD), (9907, 3072, 768, ops.EPI_BIAS_GELU2),
           (86317, 384, 1152, ops.EPI_BF16)]
if os.environ.get("WJ_COLD_EPI"):                  # e.g. "1,2": only these epilogues
    SHAPES = [sh for sh in SHAPES if str(sh[3]) in os.environ["WJ_COLD_EPI"].split(",")]
variants = [int(v) for v in sys.argv[1:]] or [3, 4]
junk = torch.empty(768 * 1024 * 1024 // 4, device=dev)
for (M, N, K, epi) in SHAPES:
    A = torch.randn(M, K, device=dev).to(bf)
    W = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    C = torch.empty(M, N, device=dev, dtype=bf)
    C2 = torch.empty(M, N, device=dev, dtype=bf) if epi == ops.EPI_BIAS_GELU2 else None
    kw = dict(M=M, N=N, K=K, lda=K, ldb=K, ldc=N, epilogue=epi, bias=torch.randn(N, device=dev))
    if C2 is not None:
        kw["C2"] = C2
    if epi == ops.EPI_MUL_GELU_GRAD:
        kw.pop("bias")
        kw.update(aux=torch.rand(M, N, device=dev).to(bf), colsum=torch.zeros(N, device=dev))
    out = []
    ts = {(v, mode): [] for v in variants for mode in ("warm", "cold")}
    for r in range(12):                      # the variants interleaved, so that a clock drift hits all of them alike
        for mode in ("warm", "cold"):
            for v in variants:
                ops.gemm_set_variant(v)
                if mode == "cold":
                    junk.fill_(float(r))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.gemm(A, W, C, **kw)
                e1.record()
                torch.cuda.synchronize()
                if r > 1:
                    ts[(v, mode)].append(e0.elapsed_time(e1) * 1e3)
    for v in variants:
        res = {mode: sorted(ts[(v, mode)])[len(ts[(v, mode)]) // 2] for mode in ("warm", "cold")}
        out.append(f"variant {v}: warm {res['warm']:7.1f} us, cold {res['cold']:7.1f} us")
    print(f"M={M} N={N} K={K} epi={epi}: " + " | ".join(out))
ops.gemm_set_variant(-1)

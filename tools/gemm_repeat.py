#!/usr/bin/env python3
"""Repeat screen for the persistent GEMM (GPU box): one set of operands, 150 launches into NaN-filled outputs, each compared bit for bit
with the first.  A tile the scheduler skipped stays NaN; a tile computed from half-landed operands differs.  (This caught the
mailbox write that could run ahead of the pull it publishes: about 1 launch in 40 skipped a tile.)"""
import os
os.environ.setdefault("WAVJEPA_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wavjepa_amd", "lib", "libwavjepa_hip_lab.so"))  # laboratory build: honours the WJ_* A/B switches, exports the stamp reader
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavjepa_amd import ops
dev = torch.device("cuda:0"); bf = torch.bfloat16
for (M, N, K) in [(51200, 2304, 768), (86317, 1152, 384), (86317, 384, 1536)]:
    torch.manual_seed(1)
    A = torch.randn(M, K, device=dev).to(bf); W = (torch.randn(N, K, device=dev) * 0.05).to(bf); bias = torch.randn(N, device=dev)
    kw = dict(M=M, N=N, K=K, lda=K, ldb=K, ldc=N, epilogue=ops.EPI_BF16, bias=bias)
    ops.gemm_set_variant(4)
    ref = torch.empty(M, N, device=dev, dtype=bf); ops.gemm(A, W, ref, **kw); torch.cuda.synchronize()
    nbad = 0
    for r in range(150):
        C = torch.full((M, N), float("nan"), device=dev, dtype=bf)
        ops.gemm(A, W, C, **kw); torch.cuda.synchronize()
        d = (C.view(torch.int16) != ref.view(torch.int16))
        if bool(d.any()):
            nbad += 1
            idx = d.nonzero()
            rows, cols = idx[:, 0], idx[:, 1]
            tiles = torch.unique(torch.stack([rows // 256, cols // 256], 1), dim=0)
            print(f"M={M} N={N} K={K} rep {r}: {idx.shape[0]} differ; tiles {tiles.tolist()[:6]} ({tiles.shape[0]} tiles); rows%256 {sorted(set((rows % 256).tolist()))[:40]} cols%256 min {int((cols%256).min())} max {int((cols%256).max())}; "
                  f"max abs diff {float((C.float()-ref.float()).abs().max()):.4g} nan {int(torch.isnan(C.float()).sum())}")
    print(f"M={M} N={N} K={K}: {nbad} of 150 repetitions differ from the first run")
    bad_total = bad_total + nbad if "bad_total" in dir() else nbad
ops.gemm_set_variant(-1)
sys.exit(1 if bad_total else 0)

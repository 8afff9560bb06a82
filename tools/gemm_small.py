#!/usr/bin/env python3
"""Which schedule for the student's ragged shapes (M ~ 10 k rows: 117 / 351 / 468 output tiles of 256 x 256 on 256 CUs)?  Row-form
variants 0 (256 x 128, two workgroups per CU), 2 (ping-pong), 3 (eight-phase), 4 (persistent, WJ_PERSIST_MIN_TILES=1) and the col-form B
dgrad of the same product, warm (back to back) and cold (a 768 MB write in between), interleaved in one process."""
import os
os.environ.setdefault("WAVJEPA_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wavjepa_amd", "lib", "libwavjepa_hip_lab.so"))  # laboratory build: honours the WJ_* A/B switches, exports the stamp reader
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavjepa_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 9907
SHAPES = [(M, 768, 3072), (M, 768, 2304), (M, 768, 768), (M, 2304, 768), (M, 384, 768)]
junk = torch.empty(768 * 1024 * 1024 // 4, device=dev)
for (m, N, K) in SHAPES:
    A = torch.randn(m, K, device=dev).to(bf)
    W = (torch.randn(N, K, device=dev) * 0.05).to(bf)        # row form [N][K]
    Wt = W.t().contiguous()                                  # col form [K][N]: the nn.Linear weight of the dgrad
    C = torch.empty(m, N, device=dev, dtype=bf)
    need = ops.workspace_bytes("wj_gemm_bf16", M=m, N=N, K=K, lda=K, ldb=K, ldc=N, epilogue=ops.EPI_BF16)
    os.environ.setdefault("WJ_PAIR_MIN_K", "256")           # (read once by the library: set before the first launch to try short K too)
    ws = torch.zeros(max(need, 256), dtype=torch.uint8, device=dev)
    cases = [("row v0", 0, False), ("row v2", 2, False), ("row v3", 3, False), ("row v4", 4, False), ("col v0", 0, True)]
    if need:
        cases.append(("row v3 pair", 3, False))
    ts = {(c[0], mode): [] for c in cases for mode in ("warm", "cold")}
    for r in range(14):
        for mode in ("warm", "cold"):
            for name, v, col in cases:
                ops.gemm_set_variant(v)
                if mode == "cold":
                    junk.fill_(float(r))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if col:
                    ops.gemm(A, Wt, C, M=m, N=N, K=K, lda=K, ldb=N, ldc=N, b_trans=1)
                else:
                    ops.gemm(A, W, C, M=m, N=N, K=K, lda=K, ldb=K, ldc=N, workspace=ws if name.endswith("pair") else None)
                e1.record()
                torch.cuda.synchronize()
                if r > 1:
                    ts[(name, mode)].append(e0.elapsed_time(e1) * 1e3)
    med = lambda x: sorted(x)[len(x) // 2]
    print(f"M={m} N={N} K={K}: " + " | ".join(f"{c[0]} {med(ts[(c[0], 'warm')]):6.1f}/{med(ts[(c[0], 'cold')]):6.1f}" for c in cases) + "  (warm/cold us)")
ops.gemm_set_variant(-1)

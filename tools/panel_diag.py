#!/usr/bin/env python3
"""What bounds the row-panel GEMM (and, by the same mechanism, the N = 384 shapes on any schedule)?  The laboratory build can leave out
one ingredient at a time (WJ_PANEL_DIAG bits: 1 no W pieces, 2 no A pieces, 4 no MFMAs, 8 no fragment reads, 16 no store traffic --
the instruction counts stay, so the counted waits are unchanged; results are wrong by design): cold launches, interleaved."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("WAVJEPA_HIP_LIB", os.path.join(ROOT, "wavjepa_amd", "lib", "libwavjepa_hip_lab.so"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from wavjepa_amd import ops  # noqa: E402

dev, bf = torch.device("cuda:0"), torch.bfloat16
M, N = 87421, 384
junk = torch.empty(768 * 1024 * 1024 // 4, device=dev)
MODES = [(0, "everything"), (32, "everything, K order rotated per workgroup"), (33, "rotated, no W pieces"), (34, "rotated, no A pieces"), (16, "no store traffic"), (4, "no MFMAs"), (8, "no fragment reads"), (12, "no MFMAs, no fragment reads"),
         (1, "no W pieces (L2 -> LDS)"), (2, "no A pieces (HBM -> LDS)"), (3, "no pieces at all"), (31, "barriers and waits only")]
for K in (384, 1536):
    A = torch.randn(M, K, device=dev).to(bf)
    W = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    C = torch.empty(M, N, device=dev, dtype=bf)
    res = {m: [] for m, _ in MODES}
    for r in range(9):
        for m, _ in MODES:
            os.environ["WJ_PANEL_DIAG"] = str(m)
            junk.fill_(float(r))
            e0, e1 = ops.TimingEvent(), ops.TimingEvent()
            s = torch.cuda.current_stream().cuda_stream
            e0.record(s)
            ops.gemm(A, W, C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, schedule=5)
            e1.record(s)
            torch.cuda.synchronize()
            if r > 1:
                res[m].append(e0.elapsed_time(e1) * 1e3)
    os.environ["WJ_PANEL_DIAG"] = "32"
    ops.gemm(A, W, C, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, schedule=5)
    torch.cuda.synchronize()
    ref = A[:4096].float() @ W.float().t()
    print(f"K={K}: rotated K order, relative error of the first 4096 rows against fp32 math: {float((C[:4096].float() - ref).norm() / ref.norm()):.2e}", flush=True)
    steps = 3 * (K // 64) * 3                     # 2.67 rounds of items -> three items per workgroup
    for m, name in MODES:
        t = sorted(res[m])[len(res[m]) // 2]
        print(f"K={K:5d} diag {m:2d} {name:36s} {t:7.1f} us   {t / steps * 1e3:6.0f} ns per step", flush=True)
os.environ["WJ_PANEL_DIAG"] = "0"

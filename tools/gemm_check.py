#!/usr/bin/env python3
"""Race screen + A/B for a GEMM schedule variant (run on the GPU box): every listed shape is computed with the reference
variant and with the variant under test, many times, on fresh random operands; outputs must agree BIT for bit (all schedules add
the k index in the same order) on every repetition.  Then interleaved timing rounds of both variants in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavjepa_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16
SHAPES = [  # M, N, K, epilogue
    (51200, 2304, 768, ops.EPI_BF16), (51200, 768, 768, ops.EPI_BF16), (51200, 768, 3072, ops.EPI_BF16),
    (51200, 3072, 768, ops.EPI_BIAS_GELU), (9907, 2304, 768, ops.EPI_BF16), (9907, 768, 3072, ops.EPI_BF16),
    (9907, 3072, 768, ops.EPI_BIAS_GELU2), (86317, 1536, 384, ops.EPI_BIAS_GELU2), (86317, 1152, 384, ops.EPI_BF16),
    (86317, 384, 1536, ops.EPI_BF16), (86317, 384, 384, ops.EPI_BF16), (8192, 8192, 8192, ops.EPI_BF16),
    (300, 256, 128, ops.EPI_BF16), (257, 512, 256, ops.EPI_BF16), (1000, 264, 384, ops.EPI_BF16),
]
ref_v, new_v = int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 3
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6


def run(v, A, W, C, kw):
    ops.gemm_set_variant(v)
    ops.gemm(A, W, C, **kw)


bad = 0
for (M, N, K, epi) in SHAPES:
    kw = dict(M=M, N=N, K=K, lda=K, ldb=K, ldc=N, epilogue=epi)
    bias = torch.randn(N, device=dev)
    kw["bias"] = bias
    C2a = C2b = None
    if epi == ops.EPI_BIAS_GELU2:
        C2a, C2b = torch.empty(M, N, device=dev, dtype=bf), torch.empty(M, N, device=dev, dtype=bf)
    worst = 0
    for r in range(reps):
        A = torch.randn(M, K, device=dev).to(bf)
        W = (torch.randn(N, K, device=dev) * 0.05).to(bf)
        Ca = torch.full((M, N), float("nan"), device=dev, dtype=bf)
        Cb = torch.full((M, N), float("nan"), device=dev, dtype=bf)
        run(ref_v, A, W, Ca, dict(kw, **({"C2": C2a} if C2a is not None else {})))
        run(new_v, A, W, Cb, dict(kw, **({"C2": C2b} if C2b is not None else {})))
        torch.cuda.synchronize()
        same = torch.equal(Ca.view(torch.int16), Cb.view(torch.int16)) and (C2a is None or torch.equal(C2a.view(torch.int16), C2b.view(torch.int16)))
        if not same:
            worst += 1
            d = (Ca.float() - Cb.float()).abs()
            print(f"  MISMATCH M={M} N={N} K={K} epi={epi} rep={r}: {int((d > 0).sum())} elements differ, max {float(d.max()):.4g}, "
                  f"nan {int(torch.isnan(Cb.float()).sum())}")
    if r == reps - 1 and M <= 10000:   # fp32 torch reference on the last operands (small shapes)
        ref = A.float() @ W.float().t() + bias
        if epi in (ops.EPI_BIAS_GELU, ops.EPI_BIAS_GELU2):
            ref = torch.nn.functional.gelu(ref.to(bf).float())
            got = (C2b if epi == ops.EPI_BIAS_GELU2 else Cb).float()
        else:
            got = Cb.float()
        print(f"  vs fp32 torch: rel err {float((got - ref).norm() / ref.norm()):.3e}")
    bad += worst
    # timing, interleaved
    ts = {ref_v: [], new_v: []}
    for r in range(7):
        for v in (ref_v, new_v):
            C = Ca if v == ref_v else Cb
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ops.gemm_set_variant(v)
            e0.record()
            for _ in range(3):
                ops.gemm(A, W, C, **dict(kw, **({"C2": C2a} if C2a is not None else {})))
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts[v].append(e0.elapsed_time(e1) / 3)
    fl = 2.0 * M * N * K
    ta, tb = sorted(ts[ref_v])[3], sorted(ts[new_v])[3]
    print(f"M={M:6d} N={N:5d} K={K:5d} epi={epi}: variant {ref_v} {ta * 1e3:7.1f} us {fl / ta / 1e9:7.1f} TF | variant {new_v} {tb * 1e3:7.1f} us "
          f"{fl / tb / 1e9:7.1f} TF | {'OK' if worst == 0 else 'MISMATCH x%d' % worst}")
ops.gemm_set_variant(-1)
print("mismatching repetitions:", bad)
sys.exit(1 if bad else 0)

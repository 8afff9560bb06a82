#!/usr/bin/env python3
"""Weight-gradient GEMMs of one transformer layer: four split-K launches (wj_gemm_bf16, EPI_ATOMIC_F32) against one grouped launch
(wj_wgrad_grouped), student (d=768, ~10 k context rows) and predictor (d=384, ~88 k visible rows) shapes.  GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavjepa_amd import ops  # noqa: E402

dev, bf = torch.device("cuda:0"), torch.bfloat16


def layer(d, m):
    dims = [(3 * d, d), (d, d), (4 * d, d), (d, 4 * d)]
    return [(torch.randn(m, n, device=dev).to(bf), torch.randn(m, k, device=dev).to(bf), torch.zeros(n, k, device=dev), n, k, m) for n, k in dims]


def timeit(fn, reps=3, rounds=6):
    ts = []
    for r in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) / reps)
    return sorted(ts)[len(ts) // 2]


for tag, d, m, nl in (("student", 768, 10300, 2), ("predictor", 384, 87916, 1), ("teacher-sized dense student", 768, 51200, 1)):
    layers = [layer(d, m) for _ in range(nl)]
    fl = sum(2.0 * p[3] * p[4] * p[5] for L in layers for p in L)

    def separate():
        for L in layers:
            for dy, x, gw, n, k, mm in L:
                ops.gemm(dy, x, gw, M=n, N=k, K=mm, lda=n, ldb=k, ldc=k, a_trans=1, b_trans=1, epilogue=ops.EPI_ATOMIC_F32,
                         split_k=ops.pick_split_k(n, k, mm))

    def per_layer():
        for L in layers:
            ops.wgrad_grouped(L)

    def all_layers():
        ops.wgrad_grouped([p for L in layers for p in L])

    t0, t1 = timeit(separate), timeit(per_layer)
    line = f"{tag:28s} d={d} rows={m} layers={nl}: separate {t0 * 1e3:7.1f} us {fl / t0 / 1e9:6.1f} TF | grouped per layer {t1 * 1e3:7.1f} us {fl / t1 / 1e9:6.1f} TF"
    if nl > 1:
        t2 = timeit(all_layers)
        line += f" | grouped {nl} layers {t2 * 1e3:7.1f} us {fl / t2 / 1e9:6.1f} TF"
    print(line)

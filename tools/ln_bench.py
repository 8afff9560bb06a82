#!/usr/bin/env python3
"""LayerNorm forward / backward launches of the step in isolation: us per launch and GB/s of algorithmic bytes, cold operands
(a ring of buffer sets larger than the L2 + MALL so that no launch finds its inputs cached).

  python tools/ln_bench.py                      # the step's shapes: teacher 51200 x 768, student 10045 x 768, predictor 86781 x 384
  WJ_LN_BWD_ONE_PASS_ROWS=0 python tools/ln_bench.py   # backward with four passes per wave whatever M is

Environment switches are read by the library at its first launch, so each configuration is its own process."""
from __future__ import annotations

import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavjepa_amd import ops  # noqa: E402

SHAPES = [("teacher", 51200, 768), ("student", 10045, 768), ("predictor", 86781, 384), ("predictor-last", 46601, 384)]


def main():
    dev = torch.device("cuda", 0)
    reps = int(os.environ.get("LN_BENCH_REPS", "40"))
    for name, M, D in SHAPES:
        n_sets = max(3, int(1.2e9 // (M * D * 18)))
        sets = []
        for i in range(n_sets):
            g = torch.Generator(device=dev).manual_seed(i)
            sets.append(dict(
                x=torch.randn(M, D, device=dev, generator=g), r=torch.randn(M, D, device=dev, generator=g).to(torch.bfloat16),
                dy=torch.randn(M, D, device=dev, generator=g), dy2=torch.randn(M, D, device=dev, generator=g).to(torch.bfloat16),
                y=torch.empty(M, D, device=dev), yb=torch.empty(M, D, device=dev, dtype=torch.bfloat16),
                ds=torch.empty(M, D, device=dev), dsb=torch.empty(M, D, device=dev, dtype=torch.bfloat16),
                mean=torch.empty(M, device=dev), rstd=torch.empty(M, device=dev)))
        gamma, beta = torch.randn(D, device=dev), torch.randn(D, device=dev)
        dgam, dbet, dbias = (torch.zeros(D, device=dev) for _ in range(3))
        ws = torch.empty(ops.workspace_bytes("wj_layernorm_bwd", D=D) // 4, device=dev)

        def fwd(s):
            ops.layernorm_fwd(s["x"], gamma, beta, M=M, D=D, eps=1e-5, r=s["r"], y_f32=s["y"], y_bf16=s["yb"], mean=s["mean"], rstd=s["rstd"])

        def bwd(s):
            ops.layernorm_bwd(s["dy"], s["x"], gamma, s["mean"], s["rstd"], M=M, D=D, r=s["r"], dy2=s["dy2"], dy2_is_bf16=True,
                              ds_f32=s["ds"], ds_bf16=s["dsb"], workspace=ws)

        for label, fn, bytes_per in (("fwd", fwd, 12), ("bwd", bwd, 18)):
            for s in sets:
                fwd(s)
                fn(s)
            torch.cuda.synchronize()
            times = []
            for i in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn(sets[i % n_sets])
                e1.record()
                times.append((e0, e1))
            torch.cuda.synchronize()
            us = sorted(a.elapsed_time(b) * 1e3 for a, b in times)
            med = us[len(us) // 2]
            print(f"{name:15s} {label} M={M:6d} D={D:4d}  median {med:7.1f} us  min {us[0]:7.1f}  {M * D * bytes_per / med / 1e6:6.2f} TB/s "
                  f"({M * D * bytes_per / 1e6:.0f} MB, partial rows {ops.ln_bwd_partial_rows(M, D) if label == 'bwd' else '-'})", flush=True)


if __name__ == "__main__":
    main()

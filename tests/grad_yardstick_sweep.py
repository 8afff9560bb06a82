"""Gradient yardstick over UNPINNED mask draws (GPU box; evidence for the bounds in tests/parity_yardstick.py).

For every draw: the HIP path's parameter gradients, the oracle's bf16-flow gradients and the oracle's fp32 gradients on the same
weights / clips / masks; per parameter group d_hip = d(HIP, fp32), d_orc = d(oracle-bf16, fp32), pair = d(HIP, oracle-bf16) (relative L2).
Two cases: the 2-channel ConvChannelFeatureExtractor model of test_forward_backward_parity_channel_extractor (3 clips x 2 x 99 tokens:
the per-stack conv gradient whose fixed bound failed on some draws in round 4) and the BASE model at 64 clips.

    python3 tests/grad_yardstick_sweep.py [--draws 16] [--base-draws 4] > profiles/r05_grad_yardstick.txt
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from oracle import jepa_oracle as J                      # noqa: E402  (test infrastructure: the checker, not the product)
from tests import parity_yardstick as Y                  # noqa: E402
from tests import test_jepa_gpu as T                     # noqa: E402
import synth                                             # noqa: E402


def one_draw(m, P, cfg, audio, ctx, tgt, vis, group_of):
    dev = T.dev()
    names = J.trainable_names(P)
    for k in names:
        P[k].grad = None
        P[k].requires_grad_(True)
    m.zero_grad(set_to_none=True)
    out = m(audio, ctx, tgt, vis)
    out["loss"].backward()
    ref = J.jepa_forward(P, audio, ctx.to(dev), tgt.to(dev), vis.to(dev), mode="bf16", **T.oracle_kw(cfg))
    ref["loss"].backward()
    got = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    gbf = {k: P[k].grad for k in names}
    ref32, g32 = Y.oracle_fp32_grads(J, P, audio, ctx.to(dev), tgt.to(dev), vis.to(dev), names, **T.oracle_kw(cfg))
    table = Y.grad_yardstick(got, gbf, g32, names, group_of)
    return table, float(out["loss"].detach()), float(ref["loss"].detach()), float(ref32["loss"].detach())


def report(title, rows):
    """rows: list of (draw label, table).  Prints per group the range of d_hip, d_orc, pair, ratio over the draws + the worst draw."""
    print(f"\n== {title}: {len(rows)} draws")
    groups = list(rows[0][1].keys())
    print(f"{'group':34s} {'d_orc (oracle-bf16 vs fp32)':>30s} {'d_hip (HIP vs fp32)':>26s} {'pair (HIP vs oracle-bf16)':>28s} {'ratio d_hip/d_orc':>22s} {'bound ok':>9s}")
    for g in groups:
        col = {k: np.array([t[g][k] for _, t in rows]) for k in ("d_orc", "d_hip", "pair", "ratio")}
        ok = all(Y.grad_bound_ok(g, t[g]) for _, t in rows)
        f = lambda a: f"{a.min():.5f} .. {a.max():.5f}"
        print(f"{g:34s} {f(col['d_orc']):>30s} {f(col['d_hip']):>26s} {f(col['pair']):>28s} {col['ratio'].min():9.3f} .. {col['ratio'].max():6.3f} {str(ok):>9s}")
    return all(Y.grad_bound_ok(g, t[g]) for _, t in rows for g in groups)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--draws", type=int, default=16)
    ap.add_argument("--base-draws", type=int, default=4)
    a = ap.parse_args()
    from wavjepa_amd.masking import TimeInverseBlockMasker
    print(f"# gradient yardstick, bound: d_hip < {Y.GRAD_FACTOR} x d_orc + {Y.GRAD_EPS}  (relative L2 per parameter group; masks from OS entropy; "
          f"groups {Y.GRAD_FACTOR_BY_PREFIX}: their own factor)")
    all_ok = True
    for stacks in ("own", "shared"):
        m, P = T.build(T.SMALL, seconds=1.0, tokens=198, in_channels=2, channel_stacks=stacks)
        audio = torch.from_numpy(synth.synth_audio(3, 2, 16000, seed=31)).to(torch.bfloat16).to(T.dev())
        rows = []
        for d in range(a.draws):
            ctx, tgt, vis = TimeInverseBlockMasker(4, 0.65, 10, 0.25, 10, 0.1, channel_based_masking=True, channel_major=True)(
                batch_size=3, n_times=198, in_channels=2)
            table, lo, lb, l32 = one_draw(m, P, T.SMALL, audio, ctx, tgt, vis, T.channel_group_of)
            rows.append((d, table))
            worst = max(table.items(), key=lambda kv: kv[1]["pair"])
            print(f"channel-extractor[{stacks}] draw {d:2d}: ctx tokens {int((~ctx).sum()):4d}  loss hip {lo:.6f} oracle-bf16 {lb:.6f} fp32 {l32:.6f}  "
                  f"noisiest group {worst[0]}: pair {worst[1]['pair']:.5f} d_orc {worst[1]['d_orc']:.5f} d_hip {worst[1]['d_hip']:.5f}")
        all_ok &= report(f"ConvChannelFeatureExtractor, {stacks} stacks, 3 clips x 2 x 99 tokens (small model)", rows)
    # BASE model at 64 clips (the sizes of the headline step's kernels), AudioSet masker
    m, P = T.build(T.BASE)
    audio = torch.from_numpy(synth.synth_audio(64, 1, 32159, seed=3)).to(torch.bfloat16).to(T.dev())
    rows = []
    for d in range(a.base_draws):
        ctx, tgt, vis = TimeInverseBlockMasker(4, 0.65, 10, 0.25, 10, 0.1)(batch_size=64, n_times=200, in_channels=1)
        table, lo, lb, l32 = one_draw(m, P, T.BASE, audio, ctx, tgt, vis, T.group_of)
        rows.append((d, table))
        print(f"base-64 draw {d}: loss hip {lo:.6f} oracle-bf16 {lb:.6f} fp32 {l32:.6f}")
    all_ok &= report("BASE model, 64 clips x 200 tokens", rows)
    print("\nALL DRAWS INSIDE THE BOUND" if all_ok else "\nBOUND VIOLATED ON SOME DRAW")
    return 0 if all_ok else 1


if __name__ == "__main__":
    sys.exit(main())

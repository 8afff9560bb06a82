"""Root-cause harness for order / uninitialised-memory / timing effects in one small training step (run on the GPU box).

  python tools/debug_order.py order   : the same two-step run on several freshly built models, in a given order of conv-backward
                                        modes; prints pairwise gradient differences and each run's error against the fp32 oracle
  WJ_ARENA_FILL=nan python tools/debug_order.py nan : arena poisoned with NaN; any NaN in loss / gradients = a read of
                                        never-written memory
  python tools/debug_order.py sleep   : the same model mode with host sleeps injected between forward and backward
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import synth  # noqa: E402
from oracle import jepa_oracle as J  # noqa: E402
from tests.test_jepa_gpu import SMALL, build, oracle_kw  # noqa: E402

fx = dict(np.load(os.path.join(ROOT, "tests/golden/masks.npz")))
sets = [tuple(torch.from_numpy(fx[k][i:i + 3]) for k in ("as_ctx", "as_tgt", "as_vis")) for i in (0, 3)]
audios = [torch.from_numpy(synth.synth_audio(3, 1, 32159, seed=11 + i)).to(torch.bfloat16).cuda() for i in range(2)]
KEYS = ("extract_audio", "feature_norms", "post_extraction_mapper", "encoder.layers.0", "decoder.layers.1")


def run(sparse, ragged=True, sleep=0.0, steps=2, empty_cache=False):
    if empty_cache:
        torch.cuda.empty_cache()
    m, P = build(SMALL)
    eng = m._ensure_engine()
    eng.sparse_conv = sparse
    eng.ragged = ragged
    for i in range(steps):
        m.zero_grad(set_to_none=True)
        out = m(audios[i % 2], *sets[i % 2])
        if sleep:
            torch.cuda.synchronize()
            time.sleep(sleep)
        out["loss"].backward()
    torch.cuda.synchronize()
    g = {k: p.grad.double().clone() for k, p in m.named_parameters() if p.grad is not None}
    return float(out["loss"]), g, P


def oracle_grads(P, i, mode):
    P = {k: v.detach().clone() for k, v in P.items()}
    names = J.trainable_names(P)
    for k in names:
        P[k].requires_grad_(True)
    a = audios[i] if mode == "bf16" else audios[i].float()
    ctx, tgt, vis = (t.cuda() for t in sets[i])
    ref = J.jepa_forward(P, a, ctx, tgt, vis, mode=mode, **oracle_kw(SMALL))
    ref["loss"].backward()
    return float(ref["loss"]), {k: P[k].grad.double() for k in names}


def diff(a, b, keys=None):
    out = {}
    for k in b:
        if keys and not k.startswith(keys):
            continue
        out[k] = float((a[k] - b[k]).norm() / (b[k].norm() + 1e-30))
    return out


def worst(d, n=4):
    return sorted(((v, k) for k, v in d.items()), reverse=True)[:n]


mode = sys.argv[1] if len(sys.argv) > 1 else "order"
if mode == "order":
    order = sys.argv[2] if len(sys.argv) > 2 else "sdd"        # s = sparse conv backward, d = dense conv backward
    ec = len(sys.argv) > 3 and sys.argv[3] == "empty_cache"
    runs = [run(ch == "s", empty_cache=ec) for ch in order]
    lo32, g32 = oracle_grads(runs[0][2], 1, "fp32")
    lo16, g16 = oracle_grads(runs[0][2], 1, "bf16")
    print("order", order, "empty_cache" if ec else "", "losses", [r[0] for r in runs], "oracle fp32/bf16", lo32, lo16)
    for i, r in enumerate(runs):
        print(f"  run {i} ({order[i]}) vs oracle fp32: worst", worst(diff(r[1], g32)), "| vs oracle bf16: worst", worst(diff(r[1], g16)))
    for i in range(len(runs)):
        for j in range(i):
            print(f"  run {i} ({order[i]}) vs run {j} ({order[j]}): worst", worst(diff(runs[i][1], runs[j][1])))
elif mode == "nan":
    for sparse in (True, False):
        for ragged in (True, False):
            lo, g, _ = run(sparse, ragged=ragged)
            bad = [k for k, v in g.items() if not torch.isfinite(v).all()]
            print(f"arena fill {os.environ.get('WJ_ARENA_FILL')!r} sparse={sparse} ragged={ragged}: loss {lo}, non-finite grads: {len(bad)} {bad[:6]}")
elif mode == "sleep":
    base = run(True)
    for s in (0.0, 0.05, 0.2):
        r = run(True, sleep=s)
        print("sleep", s, "loss", r[0], "worst vs first:", worst(diff(r[1], base[1])))

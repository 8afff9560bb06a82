"""Run-to-run reproducibility of one small training step (fp32 atomics in the reductions reorder sums; the noise grows
down the conv stack).  Yardstick for the sparse-vs-dense and ragged-vs-dense equality tests.  Run on the GPU box."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from tests.test_jepa_gpu import build, SMALL
fx = dict(np.load(os.path.join(ROOT, "tests/golden/masks.npz")))
sets = [tuple(torch.from_numpy(fx[k][i:i + 3]) for k in ("as_ctx", "as_tgt", "as_vis")) for i in (0, 3)]
audios = [torch.from_numpy(synth.synth_audio(3, 1, 32159, seed=11 + i)).to(torch.bfloat16).cuda() for i in range(2)]
G = []
for r in range(5):
    m, _ = build(SMALL)
    eng = m._ensure_engine()
    eng.sparse_conv = bool(int(os.environ.get("SP", "0")))
    eng.use_side = bool(int(os.environ.get("SIDE", "1")))
    for i in range(2):
        m.zero_grad(set_to_none=True)
        out = m(audios[i], *sets[i]); out["loss"].backward()
    torch.cuda.synchronize()
    G.append((float(out["loss"]), {k: p.grad.double().clone() for k, p in m.named_parameters() if p.grad is not None}))
for r in range(1, 5):
    worst = max(((float((G[r][1][k] - G[0][1][k]).norm() / (G[0][1][k].norm() + 1e-30)), k) for k in G[0][1]))
    print(r, "loss", G[r][0], G[0][0], "worst grad rel diff vs run 0:", worst)

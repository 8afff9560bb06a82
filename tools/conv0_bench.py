#!/usr/bin/env python3
"""conv0 + GroupNorm + GELU forward at the step's size (256 clips of 32 160 samples, 512 channels): us per launch of the whole entry
point, and with WJ_CONV0_DUMP=<file> the activation of the first two clips (to compare the MFMA and the VALU apply pass:
WJ_CONV0_APPLY_MFMA=0/1 is read once per process).  usage: conv0_bench.py [other_dump.pt]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavjepa_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
N, C_in, L, C, k, s = 256, 1, 32160, 512, 10, 5
L_out = (L - k) // s + 1
P = (L_out + 15) // 16 * 16
g = torch.Generator(device="cpu").manual_seed(0)
audio = torch.randn(N, C_in, L, generator=g).to(torch.bfloat16).to(dev)
w = (torch.randn(C, C_in, k, generator=g) * (2.0 / (C_in * k)) ** 0.5).to(torch.bfloat16).to(dev)
gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(dev)
beta = (0.1 * torch.randn(C, generator=g)).to(dev)
act = torch.empty(N, P, C, dtype=torch.bfloat16, device=dev)
stats = torch.empty(2, N, C, device=dev)
ws = torch.empty(ops.workspace_bytes("wj_conv0_gn_gelu_fwd", N=N, C_in=C_in, C=C, k=k, L_out=L_out) // 4, device=dev)
yx, x1 = torch.empty(N, C, C_in * k, device=dev), torch.empty(N, C_in * k, device=dev)
junk = torch.empty(256 * 1024 * 1024, device=dev)


def run():
    ops.conv0_fwd(audio, w, gamma, beta, act, stats[0], stats[1], ws, N=N, C_in=C_in, L=L, C=C, k=k, stride=s, L_out=L_out, P=P, yx=yx, x1=x1)


for _ in range(3):
    run()
torch.cuda.synchronize()
ts = []
for r in range(12):
    junk.fill_(float(r))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print(f"conv0 forward (stats + fold + apply) N={N}: median {ts[len(ts) // 2]:.1f} us, min {ts[0]:.1f} us; activation {act.numel() * 2 / 1e9:.2f} GB "
      f"(WJ_CONV0_APPLY_MFMA={os.environ.get('WJ_CONV0_APPLY_MFMA', '1')})")
if os.environ.get("WJ_CONV0_DUMP"):
    torch.save(dict(act=act[:2].cpu(), mean=stats[0].cpu(), rstd=stats[1].cpu()), os.environ["WJ_CONV0_DUMP"])
if len(sys.argv) > 1:
    other = torch.load(sys.argv[1])
    a, b = act[:2].cpu().float(), other["act"].float()
    diff = (a != b)
    print(f"against {sys.argv[1]}: {int(diff.sum())} of {a.numel()} elements differ ({float(diff.float().mean()):.2e}), "
          f"max |diff| {float((a - b).abs().max()):.3e}, rel l2 {float((a - b).norm() / b.norm()):.3e}; "
          f"mean equal {torch.equal(stats[0].cpu(), other['mean'])}, rstd equal {torch.equal(stats[1].cpu(), other['rstd'])}")
